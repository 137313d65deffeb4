# GPU call r06zzzs: tests/test_gpu_parity.py in full on the last tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzzs; mkdir -p $O; export TMPDIR=/tmp
timeout 65 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 60 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -1 $O/tests.txt | cut -c1-200
