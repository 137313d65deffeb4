# GPU call r06zzk: tiny-magnitude dot test (ties at 1 - x.c): assign, find_partitions, flat filters
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzk; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_dot_flow.py -x -q -m gpu --timeout 600 -k "tiny" > $O/tiny.txt 2>&1; echo "rc=$?"; tail -25 $O/tiny.txt | cut -c1-300
