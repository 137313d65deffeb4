# GPU call r06zzw: the new merge-refill / flag-word test (and the rest of the single-query flat tests)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzw; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_flat_small.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt | cut -c1-300
