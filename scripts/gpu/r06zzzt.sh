# GPU call r06zzzt: the remaining suite files that call the flat scan, on the last tree (two queries on the single-pass kernel by default)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzzt; mkdir -p $O; export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_zz_gpu_flat_wide.py tests/test_zz_gpu_refine_u8.py tests/test_zz_gpu_dot_flow.py tests/test_abi.py -x -q -m gpu --timeout 90 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -1 $O/tests.txt | cut -c1-200
