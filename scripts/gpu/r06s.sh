cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_zz_gpu_two_ranks.py tests/test_zz_gpu_comm.py tests/test_abi.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | head -12 | cut -c1-500
