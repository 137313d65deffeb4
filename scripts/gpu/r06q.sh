cd $GRAFT_REPO_ROOT; O=gpurun_out/r06q; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sharded or list_shard or refine" 2>&1 | grep -E "^E  |passed|failed" | head -12 | cut -c1-400
timeout 600 python -m pytest tests/test_abi.py tests/test_zz_gpu_two_ranks.py tests/test_zz_gpu_refine_u8.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-400
