cd $GRAFT_REPO_ROOT; O=gpurun_out/r06n; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-400
LANCE_HIP_XF_PROF=1 timeout 600 python scripts/probe_xform_c3.py 2>&1 | grep -E "xf tail prof|C3_XFORM" | tail -3 | cut -c1-300
