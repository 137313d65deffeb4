cd $GRAFT_REPO_ROOT; O=gpurun_out/r06o; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-400
LANCE_HIP_XF_PROF=1 timeout 600 python scripts/probe_xform_c3.py 2>&1 | grep -E "xf tail prof|C3_XFORM" | tail -3 | cut -c1-300
LANCE_HIP_XF_PROF=1 timeout 600 python scripts/probe_xform.py c2 2>&1 | grep -v amdgpu | grep -E "xf prof" | tail -1 | cut -c1-420
OUT=$O/probe_fused.json timeout 600 python scripts/probe_xform.py all 2>&1 | grep -v amdgpu | cut -c1-300
timeout 600 python scripts/diff_routes.py > $O/diff_routes.txt 2>&1; echo "diff rc=$?"; grep -v amdgpu $O/diff_routes.txt | tail -5 | cut -c1-300
