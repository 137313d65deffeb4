cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-300
timeout 300 python scripts/probe_xf_bug.py 2>&1 | grep -v amdgpu | grep -E "run 0|vs run" | cut -c1-300
LANCE_HIP_XF_PROF=1 timeout 600 python scripts/probe_xform.py c2 2>&1 | grep -v amdgpu | grep -E "xf prof" | tail -1 | cut -c1-420
OUT=$O/probe_fused.json timeout 600 python scripts/probe_xform.py all 2>&1 | grep -v amdgpu | cut -c1-420
