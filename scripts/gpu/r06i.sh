cd $GRAFT_REPO_ROOT; O=gpurun_out/r06i; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-300
LANCE_HIP_XF_PROF=1 timeout 600 python scripts/probe_xform.py c2 2>&1 | grep -v amdgpu | grep -E "xf prof" | tail -1 | cut -c1-420
for g in 0 256 384 512 768; do LANCE_HIP_XF_GRID=$g timeout 600 python scripts/probe_xform.py c2 2>&1 | grep -v amdgpu | grep "^c2" | cut -c100-330; done
