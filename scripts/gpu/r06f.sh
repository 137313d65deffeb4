cd $GRAFT_REPO_ROOT; O=gpurun_out/r06f; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
LANCE_HIP_XF_PROF=1 OUT=$O/probe_prof.json timeout 600 python scripts/probe_xform.py c2 2>&1 | grep -v amdgpu | grep -E "xf prof" | tail -3 | cut -c1-420
