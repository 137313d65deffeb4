cd $GRAFT_REPO_ROOT; O=gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/probe_xf_bug.py 2>&1 | grep -v amdgpu | tail -12 | cut -c1-300
