cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-300
timeout 300 python scripts/probe_xf_bug.py 2>&1 | grep -v amdgpu | tail -12 | cut -c1-300
OUT=$O/probe_fused.json timeout 600 python scripts/probe_xform.py all 2>&1 | grep -v amdgpu | cut -c1-420
timeout 600 python scripts/diff_routes.py > $O/diff_routes.txt 2>&1; echo "diff rc=$?"; grep -v amdgpu $O/diff_routes.txt | tail -8 | cut -c1-300
