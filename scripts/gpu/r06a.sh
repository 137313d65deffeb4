# GPU call r06a: first contact of the single-pass transform kernel (xform_fused.hip): parity, then timing against the round-5 route
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q --timeout 600 -x > $O/tests_xf.txt 2>&1; echo "xform tests rc=$?"; tail -3 $O/tests_xf.txt; grep -E "^E  |^FAILED" $O/tests_xf.txt | head -12 | cut -c1-400
OUT=$O/probe_fused.json timeout 600 python scripts/probe_xform.py all 2>&1 | grep -v amdgpu | cut -c1-600
LANCE_HIP_NO_XFORM_FUSED=1 OUT=$O/probe_r05route.json timeout 600 python scripts/probe_xform.py all 2>&1 | grep -v amdgpu | cut -c1-600
