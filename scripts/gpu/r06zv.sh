# GPU call r06zv: dot flow with the skew guard: rates on SIFT-like rows as they are / centred / unit-normalised (dot, cosine, l2 side by side)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zv; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/probe_dot_flow.py child > $O/dot_probe.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/dot_probe.txt | cut -c1-600
LANCE_HIP_NO_DOT_FLOW=1 timeout 900 python scripts/probe_dot_flow.py child > $O/dot_probe_exact.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/dot_probe_exact.txt | grep -A1 " dot" | cut -c1-600
