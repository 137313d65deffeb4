# GPU call r06zw: dot flow, error terms separated by what they scale with: parity, rates, survivor statistics
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zw; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_dot_flow.py -x -q -m gpu --timeout 600 > $O/dot_flow.txt 2>&1; echo "dot flow rc=$?"; tail -3 $O/dot_flow.txt | cut -c1-400
timeout 900 python scripts/probe_dot_flow.py child > $O/dot_probe.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/dot_probe.txt | cut -c1-600
LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_dot_flow.py child 2>&1 | grep "qscan\]" | uniq -c | head -12 | cut -c1-400
