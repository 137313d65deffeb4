# GPU call r06zs: dot flow with the bound pass capped at 65,535 rows of a list; parity, rate, then the whole GPU suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zs; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_dot_flow.py -x -q -m gpu --timeout 600 > $O/dot_flow.txt 2>&1; echo "dot flow rc=$?"; tail -5 $O/dot_flow.txt | cut -c1-400
timeout 600 python scripts/probe_metrics.py > $O/metrics.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/metrics.txt | cut -c1-300
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -8
