# GPU call r06zzzx: the last tree of round 6 (LANCE_HIP_MS_RS2 switch added, default behaviour unchanged): full -m gpu suite, smoke(), bench (no PMC / CPU leg)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzzx; mkdir -p $O; export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests -x -q -m gpu --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -5
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-200
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench.json 2> $O/bench.err; python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['kernel_ms_per_step']['ivfpq_scan_c1'])"
