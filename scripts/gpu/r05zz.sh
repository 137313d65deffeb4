# GPU call r05zz: captured graphs keyed on the raw attachment's generation: the new test, then the driver's own suite command, smoke, a short bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05zz; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_refine_u8.py tests/test_zz_gpu_graph.py -m gpu -q --timeout 500 > $O/tests_new.txt 2>&1; echo "new tests rc=$?"; tail -1 $O/tests_new.txt; grep -E "^E  |^FAILED" $O/tests_new.txt | head -6 | cut -c1-300
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-200
timeout 300 python bench.py --no-pmc --no-cpu-baseline --no-grid > $O/bench.json 2> $O/bench.err; python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'])"
