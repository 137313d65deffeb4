# GPU call r06zn: validation after the search paths stopped synchronising the whole device (stream wait instead): the order that failed x3, the whole suite, smoke, bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zn; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do timeout 1500 python -m pytest tests/test_zz_gpu_coarse_mfma.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_threads.py -x -q -m gpu --timeout 1400 > $O/tests_order$i.txt 2>&1; echo "order run $i rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_order$i.txt | cut -c1-300 | tail -3; done
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-300
timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c2.json 2> $O/c2.err; echo "c2 rc=$?"
python -c "
import json; j=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print('C2', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'])" 2>&1 | cut -c1-300
