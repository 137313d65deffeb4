# GPU call r03g: merge with the integer-sum cut + compaction -- parity (suite), C2 bench A/B (LANCE_HIP_NO_QCUT), C3 breakdown A/B; fuzz run 3
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 400 -x > $O/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/pytest_all.log | cut -c1-300 | tail -10
timeout 200 python bench.py --no-cpu-baseline > $O/bench_cut.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/bench_cut.json').read().strip().splitlines()[-1]); print('cut   ', j['value'], j['ms_per_step'], j['kernel_ms_per_step'], j['recall_at_10'])"
LANCE_HIP_NO_QCUT=1 timeout 200 python bench.py --no-cpu-baseline > $O/bench_nocut.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/bench_nocut.json').read().strip().splitlines()[-1]); print('no cut', j['value'], j['ms_per_step'], j['kernel_ms_per_step'], j['recall_at_10'])"
timeout 300 python scripts/probe_c3_search.py > $O/c3_probe_cut.json 2> $O/c3_probe.err; cat $O/c3_probe_cut.json | cut -c1-2500
LANCE_HIP_NO_QCUT=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_probe_nocut.json 2> $O/c3_probe2.err; cat $O/c3_probe_nocut.json | cut -c1-2500
for s in 22 31 32; do OMP_NUM_THREADS=5 OMP_WAIT_POLICY=PASSIVE timeout 640 python tests/fuzz_parity.py 520 $s --watchdog 120 --log $O/fuzz_seed$s.log > $O/fuzz_seed$s.out 2> $O/fuzz_seed$s.err & done
wait
for s in 22 31 32; do tail -n 3 $O/fuzz_seed$s.log | cut -c1-500; grep -c "Thread 0x\|most recent call" $O/fuzz_seed$s.err; done
