# GPU call r03x: the 8-queries-per-gather filter scan (search_q8.hip, LANCE_HIP_Q8=1): parity tests, bench A/B, survivor counts, fuzz burst
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03x; mkdir -p $O
LANCE_HIP_Q8=1 timeout 400 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_fuzz_findings.py -m gpu -q -x --timeout 300 > $O/tests_q8.txt 2>&1; echo "rc=$?"; tail -4 $O/tests_q8.txt | cut -c1-400
LANCE_HIP_Q8=1 timeout 200 python bench.py --no-cpu-baseline --steps 20 > $O/bench_q8.json 2>$O/bench_q8.err; python -c "import json; j=json.loads(open('$O/bench_q8.json').read().strip().splitlines()[-1]); print('bench q8', j['value'], j['ms_per_step'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])"
timeout 200 python bench.py --no-cpu-baseline --steps 20 > $O/bench_base.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/bench_base.json').read().strip().splitlines()[-1]); print('bench base', j['value'], j['ms_per_step'], j['recall_at_10'], j['kernel_ms_per_step'])"
LANCE_HIP_Q8=1 LANCE_HIP_Q_STATS=1 timeout 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $O/stats_q8.err; grep qscan $O/stats_q8.err | sort | uniq -c | sort -rn | head -3 | cut -c1-250
LANCE_HIP_Q_STATS=1 timeout 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $O/stats_base.err; grep qscan $O/stats_base.err | sort | uniq -c | sort -rn | head -3 | cut -c1-250
(LANCE_HIP_Q8=1 timeout 100 python tests/fuzz_parity.py 60 81 --log $O/fuzz_81.txt > /dev/null 2>&1 &
 LANCE_HIP_Q8=1 timeout 100 python tests/fuzz_parity.py 60 82 --log $O/fuzz_82.txt > /dev/null 2>&1 &
 wait)
grep -hE "SKIP|MISMATCH|ERROR|fuzz " $O/fuzz_81.txt $O/fuzz_82.txt | cut -c1-400
