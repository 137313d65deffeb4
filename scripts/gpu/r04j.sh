# GPU call r04j: v2 with hoisted compares + slim survivor path + tile-level burst handling: parity, bench, fuzz, kernel stats
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q -x --timeout 600 > $O/mscan_tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -8 $O/mscan_tests.txt | cut -c1-400
timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_ms2.json 2> $O/bench_ms2.err; python -c "
import json; j=json.loads(open('$O/bench_ms2.json').read().strip().splitlines()[-1]); print('MSCAN2', j['value'], j['ms_per_step'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])"
LANCE_HIP_MS_V1=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_ms1.json 2> $O/bench_ms1.err; python -c "
import json; j=json.loads(open('$O/bench_ms1.json').read().strip().splitlines()[-1]); print('MSCAN1', j['value'], j['ms_per_step'], j['kernel_ms_per_step'])"
LANCE_HIP_Q_STATS=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep qscan | tail -1
t0=$(date +%s)
timeout 200 python tests/fuzz_parity.py 120 4303 --log $O/fuzz_4303.txt > $O/fuzz_4303.out 2>&1; echo "fuzz rc=$? $(( $(date +%s)-t0 )) s"; tail -3 $O/fuzz_4303.out | cut -c1-300
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); cp "$f" $O/bench_kernel_stats.csv; head -8 $O/bench_kernel_stats.csv | cut -c1-160
