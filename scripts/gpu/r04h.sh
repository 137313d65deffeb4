# GPU call r04h: first hardware run of the matrix-core filter scan (search_ms.hip): parity (new test file + pm-scan suite), fuzz, bench A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 700 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q -x --timeout 600 > $O/mscan_tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -15 $O/mscan_tests.txt | cut -c1-400
t0=$(date +%s)
timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_ms.json 2> $O/bench_ms.err; echo "bench rc=$? $(( $(date +%s)-t0 )) s"; tail -3 $O/bench_ms.err | cut -c1-300
python -c "
import json; j=json.loads(open('$O/bench_ms.json').read().strip().splitlines()[-1]); print('MSCAN', j['value'], j['ms_per_step'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])"
LANCE_HIP_NO_MSCAN=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_noms.json 2> $O/bench_noms.err; python -c "
import json; j=json.loads(open('$O/bench_noms.json').read().strip().splitlines()[-1]); print('NO_MSCAN', j['value'], j['ms_per_step'], j['recall_at_10'], j['kernel_ms_per_step'])"
LANCE_HIP_Q_STATS=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep qscan | tail -2
t0=$(date +%s)
timeout 260 python tests/fuzz_parity.py 200 4101 --log $O/fuzz_4101.txt > $O/fuzz_4101.out 2>&1; echo "fuzz rc=$? $(( $(date +%s)-t0 )) s"; tail -4 $O/fuzz_4101.out | cut -c1-300
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); cp "$f" $O/bench_kernel_stats.csv; head -12 $O/bench_kernel_stats.csv | cut -c1-200
