# GPU call r05y: validation of the FINAL round-5 tree: full -m gpu suite, smoke(), the bench line in the driver's own form (PMC traffic + CPU leg), rocprofv3
# kernel stats of the same command, C4-shaped run (f16 column, nlist 4096) on the integer path
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05y; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-300
t0=$(date +%s)
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); r=j['roofline']; print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step']); print('ROOF', r['kernel'][:40], r['bound'], r['achieved'], r['frac'], r['traffic'], r['avg_launch_ms']); print('CPU', j['cpu_baseline']['value'], j['cpu_baseline']['ids_equal_gpu']); print('LAT', {k: v for k, v in j['latency'].items() if k != 'what'}); print('BUILD', j['roofline_build']['build_stages_ms']); print('TAIL', {k: (round(v['avg_launch_ms'],4), round(v.get('frac', 0),3)) for k, v in j['roofline_tail'].items()})" | cut -c1-1200
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 50 --warmup 5 --no-pmc --no-cpu-baseline --no-grid > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); cp "$f" $O/bench_kernel_stats.csv; head -8 $O/bench_kernel_stats.csv | cut -c1-170
timeout 400 python bench.py --config c4 --n 8000000 --steps 50 --warmup 5 --no-pmc --no-cpu-baseline --no-grid > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"; python -c "
import json; j=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1]); print('C4', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" | cut -c1-600
