# GPU call r04y: closing validation of the round-4 tree: full -m gpu suite, smoke(), the bench line in the driver's own form (PMC traffic + CPU leg),
# rocprofv3 kernel stats of the same command, single-query flat scan probe
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04y; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; tail -3 $O/gpu_suite.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-300
t0=$(date +%s)
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); r=j['roofline']; print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step']); print('ROOF', r['kernel'][:40], r['bound'], r['achieved'], r['frac'], r['traffic'], r['avg_launch_ms']); print('CPU', j['cpu_baseline']['value'], j['cpu_baseline']['ids_equal_gpu'])" | cut -c1-900
timeout 500 python bench.py --steps 200 --warmup 10 --no-pmc --no-cpu-baseline > $O/bench_200.json 2> $O/bench_200.err; python -c "
import json; j=json.loads(open('$O/bench_200.json').read().strip().splitlines()[-1]); print('BENCH200', j['value'], j['ms_per_step'])"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); cp "$f" $O/bench_kernel_stats.csv; head -6 $O/bench_kernel_stats.csv | cut -c1-170
timeout 200 python scripts/probe_flat_small.py > $O/flat_small.json 2> $O/flat_small.err; tail -1 $O/flat_small.json | cut -c1-600
