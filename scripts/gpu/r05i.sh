# GPU call r05i: checkpoint of the round-5 tree: full -m gpu suite, smoke(), the bench line in the driver's own form (PMC traffic + CPU leg), rocprofv3
# kernel stats of the same command; what zero rows / a zero query do under cosine in the flat scan (exact kernel vs matrix-core filter vs oracle)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05i; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in rows inf tiny query; do
  timeout 120 python scripts/gpu/r05i_debug.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-400
  LANCE_HIP_NO_MFMA_FLAT_WIDE=1 timeout 120 python scripts/gpu/r05i_debug.py $w 2>&1 | grep -v amdgpu.ids | cut -c1-400
done > $O/nan_debug.txt 2>&1; cat $O/nan_debug.txt | head -60
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-300
t0=$(date +%s)
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); r=j['roofline']; print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step']); print('ROOF', r['kernel'][:40], r['bound'], r['achieved'], r['frac'], r['traffic'], r['avg_launch_ms']); print('CPU', j['cpu_baseline']['value'], j['cpu_baseline']['ids_equal_gpu']); print('LAT', j['latency']); print('BUILD', j['roofline_build']['build_stages_ms'])" | cut -c1-1200
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 50 --warmup 5 --no-pmc --no-cpu-baseline --no-grid > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); cp "$f" $O/bench_kernel_stats.csv; head -8 $O/bench_kernel_stats.csv | cut -c1-170
