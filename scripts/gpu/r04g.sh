# GPU call r04g: state of the round-4 tree (re-entry): full suite, bench line with PMC + CPU leg, rocprof kernel stats, A/B of the opt-in switches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; tail -3 $O/gpu_suite.txt | cut -c1-300
t0=$(date +%s)
timeout 500 python bench.py --steps 200 --warmup 10 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); print('DEF', j['value'], j['ms_per_step'], j['kernel_ms_per_step'], j['build_sec'], j['roofline']['frac'], j['roofline']['traffic'], (j.get('cpu_baseline') or {}).get('value'), j['roofline_build'])" | cut -c1-2500
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); cp "$f" $O/bench_kernel_stats.csv; head -25 $O/bench_kernel_stats.csv | cut -c1-200
LANCE_HIP_MFMA_TABLE=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_mb.json 2> $O/bench_mb.err; python -c "
import json; j=json.loads(open('$O/bench_mb.json').read().strip().splitlines()[-1]); print('MFMA_TABLE', j['value'], j['kernel_ms_per_step'])"
LANCE_HIP_GRAPH=0 timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_nograph.json 2> $O/bench_nograph.err; python -c "
import json; j=json.loads(open('$O/bench_nograph.json').read().strip().splitlines()[-1]); print('NOGRAPH', j['value'], j['kernel_ms_per_step'])"
for mode in 0 1 2; do LANCE_HIP_QPT=$mode timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt$mode.json 2> $O/c3_qpt$mode.err; echo "C3 QPT=$mode"; tail -1 $O/c3_qpt$mode.json | cut -c1-1600; done
