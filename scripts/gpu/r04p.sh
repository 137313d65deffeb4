# GPU call r04p: full suite on the tree with the matrix-core scan + Lloyd-iteration graphs; full bench line (PMC traffic + CPU leg); build A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; tail -5 $O/gpu_suite.txt | cut -c1-300
t0=$(date +%s)
timeout 500 python bench.py --steps 200 --warmup 10 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$? $(( $(date +%s)-t0 )) s"; tail -2 $O/bench_n1.err | cut -c1-300
python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); print('DEF', j['value'], j['ms_per_step'], j['kernel_ms_per_step'], j['build_sec'], j['build_stages_ms'], j['roofline'], (j.get('cpu_baseline') or {}))" | cut -c1-3500
LANCE_HIP_KMEANS_GRAPH=0 timeout 300 python bench.py --no-pmc --no-cpu-baseline --steps 5 > $O/bench_nokg.json 2> $O/bench_nokg.err; python -c "
import json; j=json.loads(open('$O/bench_nokg.json').read().strip().splitlines()[-1]); print('NO_KMEANS_GRAPH', j['build_sec'], j['build_stages_ms'])"
