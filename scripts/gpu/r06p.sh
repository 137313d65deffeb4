# GPU call r06p: full-size C4 (100M x 128 f16, nlist 4096, M 16) and C5-shaped (100M x 128 int8, nlist 65536 hierarchical, M 32) on the round-6 tree, one GPU
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06p; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python bench.py --config c4 --n 100000000 --steps 50 --warmup 5 --no-pmc --no-cpu-baseline --no-extras > $O/bench_c4_100M.json 2> $O/bench_c4_100M.err; echo "c4 rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_c4_100M.json').read().strip().splitlines()[-1]); print('C4', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['build_stages_ms'], j['kernel_ms_per_step']); print([ (g['nprobes'], g['refine_factor'], g['recall_at_10'], g['ms_per_1000_queries']) for g in (j['recall_grid'] or [])])" 2>&1 | cut -c1-1200
tail -3 $O/bench_c4_100M.err | cut -c1-300
t0=$(date +%s)
timeout 1500 python bench.py --config c5 --n 100000000 --nprobes 32 --steps 50 --warmup 5 --no-pmc --no-cpu-baseline --no-extras > $O/bench_c5_100M.json 2> $O/bench_c5_100M.err; echo "c5 rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_c5_100M.json').read().strip().splitlines()[-1]); print('C5', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['build_stages_ms'], j['kernel_ms_per_step']); print([ (g['nprobes'], g['refine_factor'], g['recall_at_10'], g['ms_per_1000_queries']) for g in (j['recall_grid'] or [])])" 2>&1 | cut -c1-1200
tail -3 $O/bench_c5_100M.err | cut -c1-300
