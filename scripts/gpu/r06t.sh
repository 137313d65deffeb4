# GPU call r06t: C5-shaped run at ONE BILLION rows (1e9 x 128 int8 = 128 GB + 32 GB codes x 2 + ids), nlist 65536 hierarchical, M 32, one GPU
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06t; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 2400 python bench.py --config c5 --n 1000000000 --nprobes 32 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-extras > $O/bench_c5_1B.json 2> $O/bench_c5_1B.err; echo "c5 1B rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_c5_1B.json').read().strip().splitlines()[-1]); print('C5-1B', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['build_stages_ms'], j['kernel_ms_per_step']); print([ (g['nprobes'], g['refine_factor'], g['recall_at_10'], g['ms_per_1000_queries']) for g in (j['recall_grid'] or [])])" 2>&1 | cut -c1-1200
tail -4 $O/bench_c5_1B.err | cut -c1-400
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
