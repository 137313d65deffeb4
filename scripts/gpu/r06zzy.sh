# GPU call r06zzy: slice height 2048 vs 3072 (and neighbours) on the final kernel: the bench's batch and three other batch shapes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzy; mkdir -p $O; export TMPDIR=/tmp
run() { LANCE_HIP_MS_RS=$1 timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras ${@:2} > $O/b.json 2> $O/b.err
python -c "
import json; j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('RS $*', round(j['value']), round(j['ms_per_step'],4), j['kernel_ms_per_step']['ivfpq_scan_c1'], j['recall_at_10'])"; }
for v in 2816 3072 3328 3584; do run $v; done
for v in 2048 3072; do run $v --nprobes 25; run $v --nq 4000; run $v --n 2000000; run $v --nprobes 25 --refine 0; done
