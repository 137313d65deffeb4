# GPU call r04v: where the matrix-core scan stops paying -- C4-shaped bench (f16 rows, nlist 4096: ~24 pairs per partition at 10k x 10) with
# and without it, and the threshold in between; C3 probe at 1000- and 10,000-query batches on the final tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04v; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --config c4 --n 8000000 --no-pmc --no-cpu-baseline --steps 10 --warmup 3"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/c4_$tag.json 2>$O/c4_$tag.err; python -c "
import json; j=json.loads(open('$O/c4_$tag.json').read().strip().splitlines()[-1]); print('c4 $tag', round(j['value']), j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])"; }
run mscan X=1
run noms LANCE_HIP_NO_MSCAN=1
timeout 300 python scripts/probe_c3_search.py 1000000 1000 > $O/c3_1000.json 2> $O/c3_1000.err; python -c "
import json; j=json.loads(open('$O/c3_1000.json').read().strip().splitlines()[-1]); print('c3 nq=1000', {k: (v['qps_async_one_context'], v['sum_ms']) for k, v in j.items() if k.startswith('nprobes')})"
timeout 300 python scripts/probe_c3_search.py 1000000 10000 > $O/c3_10000.json 2> $O/c3_10000.err; python -c "
import json; j=json.loads(open('$O/c3_10000.json').read().strip().splitlines()[-1]); print('c3 nq=10000', {k: (v['qps_async_one_context'], v['sum_ms'], v['kernel_ms_per_batch']) for k, v in j.items() if k.startswith('nprobes')})"
