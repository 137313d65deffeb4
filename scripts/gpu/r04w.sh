# GPU call r04w: after the threshold / scratch-limit change: mscan + pm-scan parity, C4-shaped default, C3 at 10,000-query batches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04w; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 300 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -2 $O/tests.txt | cut -c1-300
timeout 300 python bench.py --config c4 --n 8000000 --no-pmc --no-cpu-baseline --steps 10 --warmup 3 > $O/c4.json 2>$O/c4.err; python -c "
import json; j=json.loads(open('$O/c4.json').read().strip().splitlines()[-1]); print('c4 default', round(j['value']), j['recall_at_10'], j['kernel_ms_per_step'])"
timeout 300 python scripts/probe_c3_search.py 1000000 10000 > $O/c3_10000.json 2> $O/c3_10000.err; python -c "
import json; j=json.loads(open('$O/c3_10000.json').read().strip().splitlines()[-1]); print('c3 nq=10000', {k: (v['qps_async_one_context'], v['sum_ms']) for k, v in j.items() if k.startswith('nprobes')}); print(j['nprobes50_refine10']['kernel_ms_per_batch'])"
tail -2 $O/c3_10000.err | cut -c1-300
