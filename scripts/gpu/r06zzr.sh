# GPU call r06zzr: L2 bound pass over each query's 1 / 2 / 3 nearest lists (LANCE_HIP_BOUND_LISTS): parity at 2, bench at 1 / 2 / 3
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzr; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
LANCE_HIP_BOUND_LISTS=2 timeout 900 python -m pytest tests/test_zz_gpu_mscan.py tests/test_zz_gpu_msbound.py tests/test_gpu_parity.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt | cut -c1-200
for v in 1 2 3 1 2; do
LANCE_HIP_BOUND_LISTS=$v timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']; print('BENCH $v', j['value'], j['ms_per_step'], 'bound', k['ivfpq_scan_c0'], 'resid', k['q_residual'], 'scan', k['ivfpq_scan_c1'], 'merge', k['ivfpq_merge'], 'refine', k['refine'], j['recall_at_10'])"
done
