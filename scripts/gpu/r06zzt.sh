# GPU call r06zzt: what the scan costs without its flush atomics / without survivors (LANCE_HIP_MS_DBG 1 / 2: timing only, results are wrong by design)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzt; mkdir -p $O; export TMPDIR=/tmp
export LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_mstime.so
for v in 0 1 2 3; do
LANCE_HIP_MS_DBG=$v timeout 300 python bench.py --steps 30 --warmup 3 --streams 1 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench_$v.json 2> $O/err_$v.txt
python -c "
import json; j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']; print('DBG $v', j['ms_per_step'], 'scan', k['ivfpq_scan_c1'], 'merge', k['ivfpq_merge'])"
done
