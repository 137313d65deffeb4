# GPU call r06zzx: slice height of the matrix-core scan on the final kernel (LANCE_HIP_MS_RS)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzx; mkdir -p $O; export TMPDIR=/tmp
for v in 3072 4096 6144 8192 3072 4096 2560; do
LANCE_HIP_MS_RS=$v timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
python -c "
import json; j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print('RS $v', j['value'], j['ms_per_step'], j['kernel_ms_per_step']['ivfpq_scan_c1'])"
done
