# GPU call r04zf: the sharded code path of bench.py at world size 1 (RCCL) on the final tree; merge workgroup size A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zf; mkdir -p $O
export TMPDIR=/tmp
LANCE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --steps 5 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "force_dist rc=$?"; python -c "
import json; j=json.loads(open('$O/bench_force_dist.json').read().strip().splitlines()[-1]); print(round(j['value']), j['n_gpus'], j['build_sec'], j['strong_scaling_list_sharded_qps'], j['build_sec_rows_sharded_allreduce'], (j['multi_gpu'] or {}).keys())" | cut -c1-600
tail -2 $O/bench_force_dist.err | cut -c1-300
for v in 128 256; do LANCE_HIP_QMERGE_BS=$v timeout 100 python bench.py --no-pmc --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qmerge_bs$v', round(j['value']), 'merge', j['kernel_ms_per_step']['ivfpq_merge'])"; done
