# GPU call r04zb: where the merge kernel's 0.16 ms goes: timing experiments with early returns (LANCE_HIP_QM_DBG, results wrong)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zb; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-pmc --no-cpu-baseline --steps 10 --warmup 3 --streams 1"
for d in 0 1 2 3 4 5; do LANCE_HIP_QM_DBG=$d timeout 100 $B 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg$d merge', j['kernel_ms_per_step']['ivfpq_merge'], 'refine', j['kernel_ms_per_step']['refine'])"; done
