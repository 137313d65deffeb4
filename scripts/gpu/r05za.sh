# GPU call r05za: persistent workgroups of the matrix-core scan: 256 (one per CU) against 240 / 224 / 208, three runs each, interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05za; mkdir -p $O
export TMPDIR=/tmp
B="--steps 200 --warmup 10 --no-pmc --no-cpu-baseline --no-grid"
for rep in 1 2 3; do
  for g in 256 240 224 208; do
    LANCE_HIP_MS_GRID=$g timeout 200 python bench.py $B > $O/b_${g}_$rep.json 2>/dev/null
    python -c "
import json; j=json.loads(open('$O/b_${g}_$rep.json').read().strip().splitlines()[-1]); print('GRID $g rep $rep', round(j['value']), j['ms_per_step'], j['kernel_ms_per_step']['ivfpq_scan_c1'])"
  done
done
