# GPU call r04za: run-to-run stability of the bench line in the driver's form (one 11.9 M q/s outlier in r04z): default vs GPU_MAX_HW_QUEUES=8
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04za; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-pmc --no-cpu-baseline --steps 20 --warmup 5"
for i in 1 2 3 4 5 6 7 8; do timeout 100 $B 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('def$i', round(j['value']))"; done
for i in 1 2 3 4 5 6 7 8; do GPU_MAX_HW_QUEUES=8 timeout 100 $B 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q8_$i', round(j['value']))"; done
for i in 1 2 3 4; do LANCE_HIP_MS_GRID=224 timeout 100 $B 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g224_$i', round(j['value']))"; done
