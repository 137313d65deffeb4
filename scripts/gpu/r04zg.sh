# GPU call r04zg: phase stamps and the no-survivor floor of the FINAL scan kernel (rotated loop, largest-first slices), for the record
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zg; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-pmc --no-cpu-baseline --steps 10 --warmup 2"
LANCE_HIP_GRAPH=0 LANCE_HIP_MS_PROF=1 timeout 200 $B --streams 1 2>&1 | grep "ms prof" | tail -2
LANCE_HIP_MS_DBG=2 timeout 200 $B 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NOPASS scan', j['kernel_ms_per_step']['ivfpq_scan_c1'])"
LANCE_HIP_MS_DBG=1 timeout 200 $B 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NOFLUSH scan', j['kernel_ms_per_step']['ivfpq_scan_c1'])"
timeout 200 $B 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEF scan', j['kernel_ms_per_step']['ivfpq_scan_c1'], round(j['value']))"
