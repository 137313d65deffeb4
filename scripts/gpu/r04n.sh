# GPU call r04n: where the rows-on-lanes scan (v4) spends a wave's life: s_memtime stamps, and two timing experiments (flush dropped / nothing passes)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04n; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --no-pmc --no-cpu-baseline --steps 10 --warmup 2"
timeout 200 $B > $O/b0.json 2>$O/b0.err; python -c "
import json; j=json.loads(open('$O/b0.json').read().strip().splitlines()[-1]); print('DEF', j['value'], j['kernel_ms_per_step'])"
LANCE_HIP_GRAPH=0 LANCE_HIP_MS_PROF4=1 timeout 200 $B --streams 1 2>&1 | grep "ms4 prof" | tail -2
LANCE_HIP_MS_DBG=1 timeout 200 $B > $O/b1.json 2>$O/b1.err; python -c "
import json; j=json.loads(open('$O/b1.json').read().strip().splitlines()[-1]); print('NOFLUSH', j['value'], j['kernel_ms_per_step'])"
LANCE_HIP_MS_DBG=2 timeout 200 $B > $O/b2.json 2>$O/b2.err; python -c "
import json; j=json.loads(open('$O/b2.json').read().strip().splitlines()[-1]); print('NOPASS', j['value'], j['kernel_ms_per_step'])"
