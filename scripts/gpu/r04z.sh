# GPU call r04z: does leaving a few CUs to the other engine contexts' latency-bound kernels pay?  (persistent scan grid 256 / 240 / 224 / 192)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04z; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-pmc --no-cpu-baseline --steps 100 --warmup 10"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/b_$tag.json 2>$O/b_$tag.err; python -c "
import json; j=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(j['value']), j['ms_per_step'], j['kernel_ms_per_step']['ivfpq_scan_c1'])"; }
run g256 X=1
run g240 LANCE_HIP_MS_GRID=240
run g224 LANCE_HIP_MS_GRID=224
run g192 LANCE_HIP_MS_GRID=192
run g256b X=1
