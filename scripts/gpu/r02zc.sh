# GPU call r02zc: the SURVEY 8(d) measurement grid (C1 flat, C2 recall / QPS grid + small batches, IVF_FLAT, C3 shape) on the
# round-2 kernels, and the f16 dot assign (32-lane order) on both routes
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02zc; mkdir -p $O
cd $R
timeout 120 python scripts/measure_grid.py --c3 > $O/grid.json 2> $O/grid.err; echo "grid rc=$?"; tail -3 $O/grid.err | cut -c1-300
timeout 60 python scripts/probe_f16_dot.py > $O/f16_dot.txt 2>&1; tail -4 $O/f16_dot.txt | cut -c1-400
