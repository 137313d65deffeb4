# GPU call r06zzs: s_memtime phase stamps of the matrix-core scan on the current tree (LANCE_HIP_MS_PROF)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzs; mkdir -p $O; export TMPDIR=/tmp
LANCE_HIP_MS_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --streams 1 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench.json 2> $O/prof.err
grep "ms prof" $O/prof.err | tail -4 | cut -c1-900
