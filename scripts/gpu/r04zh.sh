# GPU call r04zh: which slices make the scan's tail (per-slice durations under LANCE_HIP_MS_PROF)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
LANCE_HIP_GRAPH=0 LANCE_HIP_MS_PROF=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 3 --warmup 1 --streams 1 2>&1 | grep "ms prof" | tail -2 | cut -c1-1800
