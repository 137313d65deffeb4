# GPU call r04zd: how many segments the matrix-core scan hands to the exact rescan at C2 (bursts beyond a queue half + overflowed segments)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zd; mkdir -p $O
export TMPDIR=/tmp
LANCE_HIP_Q_STATS=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep qscan | tail -2
LANCE_HIP_NO_MSCAN=1 LANCE_HIP_Q_STATS=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep qscan | tail -1
