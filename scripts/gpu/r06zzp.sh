# GPU call r06zzp: flat_small at 2-4 queries against the batch path (wall time per call)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzp; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "maxq 1 (default)"; timeout 300 python scripts/probe_flat_one.py 2>/dev/null
echo "maxq 4"; LANCE_HIP_FLAT_SMALL_MAXQ=4 timeout 300 python scripts/probe_flat_one.py 2>/dev/null
