# GPU call r06zzzv: flat_small at 2-4 queries with the chunk loads batched: parity (incl. the MAXQ=4 child run) + wall time against the batch path
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzzv; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz_gpu_flat_small.py -x -q -m gpu --timeout 280 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -1 $O/tests.txt | cut -c1-200
echo "maxq 4"; LANCE_HIP_FLAT_SMALL_MAXQ=4 timeout 100 python scripts/probe_flat_one.py 2>/dev/null
echo "maxq 1"; timeout 100 python scripts/probe_flat_one.py 2>/dev/null | head -1
