# GPU call r06zzzu: two queries on the single-pass flat kernel by default: the flat tests of the suite + wall time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzzu; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz_gpu_flat_small.py tests/test_zz_gpu_f16_metrics.py tests/test_gpu_parity.py -x -q -m gpu --timeout 280 -k "flat or small or f16 or native" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -1 $O/tests.txt | cut -c1-200
timeout 100 python scripts/probe_flat_one.py 2>/dev/null
