# GPU call r05h: first run of the long-row matrix-core flat filter: parity, then C3-shaped timing (1M x 1536 cosine, 1000 queries) against the exact kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_zz_gpu_flat_wide.py -m gpu -q --timeout 900 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | cut -c1-300 | tail -30
grep -E "^E  " $O/tests.txt | head -20 | cut -c1-300
timeout 600 python scripts/probe_flat_wide.py > $O/probe.txt 2>&1; echo "probe rc=$?"; tail -12 $O/probe.txt | cut -c1-300
