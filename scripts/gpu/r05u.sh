# GPU call r05u: final tree: full -m gpu suite with -x (the driver's own command), 5 minutes of fuzz on a fresh seed
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05u; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -6
timeout 400 python tests/fuzz_parity.py 300 5208 --log $O/fuzz.txt > $O/fuzz.out 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.out | cut -c1-300
