# GPU call r06zx: whole GPU suite + fuzz on the tree with the dot metric's quantised flow
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zx; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -8
timeout 700 python tests/fuzz_parity.py 330 6501 --log $O/fuzz.txt --watchdog 300 > $O/fuzz_out.txt 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_out.txt | cut -c1-300; grep -c "'metric': 'dot'" $O/fuzz_out.txt
