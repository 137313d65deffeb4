# GPU call r04ze: the final tree once more: full -m gpu suite, then three fuzz processes in parallel (seeds 5001-5003, 200 s each)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04ze; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; tail -2 $O/gpu_suite.txt | cut -c1-300
for s in 5001 5002 5003; do (timeout 260 python tests/fuzz_parity.py 200 $s --log $O/fuzz_$s.txt > $O/fuzz_$s.out 2>&1; echo "fuzz $s rc=$?") & done
wait
for s in 5001 5002 5003; do tail -1 $O/fuzz_$s.out | cut -c1-250; done
