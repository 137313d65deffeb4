# GPU call r03y: the final tree of round 3 (default path + the q8 variant's child-process test): full GPU suite, smoke, bench line
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03y; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-200
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -1 $O/bench_n1.json | cut -c1-900
