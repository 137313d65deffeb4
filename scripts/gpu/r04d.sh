# GPU call r04d: graph replays with fill kernels instead of memset nodes, DPP folds, QPT table kernel per query, f16-dot long sub-vectors, full suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04d; mkdir -p $O
timeout 1200 python -m pytest tests/test_zz_gpu_coarse_mfma.py tests/test_zz_gpu_graph.py tests/test_zz_gpu_flat_small.py tests/test_zz_gpu_f16_metrics.py -m gpu -q --timeout 900 > $O/new_tests.txt 2>&1; echo "new tests rc=$?"; grep -E "passed|failed|Error|differ" $O/new_tests.txt | cut -c1-400 | tail -20
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_zz_gpu_coarse_mfma.py --deselect tests/test_zz_gpu_graph.py --deselect tests/test_zz_gpu_flat_small.py --deselect tests/test_zz_gpu_f16_metrics.py > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.txt | cut -c1-300
timeout 120 python scripts/probe_flat_small.py > $O/flat_small.json 2> $O/flat_small.err; tail -1 $O/flat_small.json | cut -c1-900
LANCE_HIP_FLAT_SMALL_MAXQ=4 timeout 120 python scripts/probe_flat_small.py > $O/flat_small4.json 2> $O/flat_small4.err; tail -1 $O/flat_small4.json | cut -c1-900
LANCE_HIP_QPT=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt.json 2> $O/c3_qpt.err; tail -1 $O/c3_qpt.json | cut -c1-2000
LANCE_HIP_QPT=1 LANCE_HIP_GRAPH=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt_graph.json 2> $O/c3_qpt_graph.err; tail -1 $O/c3_qpt_graph.json | cut -c1-2000
LANCE_HIP_GRAPH=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline --warmup 30 > $O/bench_graph.json 2> $O/bench_graph.err; tail -1 $O/bench_graph.json | cut -c1-700
