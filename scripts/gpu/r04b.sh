# GPU call r04b: round-4 batch 1 on hardware -- MFMA coarse quantiser at query time, QPT with vector table loads, thread-safe
# contexts, C4 / C5-trained goldens, the full suite, C3 timing with / without QPT, the bench line with in-run PMC traffic
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_zz_gpu_coarse_mfma.py tests/test_zz_gpu_threads.py tests/test_zz_gpu_graph.py -m gpu -q > $O/new_tests.txt 2>&1; echo "new tests rc=$?"; tail -15 $O/new_tests.txt | cut -c1-300
timeout 1200 python -m pytest tests/test_zz_gpu_fullconfig.py -m gpu -q -k "c4 or coarse_quantiser_trained" --durations=5 > $O/fullconfig_new.txt 2>&1; echo "fullconfig c4/c5t rc=$?"; tail -15 $O/fullconfig_new.txt | cut -c1-300
LANCE_TEST_UNVALIDATED=1 timeout 600 python -m pytest tests/test_zz_gpu_qpt_variant.py -m gpu -q -x > $O/qpt_test.txt 2>&1; echo "qpt test rc=$?"; tail -5 $O/qpt_test.txt | cut -c1-300
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_zz_gpu_fullconfig.py::test_c4_f16_rows_hierarchical_nlist_4096_one_million_rows --deselect tests/test_zz_gpu_fullconfig.py::test_c5_coarse_quantiser_trained_to_65536_by_the_engine --deselect tests/test_zz_gpu_coarse_mfma.py --deselect tests/test_zz_gpu_threads.py --deselect tests/test_zz_gpu_graph.py > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -4 $O/gpu_suite.txt | cut -c1-300
timeout 300 python scripts/probe_c3_search.py > $O/c3_default.json 2> $O/c3_default.err; tail -1 $O/c3_default.json | cut -c1-1800
LANCE_HIP_QPT=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt.json 2> $O/c3_qpt.err; tail -1 $O/c3_qpt.json | cut -c1-1800
LANCE_HIP_QPT=1 LANCE_HIP_GRAPH=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt_graph.json 2> $O/c3_qpt_graph.err; tail -1 $O/c3_qpt_graph.json | cut -c1-600
LANCE_HIP_MFMA_COARSE=0 timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_nocoarse.json 2> $O/bench_nocoarse.err; tail -1 $O/bench_nocoarse.json | cut -c1-1500
( time timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_time.txt; tail -1 $O/bench_n1.json | cut -c1-3000; tail -3 $O/bench_time.txt; tail -3 $O/bench_n1.err
