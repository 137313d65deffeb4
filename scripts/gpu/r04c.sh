# GPU call r04c: round-4 batch 2 -- coarse select fix, graph diagnosis, flat_small, lifted limits, comm API, C3 async timing
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04c; mkdir -p $O
timeout 1200 python -m pytest tests/test_zz_gpu_coarse_mfma.py tests/test_zz_gpu_graph.py tests/test_zz_gpu_flat_small.py tests/test_zz_gpu_limits.py tests/test_zz_gpu_comm.py tests/test_zz_gpu_threads.py -m gpu -q --timeout 900 > $O/new_tests.txt 2>&1; echo "new tests rc=$?"; grep -E "passed|failed|Error|differ" $O/new_tests.txt | cut -c1-400 | tail -30
timeout 120 python scripts/probe_flat_small.py > $O/flat_small.json 2> $O/flat_small.err; tail -1 $O/flat_small.json | cut -c1-900
LANCE_HIP_NO_FLAT_SMALL=1 timeout 120 python scripts/probe_flat_small.py > $O/flat_batch.json 2> $O/flat_batch.err; tail -1 $O/flat_batch.json | cut -c1-900
timeout 300 python scripts/probe_c3_search.py > $O/c3_default.json 2> $O/c3_default.err; tail -1 $O/c3_default.json | cut -c1-2000
LANCE_HIP_QPT=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt.json 2> $O/c3_qpt.err; tail -1 $O/c3_qpt.json | cut -c1-2000
LANCE_HIP_QPT=1 LANCE_HIP_GRAPH=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt_graph.json 2> $O/c3_qpt_graph.err; tail -1 $O/c3_qpt_graph.json | cut -c1-2000
LANCE_HIP_MFMA_COARSE=1 LANCE_HIP_QPT=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt_coarse.json 2> $O/c3_qpt_coarse.err; tail -1 $O/c3_qpt_coarse.json | cut -c1-1200
