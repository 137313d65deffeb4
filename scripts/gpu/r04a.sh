# GPU call r04a: first run ever of the per-query-table filter (LANCE_HIP_QPT=1): parity test, fuzz, C3 timing with / without it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04a; mkdir -p $O
LANCE_TEST_UNVALIDATED=1 timeout 900 python -m pytest tests/test_zz_gpu_qpt_variant.py -m gpu -q -x > $O/qpt_test.txt 2>&1; echo "qpt test rc=$?"; tail -25 $O/qpt_test.txt | cut -c1-400
(LANCE_HIP_QPT=1 timeout 200 python tests/fuzz_parity.py 150 401 --log $O/fuzz_qpt_401.txt > /dev/null 2>&1 &
 LANCE_HIP_QPT=1 timeout 200 python tests/fuzz_parity.py 150 402 --log $O/fuzz_qpt_402.txt > /dev/null 2>&1 &
 wait)
grep -hE "SKIP|MISMATCH|ERROR|fuzz " $O/fuzz_qpt_401.txt $O/fuzz_qpt_402.txt | cut -c1-300 | tail -20
timeout 300 python scripts/probe_c3_search.py > $O/c3_default.json 2> $O/c3_default.err; tail -1 $O/c3_default.json | cut -c1-1500
LANCE_HIP_QPT=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt.json 2> $O/c3_qpt.err; tail -1 $O/c3_qpt.json | cut -c1-1500; tail -3 $O/c3_qpt.err
