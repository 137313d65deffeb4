# GPU call r04e: MFMA table build in the C2 filter scan (first run), graphs on by default, QPT prep timers
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_coarse_mfma.py -m gpu -q -x --timeout 600 > $O/pm_tests.txt 2>&1; echo "pm tests rc=$?"; grep -E "passed|failed|Error|assert" $O/pm_tests.txt | cut -c1-300 | tail -12
(timeout 150 python tests/fuzz_parity.py 100 501 --log $O/fuzz_501.txt > /dev/null 2>&1 &
 timeout 150 python tests/fuzz_parity.py 100 502 --log $O/fuzz_502.txt > /dev/null 2>&1 &
 wait)
grep -hE "SKIP|MISMATCH|ERROR|fuzz " $O/fuzz_501.txt $O/fuzz_502.txt | cut -c1-300 | tail -12
timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_mb.json 2> $O/bench_mb.err; tail -1 $O/bench_mb.json | cut -c1-1500; tail -2 $O/bench_mb.err
LANCE_HIP_NO_MFMA_TABLE=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_nomb.json 2> $O/bench_nomb.err; tail -1 $O/bench_nomb.json | cut -c1-1500
LANCE_HIP_QPT=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt.json 2> $O/c3_qpt.err; tail -1 $O/c3_qpt.json | cut -c1-1300
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_pm_scan.py --deselect tests/test_zz_gpu_coarse_mfma.py > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.txt | cut -c1-300
