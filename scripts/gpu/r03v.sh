# GPU call r03v: the loose-bound / overflowed-segment test with the filter's statistics printed (how many segments really overflow)
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03v; mkdir -p $O
LANCE_HIP_Q_STATS=1 timeout 600 python -m pytest tests/test_gpu_pm_scan.py -m gpu -q -x -s -k "loose_bounds" --timeout 400 > $O/loose.txt 2>&1; echo "rc=$?"; grep -E "qscan|passed|failed" $O/loose.txt | cut -c1-220 | head -20
timeout 600 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_fuzz_findings.py -m gpu -q -x --timeout 400 > $O/tests.txt 2>&1; echo "rc=$?"; tail -2 $O/tests.txt | cut -c1-300
