set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02u; mkdir -p $O
cd $R
timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -x -v -k "ivf_flat" --timeout 60 --durations=0 > $O/pytest_flat.log 2>&1; echo "pytest rc=$?"; grep -E "PASSED|FAILED|Timeout|rror|^tests|File|line " $O/pytest_flat.log | head -60
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "4bit or prefilter" --timeout 120 > $O/pytest_4bit.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_4bit.log
