# GPU call r02ze: K-tiled MFMA assign for rows of more than 128 elements (C3) -- parity cases and the C3-shaped timing on both routes
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02ze; mkdir -p $O
cd $R
timeout 100 python -m pytest tests/test_zz_gpu_wide_mfma.py -m gpu -q --timeout 80 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^E  |^FAILED|Error" $O/pytest.log | cut -c1-300 | tail -30
timeout 80 python scripts/probe_assign_wide.py > $O/assign_wide.txt 2>&1; tail -4 $O/assign_wide.txt | cut -c1-400
