# GPU call r05zb: long-row flat filter with the margin of a bf16 PRODUCT's roundoff (0.84 %; it carried one operand's, 0.45 %): the adversarial
# case that needs it, the timing it costs, then the driver's own suite command on the final tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05zb; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_flat_wide.py -m gpu -q --timeout 500 > $O/tests_fw.txt 2>&1; echo "flat wide tests rc=$?"; tail -1 $O/tests_fw.txt; grep -E "^E  |^FAILED" $O/tests_fw.txt | head -6 | cut -c1-300
timeout 300 python scripts/probe_flat_wide.py > $O/probe.txt 2>&1; grep -v amdgpu $O/probe.txt | grep -v exact_only.:.true | cut -c1-300
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -6
