# GPU call r03d: the divergent-barrier fix -- fuzz replays of every failing case of r03a, the new named tests, the full suite
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03d; mkdir -p $O
timeout 600 python -m pytest tests/test_zz_gpu_fuzz_findings.py -m gpu -q --timeout 300 > $O/pytest_findings.log 2>&1; echo "findings rc=$?"; grep -E "passed|failed|^FAILED|^E  " $O/pytest_findings.log | cut -c1-300 | tail -8
for sc in "11 4" "11 39" "11 59" "13 4" "13 46" "13 49" "13 63" "13 77" "13 86" "12 61" "13 0"; do set -- $sc; for i in 1 2; do timeout 300 python tests/fuzz_parity.py 200 $1 --case $2 --debug > $O/fuzz_$1_$2_run$i.txt 2>&1; grep -E "MISMATCH|fuzz |ERROR" $O/fuzz_$1_$2_run$i.txt | cut -c1-300; done; done
timeout 900 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_all.log | cut -c1-300 | tail -10
