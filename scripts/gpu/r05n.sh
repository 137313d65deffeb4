# GPU call r05o: long-row flat filter with the XCD-aware tile order (parity, timing with 2 and 4 row blocks per wave)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05n; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_zz_gpu_flat_wide.py -m gpu -q --timeout 800 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | cut -c1-300 | tail -8; grep -E "^E  " $O/tests.txt | head -8 | cut -c1-300
LANCE_HIP_FW_BR=2 timeout 900 python -m pytest tests/test_zz_gpu_flat_wide.py -m gpu -q --timeout 800 -k "long_rows or ties" > $O/tests_br2.txt 2>&1; echo "tests br2 rc=$?"; tail -1 $O/tests_br2.txt
timeout 400 python scripts/probe_flat_wide.py > $O/probe_br4.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu $O/probe_br4.txt | grep -v exact_only.:.true | cut -c1-330
LANCE_HIP_FW_BR=2 timeout 400 python scripts/probe_flat_wide.py > $O/probe_br2.txt 2>&1; grep -v amdgpu $O/probe_br2.txt | grep -v exact_only.:.true | cut -c1-330
timeout 500 python scripts/probe_c3_search.py > $O/c3.json 2> $O/c3.err; echo "c3 rc=$?"; tail -c 1800 $O/c3.json
