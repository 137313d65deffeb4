# GPU call r05t: long-row flat filter with one 32-query block per wave (64-query tiles, 104 VGPRs: four waves per SIMD) against two
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05t; mkdir -p $O
export TMPDIR=/tmp
LANCE_HIP_FW_BQ=1 timeout 900 python -m pytest tests/test_zz_gpu_flat_wide.py -m gpu -q --timeout 800 > $O/tests_bq1.txt 2>&1; echo "tests bq1 rc=$?"; tail -1 $O/tests_bq1.txt
timeout 900 python -m pytest tests/test_zz_gpu_flat_wide.py -m gpu -q --timeout 800 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -1 $O/tests.txt
LANCE_HIP_FW_BQ=1 timeout 400 python scripts/probe_flat_wide.py > $O/probe_bq1.txt 2>&1; grep -v amdgpu $O/probe_bq1.txt | grep -v exact_only.:.true | cut -c1-330
timeout 400 python scripts/probe_flat_wide.py > $O/probe_bq2.txt 2>&1; grep -v amdgpu $O/probe_bq2.txt | grep -v exact_only.:.true | cut -c1-330
