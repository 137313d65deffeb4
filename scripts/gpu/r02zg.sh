# GPU call r02zg: IVF_FLAT with more ties than the pool holds (exact replay instead of "did not converge")
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02zg
timeout 40 python -m pytest tests/test_zz_gpu_zz_ivfflat_ties.py "tests/test_gpu_parity.py::test_prefilter_matches_reference_branch" -m gpu -q --timeout 30 > gpurun_out/r02zg/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^E  |Error" gpurun_out/r02zg/pytest.log | cut -c1-300 | tail -12
