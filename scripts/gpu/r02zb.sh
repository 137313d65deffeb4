# GPU call r02zb: f16 dot / cosine (new), fused IVF_FLAT prefilter + prefilter-with-range (new), nlist = 65,536 assign probe
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02zb; mkdir -p $O
cd $R
timeout 170 python -m pytest tests/test_zz_gpu_f16_metrics.py "tests/test_gpu_parity.py::test_prefilter_matches_reference_branch" "tests/test_gpu_parity.py::test_4bit_pq_bit_exact" "tests/test_gpu_parity.py::test_distance_range_search" -m gpu -q --timeout 120 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert|^FAILED|^ERROR" $O/pytest.log | cut -c1-260 | tail -40
timeout 60 python scripts/probe_assign_k65536.py > $O/assign_k65536.txt 2>&1; tail -4 $O/assign_k65536.txt | cut -c1-300
