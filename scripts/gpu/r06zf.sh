# GPU call r06zf: per-group keys from the transform kernel's sweep (xf_kernel MODE 2): parity + the probe (A/B: LANCE_HIP_COARSE_GROUPS_MA=1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zf; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_zz_gpu_coarse_mfma.py -x -q -m gpu --timeout 1400 > $O/tests_coarse.txt 2>&1; echo "coarse tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/tests_coarse.txt | cut -c1-600 | tail -12
timeout 900 python scripts/probe_coarse.py > $O/probe.txt 2>&1; echo "probe rc=$?"; grep -E "^nlist" $O/probe.txt | cut -c1-300
LANCE_HIP_COARSE_GROUPS_MA=1 timeout 300 python scripts/probe_coarse.py 65536 10000 10 int8 2>&1 | grep "^nlist" | cut -c1-300
timeout 900 python -m pytest tests/test_zz_gpu_fullconfig.py tests/test_gpu_parity.py -x -q -m gpu --timeout 900 > $O/tests_more.txt 2>&1; echo "more tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_more.txt | cut -c1-300 | tail -6
