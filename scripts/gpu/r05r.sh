# GPU call r05r: long-row flat filter with 128-element stages (parity, timing against 64)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05r; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_zz_gpu_flat_wide.py -m gpu -q --timeout 800 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | cut -c1-300 | tail -8; grep -E "^E  " $O/tests.txt | head -8 | cut -c1-300
timeout 400 python scripts/probe_flat_wide.py > $O/probe_kt128.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu $O/probe_kt128.txt | grep -v exact_only.:.true | cut -c1-330
LANCE_HIP_FW_KT=64 timeout 400 python scripts/probe_flat_wide.py > $O/probe_kt64.txt 2>&1; grep -v amdgpu $O/probe_kt64.txt | grep -v exact_only.:.true | cut -c1-330
