# GPU call r06zze: bound pass over the two nearest lists (dot), absolute 2^-22 margin of the dot assign / flat filters: parity, fuzz, rates
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zze; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tests/fuzz_dot_flow.py 1 7001 --case 25 2>&1 | grep -v amdgpu.ids | cut -c1-400
timeout 900 python -m pytest tests/test_zz_gpu_dot_flow.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_msbound.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt | cut -c1-300
timeout 1200 python tests/fuzz_dot_flow.py 900 7002 > $O/fuzz_dot.txt 2>&1; echo "dot fuzz rc=$?"; grep -v amdgpu.ids $O/fuzz_dot.txt | tail -6 | cut -c1-500
timeout 900 python scripts/probe_dot_flow.py child > $O/dot_probe.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/dot_probe.txt | grep -A1 " dot" | cut -c1-600
LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_dot_flow.py child 2>&1 | grep "qscan\]" | uniq -c | head -12 | cut -c1-400
