# GPU call r06zu: dot flow on the centred codebook plane: parity, then where a C2-shaped dot batch spends its time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zu; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_dot_flow.py -x -q -m gpu --timeout 600 > $O/dot_flow.txt 2>&1; echo "dot flow rc=$?"; tail -5 $O/dot_flow.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_msbound.py tests/test_zz_gpu_mscan.py tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "dot or msbound or mscan or prefilter" > $O/related.txt 2>&1; echo "related rc=$?"; tail -3 $O/related.txt | cut -c1-400
timeout 900 python scripts/probe_dot_flow.py child > $O/dot_probe.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/dot_probe.txt | cut -c1-600
LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_dot_flow.py child 2>&1 | grep "qscan\]" | sort | uniq -c | sort -rn | head -8 | cut -c1-400
