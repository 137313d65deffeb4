# GPU call r06zr: the dot metric's quantised flow (matrix-core bound pass + scan, dot instantiations of the merge / rescan kernels): parity, then the rate
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zr; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_dot_flow.py -x -q -m gpu --timeout 600 > $O/dot_flow.txt 2>&1; echo "dot flow rc=$?"; tail -25 $O/dot_flow.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_msbound.py tests/test_zz_gpu_mscan.py tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "dot or msbound or mscan or prefilter" > $O/related.txt 2>&1; echo "related rc=$?"; tail -8 $O/related.txt | cut -c1-400
timeout 600 python scripts/probe_metrics.py > $O/metrics.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/metrics.txt | cut -c1-300
