# GPU call r06c: underflow guards; kernel trace of the C2 transform; where the two routes disagree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-300
(cd /tmp && PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_xf -- python $R/scripts/probe_xform.py c2 > $R/$O/prof_run.txt 2>&1); echo "rocprof rc=$?"
f=$(find /tmp/prof_xf -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/xform_c2_kernel_stats.csv; head -12 $O/xform_c2_kernel_stats.csv | cut -c1-200; fi
grep -E "^c2" $O/prof_run.txt | cut -c1-500
timeout 600 python scripts/diff_routes.py > $O/diff_routes.txt 2>&1; echo "diff rc=$?"; grep -v amdgpu $O/diff_routes.txt | tail -16 | cut -c1-300
