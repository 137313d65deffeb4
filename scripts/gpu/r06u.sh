# GPU call r06u: single-f16-product coarse sweep for unit-length long rows: parity, C3 transform timing + kernel breakdown
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06u; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-400
timeout 900 python -m pytest tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_wide_mfma.py tests/test_zz_gpu_f16_metrics.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-400
(cd /tmp && PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/scripts/probe_xform_c3.py > $R/$O/c3_run.txt 2>&1); echo "rocprof rc=$?"
grep C3_XFORM $O/c3_run.txt | cut -c1-300
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/c3_xform_kernel_stats.csv; head -9 $O/c3_xform_kernel_stats.csv | cut -c1-170; fi
