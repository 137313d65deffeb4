# GPU call r06b: which scale fails (and does the round-5 route fail it too); kernel trace of the C2 transform
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q -k large_values 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-300
LANCE_HIP_NO_XFORM_FUSED=1 XF_ALLOW_OLD=1 timeout 300 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q -k large_values 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-300
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o c2 -- python $GRAFT_REPO_ROOT/scripts/probe_xform.py c2 > $GRAFT_REPO_ROOT/$O/prof_run.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -12 $f | cut -c1-200
python scripts/diff_routes.py > $O/diff_routes.txt 2>&1; tail -30 $O/diff_routes.txt | cut -c1-300
