# GPU call r03s: full GPU suite on the tree with the 1024-lane rescan + analytic bound scale; kernel trace of the C3 probe
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03s; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 400 > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/scripts/probe_c3_search.py > $R/$O/probe.json 2> $R/$O/probe.err
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/c3_kernel_stats.csv; grep -E "qrescan|qmerge|qbound|qscan_tiled|refine_kernel|select_probes|group_|q_item|q_tclass|q_nearest|q_residual" $O/c3_kernel_stats.csv | cut -c1-170
rm -rf $O/prof
