# GPU call r03q: kernel trace of the C3 probe (which kernel inside the "merge" timer costs what)
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/scripts/probe_c3_search.py > $R/$O/probe.json 2> $R/$O/probe.err
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/c3_kernel_stats.csv; head -30 $O/c3_kernel_stats.csv | cut -c1-170
rm -rf $O/prof
