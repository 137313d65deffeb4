# GPU call r06zzm: C1 single-query flat: wall time + kernel trace (where do the 0.172 ms go)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzm; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one -- python $R/scripts/probe_flat_one.py > $R/$O/probe_prof.txt 2>&1)
f=$(find /tmp/prof_one -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; head -12 $O/kernel_stats.csv | cut -c1-200; fi
