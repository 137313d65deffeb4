# GPU call r06zzn: flat_small merge in two parallel passes + host-flag protocol: parity, wall time, kernel trace
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzn; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_zz_gpu_flat_small.py tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "flat or small" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt | cut -c1-200
timeout 300 python scripts/probe_flat_one.py > $O/probe.txt 2>&1; cat $O/probe.txt
LANCE_HIP_NO_FLAT_HOST_FLAG=1 timeout 300 python scripts/probe_flat_one.py > $O/probe_noflag.txt 2>&1; head -2 $O/probe_noflag.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one -- python $R/scripts/probe_flat_one.py > $R/$O/probe_prof.txt 2>&1)
f=$(find /tmp/prof_one -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; head -8 $O/kernel_stats.csv | cut -c1-200; fi
