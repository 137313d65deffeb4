# GPU call r06zzo: flat_small: vectorised merge passes; A/B of workgroups per CU and of the row unroll
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzo; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_zz_gpu_flat_small.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt | cut -c1-200
echo base; timeout 300 python scripts/probe_flat_one.py 2>/dev/null | head -2
for w in 6 8; do echo "wgs/cu $w"; LANCE_HIP_FS_WGS_PER_CU=$w timeout 300 python scripts/probe_flat_one.py 2>/dev/null | head -2; done
for v in fsu4 fsu8; do for w in 4 8; do echo "$v wgs/cu $w"; LANCE_HIP_LIB=$R/build/variants/liblance_hip_$v.so LANCE_HIP_FS_WGS_PER_CU=$w timeout 300 python scripts/probe_flat_one.py 2>/dev/null | head -2; done; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_one -- python $R/scripts/probe_flat_one.py > $R/$O/probe_prof.txt 2>&1)
f=$(find /tmp/prof_one -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; grep flat_small $O/kernel_stats.csv | cut -c1-200; fi
