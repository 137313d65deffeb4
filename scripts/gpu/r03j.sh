# GPU call r03j: kernel trace of the bench (build + search) with the PQ MFMA E-step; PQ parity tests
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03j; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_fullconfig.py -m gpu -q --timeout 300 -k "pq or kmeans or c3 or c5 or f16" > $O/pytest_pq.log 2>&1; echo "pq tests rc=$?"; tail -3 $O/pytest_pq.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/bench_kernel_stats.csv; head -32 $O/bench_kernel_stats.csv | cut -c1-170; tail -1 $O/prof.log | cut -c1-600; rm -rf $O/prof
