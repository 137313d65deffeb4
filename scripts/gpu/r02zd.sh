# GPU call r02zd: C3 shape (1M x 1536 cosine, IVF1024, PQ96) under rocprofv3 --kernel-trace --stats (cold + warm build, flat scan,
# search grid) with the tiled normalize kernel; normalize parity cases
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02zd; mkdir -p $O
cd $R
timeout 60 python -m pytest "tests/test_zz_gpu_f16_metrics.py::test_f16_normalize_half_precision" "tests/test_gpu_parity.py::test_ivfpq_encode_matches_transform_chain" -m gpu -q --timeout 50 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof -o c3 -- python $R/scripts/measure_grid.py --c3 --skip-c2 > $O/grid_c3.json 2> $O/grid_c3.err; echo "grid rc=$?"
ls -R $O/prof | head -20
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/c3_kernel_stats.csv && head -12 $O/c3_kernel_stats.csv | cut -c1-160
find $O/prof -type f ! -name '*stats.csv' -delete
