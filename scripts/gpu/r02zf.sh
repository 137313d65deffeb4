# GPU call r02zf (final validation of the round-2 tree): the whole -m gpu suite, the C3 grid with the K-tiled MFMA assign,
# and a kernel trace of the C3-shaped assign
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02zf; mkdir -p $O
cd $R
timeout 240 python -m pytest tests -m gpu -x -q --timeout 150 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-300
timeout 45 python tests/fuzz_parity.py 22 11 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log | cut -c1-400
timeout 60 python bench.py --no-cpu-baseline > $O/bench_nocpu.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-330 $O/bench_nocpu.json
timeout 100 python scripts/measure_grid.py --c3 --skip-c2 > $O/grid_c3.json 2> $O/grid_c3.err; echo "grid rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/scripts/probe_assign_wide.py mfma > $O/prof.log 2>&1; echo "prof rc=$?"
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/assign_wide_kernel_stats.csv && head -8 $O/assign_wide_kernel_stats.csv | cut -c1-200
rm -rf $O/prof
