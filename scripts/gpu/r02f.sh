# GPU call r02f: overflow flagging fix, MFMA assign top-4, ADVICE fixes; full suite + bench with cpu baseline + profiles
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python scripts/probe_assign.py > $O/assign.log 2>&1; grep -v amdgpu.ids $O/assign.log | cut -c1-400
timeout 600 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 20 --no-cpu-baseline > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -30 {} | cut -c1-200
rm -rf $O/prof/*/*.db $O/prof/*/*kernel_trace.csv 2>/dev/null
du -sh $O
