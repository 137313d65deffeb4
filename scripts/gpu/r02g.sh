# GPU call r02g: MFMA PQ assign (encode + training E-step), control kernel in LDS
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log | cut -c1-300
timeout 300 python scripts/probe_assign.py > $O/assign.log 2>&1; grep -v amdgpu.ids $O/assign.log | cut -c1-700
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['build_sec'], r['build_stages_ms'])" || tail -5 $O/bench_$name.err
}
run mfma A=1
run nomfmapq LANCE_HIP_NO_MFMA_PQ=1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 5 --no-cpu-baseline > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -16 {} | cut -c1-160
rm -rf $O/prof/*/*.db $O/prof/*/*kernel_trace.csv 2>/dev/null
