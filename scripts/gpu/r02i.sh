# GPU call r02i: fused prefilter mask, adaptive probing, device-resident sharded Lloyd loop, merge kernel
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log | cut -c1-400
run() { # name, env...
  name=$1; shift
  env "$@" timeout 400 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['build_sec'], r['build_stages_ms'], r['multi_gpu'])" || tail -15 $O/bench_$name.err
}
run n1 A=1
run dist1 LANCE_BENCH_FORCE_DIST=1
