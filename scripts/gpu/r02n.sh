set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02n; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_pm_scan.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log | cut -c1-300
run() { # name, args...
  name=$1; shift
  timeout 400 python bench.py --steps 40 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['kernel_ms_per_step'])" || tail -15 $O/bench_$name.err
}
run s1 --streams 1
run s2 --streams 2
run s3 --streams 3
run s4 --streams 4
