set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['kernel_ms_per_step'], r['roofline'])"; }
run pk A=1
LANCE_HIP_Q_STATS=1 timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -i -E "surviv|stats" | tail -2
timeout 500 python -m pytest tests/test_gpu_pm_scan.py -m gpu -x -q > $O/pytest_pm.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_pm.log
