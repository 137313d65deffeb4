set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02t; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['kernel_ms_per_step'], r['roofline']['frac'])"; tail -3 $O/bench_$name.err; }
run sload A=1
timeout 400 python -m pytest tests/test_gpu_pm_scan.py -m gpu -x -q -k "(every_instantiation and l2) or two_class or overflow or f16_column or int8_column" > $O/pytest_pm.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_pm.log
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ivf_flat" > $O/pytest_flat.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_flat.log
