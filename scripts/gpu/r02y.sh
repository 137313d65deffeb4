# GPU call r02y: refine kernel with 16 lanes per candidate row; full suite + fuzz on the final defaults
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02y; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['kernel_ms_per_step'])" || tail -5 $O/bench_$name.err; }
run refine16 A=1
timeout 200 python -m pytest tests -m gpu -x -q --timeout 150 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 70 python tests/fuzz_parity.py 40 41 > $O/fuzz_41.log 2>&1; tail -2 $O/fuzz_41.log
