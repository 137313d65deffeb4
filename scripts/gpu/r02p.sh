# GPU call r02p: full suite (with durations) on the new defaults, merge-kernel register variants
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O
cd $R
nproc
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -16 $O/pytest.log | cut -c1-200
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['kernel_ms_per_step'], r['roofline']['frac'])" || tail -15 $O/bench_$name.err
}
run base A=1
run qm6_4 LANCE_HIP_LIB=$R/build/variants/liblance_hip_qm6_4.so
run qm8_2 LANCE_HIP_LIB=$R/build/variants/liblance_hip_qm8_2.so
