# GPU call r02d: qmerge restructure + MFMA assign first contact
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02d; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 300 python scripts/probe_assign.py > $O/assign.log 2>&1; cat $O/assign.log | cut -c1-600
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['build_sec'], r['build_stages_ms'], r['kernel_ms_per_step'])" || tail -5 $O/bench_$name.err
}
run q64 A=1
run q256 LANCE_HIP_QMERGE_BS=256
run q64mpf4 LANCE_HIP_LIB=$R/build/variants/liblance_hip_mpf4.so
run q64nomfma LANCE_HIP_NO_MFMA=1
du -sh $O
