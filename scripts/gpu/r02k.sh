# GPU call r02k: native f16/int8 rows (MFMA assign + fused encode), sharded loop init fix, C4-shaped probe
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02k; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log | cut -c1-300
timeout 300 python scripts/probe_sharded_kmeans.py 2>&1 | grep -E "sharded|estep|update" | tee $O/sharded_kmeans.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 400 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['build_sec'], r['build_stages_ms'], r['multi_gpu'])" || tail -15 $O/bench_$name.err
}
run n1 A=1
run nofused LANCE_HIP_NO_FUSED_ENCODE=1
run dist1 LANCE_BENCH_FORCE_DIST=1
timeout 900 python scripts/scale_probe.py 20000000 f16 > $O/scale_f16_20M.txt 2>&1; tail -12 $O/scale_f16_20M.txt
