# GPU call r02e: device-resident Lloyd loop, qmerge BS=128, MFMA assign in the E-step; kernel-level profile of assign
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02e; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['build_sec'], r['build_stages_ms'], r['kernel_ms_per_step'], r['roofline']['frac'])" || tail -5 $O/bench_$name.err
}
run q128 LANCE_HIP_Q_STATS=1
grep qscan $O/bench_q128.err | tail -3
run q256 LANCE_HIP_QMERGE_BS=256
run q128mpf4 LANCE_HIP_LIB=$R/build/variants/liblance_hip_mpf4.so
run q128nomfma LANCE_HIP_NO_MFMA=1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_assign -- python $R/scripts/probe_assign.py mfma > $O/prof_assign.log 2>&1
find $O/prof_assign -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
rm -rf $O/prof_assign/*/*.db $O/prof_assign/*/*kernel_trace.csv 2>/dev/null
du -sh $O
