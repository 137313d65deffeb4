# GPU call r02b: parity of the quantised flow + A/B of its variants
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['kernel_ms_per_step'])" || tail -5 $O/bench_$name.err
}
run qw8 A=1
run qw6 LANCE_HIP_LIB=$R/build/variants/liblance_hip_qw6.so
run old LANCE_HIP_NO_QSCAN=1
run qw8_p4 LANCE_HIP_QSCAN_WGS=4
run qw8_p8 LANCE_HIP_QSCAN_WGS=8
run qw6_p3 LANCE_HIP_LIB=$R/build/variants/liblance_hip_qw6.so LANCE_HIP_QSCAN_WGS=3
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 20 --no-cpu-baseline > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -25 {}
rm -rf $O/prof/*/*.db $O/prof/*/*kernel_trace.csv 2>/dev/null
du -sh $O
