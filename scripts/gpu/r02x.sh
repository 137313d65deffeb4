# GPU call r02x: merge kernel occupancy variants (codebook prefetch depth x waves per SIMD)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02x; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['kernel_ms_per_step'])" || tail -5 $O/bench_$name.err; }
run base A=1
run m4w6 LANCE_HIP_LIB=$R/build/variants/liblance_hip_m4w6.so
run m2w8 LANCE_HIP_LIB=$R/build/variants/liblance_hip_m2w8.so
run m4w8 LANCE_HIP_LIB=$R/build/variants/liblance_hip_m4w8.so
timeout 120 python -m pytest tests/test_gpu_pm_scan.py -m gpu -x -q --timeout 100 > $O/pytest_pm.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_pm.log
