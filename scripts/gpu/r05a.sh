# GPU call r05a: the round-5 tree after the CPU-side work (parity tests on the headline path, graph cache rework, comm hook, housekeeping):
# full -m gpu suite, bench line (200 steps), then the survivor-records variant of the scan (A/B: tests + bench)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05a; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; tail -15 $O/gpu_suite.txt | cut -c1-400
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['exact_replays_last_step'], j['kernel_ms_per_step'])" | cut -c1-600; }
timeout 300 python bench.py --steps 200 --warmup 10 --no-pmc --no-cpu-baseline > $O/bench_main.json 2> $O/bench_main.err; echo "bench rc=$?"; brief $O/bench_main.json MAIN
timeout 300 python bench.py --steps 100 --warmup 10 --no-pmc --no-cpu-baseline --streams 1 > $O/bench_main_s1.json 2> $O/bench_main_s1.err; brief $O/bench_main_s1.json MAIN_S1
export LANCE_HIP_LIB=$R/build/variants/liblance_hip_rec.so
t0=$(date +%s)
timeout 600 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q -x --timeout 600 > $O/rec_tests.txt 2>&1; echo "rec tests rc=$? $(( $(date +%s)-t0 )) s"; tail -3 $O/rec_tests.txt | cut -c1-300
timeout 300 python bench.py --steps 200 --warmup 10 --no-pmc --no-cpu-baseline > $O/bench_rec.json 2> $O/bench_rec.err; echo "bench rc=$?"; brief $O/bench_rec.json REC
timeout 300 python bench.py --steps 100 --warmup 10 --no-pmc --no-cpu-baseline --streams 1 > $O/bench_rec_s1.json 2> $O/bench_rec_s1.err; brief $O/bench_rec_s1.json REC_S1
