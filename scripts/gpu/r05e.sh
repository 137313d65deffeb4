# GPU call r05e: the bound pass on the matrix cores (ms_bound_kernel) + the leaner u8 refine (52 VGPRs, ranked top-k instead of the sort):
# parity first, then A/B by switch on the bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_zz_gpu_msbound.py tests/test_zz_gpu_refine_u8.py tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_graph.py tests/test_gpu_parity.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_fuzz_findings.py tests/test_zz_gpu_f16_metrics.py -m gpu -q --timeout 900 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -12 $O/tests.txt | cut -c1-300
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['exact_replays_last_step'], j['kernel_ms_per_step'])" | cut -c1-700; }
B="--steps 200 --warmup 10 --no-pmc --no-cpu-baseline --no-grid"
timeout 300 python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo "rc=$?"; brief $O/bench_new.json NEW; tail -2 $O/bench_new.err | cut -c1-300
LANCE_HIP_NO_MSBOUND=1 timeout 300 python bench.py $B > $O/bench_intbound.json 2> $O/bench_intbound.err; brief $O/bench_intbound.json INTBOUND
LANCE_HIP_NO_RAW_COMPACT=1 timeout 300 python bench.py $B > $O/bench_f32.json 2> $O/bench_f32.err; brief $O/bench_f32.json F32REFINE
timeout 300 python bench.py $B --streams 1 > $O/bench_s1.json 2> $O/bench_s1.err; brief $O/bench_s1.json S1
LANCE_HIP_Q_STATS=1 LANCE_HIP_GRAPH=0 timeout 200 python bench.py --no-pmc --no-cpu-baseline --no-grid --steps 2 --warmup 1 --streams 1 > $O/stats_new.json 2> $O/stats_new.err; grep "qscan" $O/stats_new.err | tail -2 | cut -c1-300
LANCE_HIP_NO_MSBOUND=1 LANCE_HIP_Q_STATS=1 LANCE_HIP_GRAPH=0 timeout 200 python bench.py --no-pmc --no-cpu-baseline --no-grid --steps 2 --warmup 1 --streams 1 > $O/stats_int.json 2> $O/stats_int.err; grep "qscan" $O/stats_int.err | tail -2 | cut -c1-300
timeout 200 python tests/fuzz_parity.py 60 5203 --log $O/fuzz.txt > $O/fuzz.out 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.out | cut -c1-300
