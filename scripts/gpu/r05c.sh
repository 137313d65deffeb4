# GPU call r05c: scan with helper workgroups (chunks from per-slice global counters), merge without the sort when refine follows,
# refine with two lanes per row: parity, then A/B by switch on the bench, then the scan's phase stamps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_graph.py tests/test_zz_gpu_fuzz_findings.py tests/test_gpu_parity.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_f16_metrics.py tests/test_zz_gpu_threads.py -m gpu -q --timeout 900 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -12 $O/tests.txt | cut -c1-400
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), j['ms_per_step'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])" | cut -c1-600; }
B="--steps 200 --warmup 10 --no-pmc --no-cpu-baseline"
timeout 300 python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo "rc=$?"; brief $O/bench_new.json NEW
LANCE_HIP_MS_HELP=0 timeout 300 python bench.py $B > $O/bench_nohelp.json 2> $O/bench_nohelp.err; brief $O/bench_nohelp.json NOHELP
LANCE_HIP_MS_HELP=4 timeout 300 python bench.py $B > $O/bench_help4.json 2> $O/bench_help4.err; brief $O/bench_help4.json HELP4
LANCE_HIP_MS_HELP=16 timeout 300 python bench.py $B > $O/bench_help16.json 2> $O/bench_help16.err; brief $O/bench_help16.json HELP16
LANCE_HIP_MS_RS=1024 timeout 300 python bench.py $B > $O/bench_rs1024.json 2> $O/bench_rs1024.err; brief $O/bench_rs1024.json RS1024
LANCE_HIP_MS_RS=4096 timeout 300 python bench.py $B > $O/bench_rs4096.json 2> $O/bench_rs4096.err; brief $O/bench_rs4096.json RS4096
LANCE_HIP_REFINE_V1=1 timeout 300 python bench.py $B > $O/bench_refine_v1.json 2> $O/bench_refine_v1.err; brief $O/bench_refine_v1.json REFINE_V1
timeout 300 python bench.py $B --streams 1 > $O/bench_new_s1.json 2> $O/bench_new_s1.err; brief $O/bench_new_s1.json NEW_S1
LANCE_HIP_GRAPH=0 LANCE_HIP_MS_PROF=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 3 --warmup 1 --streams 1 > $O/prof.json 2> $O/prof.err; grep "ms prof" $O/prof.err | tail -4 | cut -c1-1500
LANCE_HIP_MS_HELP=0 LANCE_HIP_GRAPH=0 LANCE_HIP_MS_PROF=1 timeout 200 python bench.py --no-pmc --no-cpu-baseline --steps 3 --warmup 1 --streams 1 > $O/prof_nohelp.json 2> $O/prof_nohelp.err; grep "ms prof" $O/prof_nohelp.err | tail -2 | cut -c1-700
timeout 200 python tests/fuzz_parity.py 60 5202 --log $O/fuzz.txt > $O/fuzz.out 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.out | cut -c1-300
