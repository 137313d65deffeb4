# GPU call r05f: round-5 scan kernel (codebook in LDS, pair blocks of 256, whole-partition slices with LDS survivor counters): parity, A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests/test_zz_gpu_mscan.py tests/test_zz_gpu_msbound.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_graph.py tests/test_gpu_parity.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_fuzz_findings.py -m gpu -q --timeout 1200 -x > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -12 $O/tests.txt | cut -c1-300
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['exact_replays_last_step'], j['kernel_ms_per_step'])" | cut -c1-700; }
B="--steps 200 --warmup 10 --no-pmc --no-cpu-baseline --no-grid"
timeout 300 python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo "rc=$?"; brief $O/bench_new.json V5LOCAL; tail -2 $O/bench_new.err | cut -c1-300
LANCE_HIP_MS_LOCAL=0 timeout 300 python bench.py $B > $O/bench_rows.json 2> $O/bench_rows.err; brief $O/bench_rows.json V5ROWS
LANCE_HIP_MS_V4=1 timeout 300 python bench.py $B > $O/bench_v4.json 2> $O/bench_v4.err; brief $O/bench_v4.json V4
timeout 300 python bench.py $B --streams 1 > $O/bench_s1.json 2> $O/bench_s1.err; brief $O/bench_s1.json V5LOCAL_S1
timeout 300 python bench.py $B --streams 2 > $O/bench_s2.json 2> $O/bench_s2.err; brief $O/bench_s2.json V5LOCAL_S2
timeout 300 python tests/fuzz_parity.py 100 5204 --log $O/fuzz.txt > $O/fuzz.out 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.out | cut -c1-300
