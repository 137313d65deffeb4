# GPU call r05b: scan with the wait-count fixes + accumulator started at -limit, merge kernel with the short dependent chain: parity
# (scan / merge / graph / two-rank / full-size tests), then A/B on the bench: new tree | old merge (env) | old scan (variant) | 1 stream
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05b; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_graph.py tests/test_zz_gpu_two_ranks.py tests/test_zz_gpu_fuzz_findings.py "tests/test_gpu_parity.py::test_full_size_properties_sift1m" tests/test_zz_gpu_fullconfig.py -m gpu -q --timeout 900 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -12 $O/tests.txt | cut -c1-400
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), j['ms_per_step'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])" | cut -c1-600; }
B="--steps 200 --warmup 10 --no-pmc --no-cpu-baseline"
timeout 300 python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo "rc=$?"; brief $O/bench_new.json NEW
timeout 300 python bench.py $B --streams 1 > $O/bench_new_s1.json 2> $O/bench_new_s1.err; brief $O/bench_new_s1.json NEW_S1
LANCE_HIP_QMERGE_V1=1 timeout 300 python bench.py $B > $O/bench_merge_v1.json 2> $O/bench_merge_v1.err; brief $O/bench_merge_v1.json MERGE_V1
LANCE_HIP_LIB=$R/build/variants/liblance_hip_oldscan.so timeout 300 python bench.py $B > $O/bench_oldscan.json 2> $O/bench_oldscan.err; brief $O/bench_oldscan.json OLDSCAN
timeout 300 python bench.py $B --streams 4 > $O/bench_new_s4.json 2> $O/bench_new_s4.err; brief $O/bench_new_s4.json NEW_S4
timeout 200 python tests/fuzz_parity.py 60 5201 --log $O/fuzz.txt > $O/fuzz.out 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.out | cut -c1-300
