# GPU call r05x: the k-means loop's group-by in one launch (group_small_kernel): parity, build time, QPS
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05x; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_fullconfig.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_mscan.py tests/test_zz_gpu_msbound.py tests/test_abi.py tests/test_zz_gpu_two_ranks.py -m gpu -q --timeout 900 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | cut -c1-300 | tail -8; grep -E "^E  " $O/tests.txt | head -8 | cut -c1-300
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), j['ms_per_step'], 'build', j['build_sec'], j['roofline_build']['build_stages_ms'], j['kernel_ms_per_step'])" | cut -c1-700; }
B="--steps 200 --warmup 10 --no-pmc --no-cpu-baseline --no-grid"
timeout 300 python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo "rc=$?"; brief $O/bench_new.json NEW
timeout 300 python bench.py $B > $O/bench_new2.json 2> $O/bench_new2.err; brief $O/bench_new2.json NEW2
timeout 120 python scripts/probe_build_steps.py 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('pq_train', j['pq: pq_train'], j['pq_train kernel ms by stage'], 'ivf', j['ivf: kmeans_train'], j['ivf: iterations'])"
timeout 200 python tests/fuzz_parity.py 60 5209 --log $O/fuzz.txt > $O/fuzz.out 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.out | cut -c1-300
