# GPU call r04t: the cleaned-up tree (one matrix-core scan kernel): mscan + pm-scan parity, fuzz, bench with the profile stamps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04t; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 300 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_graph.py tests/test_zz_gpu_threads.py -m gpu -q --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -2 $O/tests.txt | cut -c1-300
B="python bench.py --no-pmc --no-cpu-baseline"
timeout 200 $B > $O/b0.json 2>$O/b0.err; python -c "
import json; j=json.loads(open('$O/b0.json').read().strip().splitlines()[-1]); print('DEF', j['value'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'], j['roofline']['frac'])"
LANCE_HIP_GRAPH=0 LANCE_HIP_MS_PROF=1 timeout 200 $B --steps 5 --warmup 2 --streams 1 2>&1 | grep "ms prof" | tail -1
t0=$(date +%s)
timeout 200 python tests/fuzz_parity.py 60 4808 --log $O/fuzz_4808.txt > $O/fuzz_4808.out 2>&1; echo "fuzz rc=$? $(( $(date +%s)-t0 )) s"; tail -1 $O/fuzz_4808.out | cut -c1-300
