# GPU call r04o: rows-on-lanes scan with ping-pong queue halves (deferred stores, f32 segment values scaled in the merge kernel), 2048-row slices
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04o; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q --timeout 600 > $O/mscan_tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -4 $O/mscan_tests.txt | cut -c1-300
for i in 1 2; do timeout 200 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_mscan.py -m gpu -q -k "loose_bounds or many_ties or tiny" --timeout 300 2>&1 | tail -1; done
B="python bench.py --no-pmc --no-cpu-baseline"
timeout 200 $B > $O/b0.json 2>$O/b0.err; python -c "
import json; j=json.loads(open('$O/b0.json').read().strip().splitlines()[-1]); print('DEF', j['value'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])"
LANCE_HIP_GRAPH=0 LANCE_HIP_MS_PROF4=1 timeout 200 $B --steps 5 --warmup 2 --streams 1 2>&1 | grep "ms4 prof" | tail -1
LANCE_HIP_MS_DBG=2 timeout 200 $B --steps 10 --warmup 2 > $O/b2.json 2>$O/b2.err; python -c "
import json; j=json.loads(open('$O/b2.json').read().strip().splitlines()[-1]); print('NOPASS', j['value'], j['kernel_ms_per_step'])"
LANCE_HIP_Q_STATS=1 timeout 200 $B --steps 2 --warmup 1 2>&1 | grep qscan | tail -1
t0=$(date +%s)
timeout 200 python tests/fuzz_parity.py 100 4606 --log $O/fuzz_4606.txt > $O/fuzz_4606.out 2>&1; echo "fuzz rc=$? $(( $(date +%s)-t0 )) s"; tail -1 $O/fuzz_4606.out | cut -c1-300
