# GPU call r04zi: survivors as records + scatter kernel (no queue / flush in the scan): parity, bench, short fuzz
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zi; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 300 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q -x --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -3 $O/tests.txt | cut -c1-300
timeout 100 python bench.py --no-pmc --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEF', round(j['value']), j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])"
timeout 100 python bench.py --no-pmc --no-cpu-baseline --steps 20 --warmup 5 --streams 1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S1', round(j['value']))"
timeout 100 python tests/fuzz_parity.py 45 5104 --log $O/fuzz.txt > $O/fuzz.out 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.out | cut -c1-200
