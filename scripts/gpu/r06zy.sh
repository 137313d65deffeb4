# GPU call r06zy: survivors as {position, value} records (one 8-byte store, one 8-byte load): parity of the scan / merge tests, bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zy; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_mscan.py tests/test_zz_gpu_msbound.py tests/test_zz_gpu_dot_flow.py tests/test_gpu_pm_scan.py tests/test_gpu_parity.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt | cut -c1-300
for i in 1 2; do
timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); r=j['roofline']; print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step']); print('ROOF', r['avg_launch_ms'], r['frac'])"
done
