# GPU call r06zzl: merge kernel with the compaction list in the histogram's words (2 KiB less LDS): parity + bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzl; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_dot_flow.py tests/test_gpu_parity.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt | cut -c1-200
for i in 1 2; do
timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print('BENCH', j['value'], j['ms_per_step'], j['kernel_ms_per_step']['ivfpq_merge'], j['kernel_ms_per_step']['ivfpq_scan_c1'])"
done
