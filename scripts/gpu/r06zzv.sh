# GPU call r06zzv: three queue parts per wave in the matrix-core scan (-DMS_DEEP_FLUSH variant): parity + bench A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzv; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
LANCE_HIP_LIB=$R/build/variants/liblance_hip_msdeep.so timeout 900 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_dot_flow.py tests/test_gpu_parity.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt | cut -c1-200
for v in main msdeep main msdeep; do
if [ $v == main ]; then unset LANCE_HIP_LIB; else export LANCE_HIP_LIB=$R/build/variants/liblance_hip_$v.so; fi
timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print('BENCH $v', j['value'], j['ms_per_step'], j['kernel_ms_per_step']['ivfpq_merge'], j['kernel_ms_per_step']['ivfpq_scan_c1'], j['recall_at_10'])"
done
