# GPU call r06zzz: slice height 3072 as the default: scan parity tests; engine contexts 2 / 3 / 4 and merge lanes 128 / 256 beside it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzz; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_mscan.py tests/test_zz_gpu_msbound.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_dot_flow.py tests/test_gpu_parity.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/tests.txt | cut -c1-200
run() { env $1 timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras ${@:2} > $O/b.json 2> $O/b.err
python -c "
import json; j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']; print('RUN $*', round(j['value']), round(j['ms_per_step'],4), 'scan', k['ivfpq_scan_c1'], 'merge', k['ivfpq_merge'], j['recall_at_10'])"; }
run A=1; run A=1 --streams 2; run A=1 --streams 4; run LANCE_HIP_QMERGE_BS=256; run A=1
