# GPU call r06zzzy: a partition's slices after its first one at their own height (LANCE_HIP_MS_RS2): parity of the scan tests at RS2 = 1024, bench at RS2 = off / 2048 / 1536 / 1024 / 512
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzzy; mkdir -p $O; export TMPDIR=/tmp
LANCE_HIP_MS_RS2=1024 timeout 600 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_parity.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -1 $O/tests.txt | cut -c1-200
run() { env $1 timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras ${@:2} > $O/b.json 2> $O/b.err
python -c "
import json; j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); k=j['kernel_ms_per_step']; print('RUN $*', round(j['value']), round(j['ms_per_step'],4), 'scan', k['ivfpq_scan_c1'], j['recall_at_10'])"; }
run A=1; run LANCE_HIP_MS_RS2=2048; run LANCE_HIP_MS_RS2=1536; run LANCE_HIP_MS_RS2=1024; run LANCE_HIP_MS_RS2=512; run A=1
