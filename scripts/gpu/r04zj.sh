# GPU call r04zj: last check of the library rebuilt from the committed tree (after the reverted records attempt): scan parity + smoke + bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04zj; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -1 $O/tests.txt | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 100 python bench.py --no-pmc --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEF', round(j['value']), j['recall_at_10'], j['kernel_ms_per_step']['ivfpq_scan_c1'], j['kernel_ms_per_step']['ivfpq_merge'])"
