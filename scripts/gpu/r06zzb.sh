# GPU call r06zzb: dot flow with per-list bounds of the centred reconstruction norm: parity, rates; bench line with other_metrics
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzb; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_dot_flow.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_mscan.py tests/test_zz_gpu_msbound.py -x -q -m gpu --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt | cut -c1-300
timeout 900 python scripts/probe_dot_flow.py child > $O/dot_probe.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/dot_probe.txt | grep -A1 " dot" | cut -c1-600
LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_dot_flow.py child 2>&1 | grep "qscan\]" | uniq -c | head -12 | cut -c1-400
t0=$(date +%s); timeout 900 python bench.py --no-pmc --no-cpu-baseline --no-grid > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('BENCH', j['value'], j['ms_per_step']); print(json.dumps(j.get('other_metrics'))[:1500])"
