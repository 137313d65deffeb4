# GPU call r04r: largest-first slice order (32 work classes) -- A/B against index order and slice heights 1024 / 4096
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04r; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 300 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q --timeout 600 > $O/mscan_tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -2 $O/mscan_tests.txt | cut -c1-300
B="python bench.py --no-pmc --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/b_$tag.json 2>$O/b_$tag.err; python -c "
import json; j=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(j['value']), j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step']['ivfpq_scan_c1'], j['kernel_ms_per_step']['q_residual'], j['kernel_ms_per_step']['ivfpq_merge'])"; }
run ordered2048 X=1
run noorder2048 LANCE_HIP_MS_NOORDER=1
run ordered1024 LANCE_HIP_MS_RS=1024
run ordered4096 LANCE_HIP_MS_RS=4096
run ordered512 LANCE_HIP_MS_RS=512
