# GPU call r06j: full suite on the tree with the fused transform; probe; bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06j; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -12; grep -E "^E  " $O/gpu_suite.txt | head -10 | cut -c1-300
OUT=$O/probe_fused.json timeout 600 python scripts/probe_xform.py all 2>&1 | grep -v amdgpu | cut -c1-420
timeout 600 python bench.py --no-pmc > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); r=j['roofline']; print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec']); print('BUILD', j['roofline_build'])" | cut -c1-1500
