# GPU call r03p: C3 merge kernel -- 256 lanes per query, survivor statistics
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03p; mkdir -p $O
LANCE_HIP_QMERGE_BS=256 timeout 200 python scripts/probe_c3_search.py > $O/c3_qm256.json 2>/dev/null; python -c "import json; j=json.load(open('$O/c3_qm256.json')); print('qm256', j['nprobes10_refine10']['wall_ms_per_batch'], j['nprobes10_refine10']['kernel_ms_per_batch']); print(j['nprobes50_refine10']['wall_ms_per_batch'], j['nprobes50_refine10']['kernel_ms_per_batch'])"
LANCE_HIP_Q_STATS=1 timeout 200 python scripts/probe_c3_search.py > $O/c3_stats.json 2> $O/c3_stats.err; grep -i "stat\|surviv" $O/c3_stats.err | sort | uniq -c | sort -rn | head -12 | cut -c1-300
python -c "import json; j=json.load(open('$O/c3_stats.json')); print('base', j['nprobes10_refine10']['wall_ms_per_batch'], j['nprobes10_refine10']['kernel_ms_per_batch']); print(j['nprobes50_refine10']['wall_ms_per_batch'], j['nprobes50_refine10']['kernel_ms_per_batch'])"
LANCE_HIP_QMERGE_BS=256 timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench_qm256.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/bench_qm256.json').read().strip().splitlines()[-1]); print('bench qm256', j['value'], j['ms_per_step'], j['kernel_ms_per_step'])"
