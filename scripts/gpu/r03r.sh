# GPU call r03r: 1024-lane rescan for the tiled shapes + bound-pass scale from per-index codebook means: tests, C3 probe, bench
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_fuzz_findings.py tests/test_gpu_search.py -m gpu -q -x --timeout 400 > $O/tests.txt 2>&1; echo "rc=$?"; tail -3 $O/tests.txt | cut -c1-300
LANCE_HIP_Q_STATS=1 timeout 200 python scripts/probe_c3_search.py > $O/c3.json 2> $O/c3.err; grep "qscan" $O/c3.err | sort | uniq -c | head -5 | cut -c1-250
python -c "import json; j=json.load(open('$O/c3.json')); print('c3', j['nprobes10_refine10']['wall_ms_per_batch'], j['nprobes10_refine10']['kernel_ms_per_batch']); print(j['nprobes10_refine0']['wall_ms_per_batch'], j['nprobes50_refine10']['wall_ms_per_batch'], j['nprobes50_refine10']['kernel_ms_per_batch'])"
timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', j['value'], j['ms_per_step'], j['recall_at_10'], j['kernel_ms_per_step'])"
