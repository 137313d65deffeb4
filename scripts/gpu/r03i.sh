# GPU call r03i: PQ sub-quantiser argmin on the matrix cores (pq_mfma.hip) -- parity suite, build-time A/B (LANCE_HIP_NO_MFMA_PQ)
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/pytest_all.log | cut -c1-300 | tail -12
timeout 200 python bench.py --no-cpu-baseline --steps 5 > $O/bench_pqm.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/bench_pqm.json').read().strip().splitlines()[-1]); print('pq mfma', j['build_sec'], j['build_stages_ms'], j['recall_at_10'])"
LANCE_HIP_NO_MFMA_PQ=1 timeout 200 python bench.py --no-cpu-baseline --steps 5 > $O/bench_nopqm.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/bench_nopqm.json').read().strip().splitlines()[-1]); print('exact  ', j['build_sec'], j['build_stages_ms'], j['recall_at_10'])"
timeout 300 python scripts/probe_c3_search.py 1000000 > $O/c3_probe.json 2>/dev/null; python -c "import json; j=json.load(open('$O/c3_probe.json')); print('c3 build pq mfma', j['build_stages_ms'])"
LANCE_HIP_NO_MFMA_PQ=1 timeout 300 python scripts/probe_c3_search.py 1000000 > $O/c3_probe_nopqm.json 2>/dev/null; python -c "import json; j=json.load(open('$O/c3_probe_nopqm.json')); print('c3 build exact  ', j['build_stages_ms'])"
