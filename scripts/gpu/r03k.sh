# GPU call r03k: MFMA fused encode + parallel k-means control kernel -- suite, build A/B
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/pytest_all.log | cut -c1-300 | tail -12
timeout 200 python bench.py --no-cpu-baseline --steps 5 > $O/bench.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('new    ', j['build_sec'], j['build_stages_ms'], j['recall_at_10'], j['value'])"
LANCE_HIP_NO_MFMA_ENCODE=1 timeout 200 python bench.py --no-cpu-baseline --steps 5 > $O/bench_noenc.json 2>/dev/null; python -c "import json; j=json.loads(open('$O/bench_noenc.json').read().strip().splitlines()[-1]); print('no mfma encode', j['build_sec'], j['build_stages_ms'])"
timeout 300 python scripts/scale_probe.py 20000000 f16 > $O/scale_c4_20M.txt 2>&1; tail -4 $O/scale_c4_20M.txt | cut -c1-600
