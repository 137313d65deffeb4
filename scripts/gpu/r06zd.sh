# GPU call r06zd: per-group keys with the larger candidate list: parity, the C5 bench at 20M and 100M rows, the probe
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zd; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_zz_gpu_coarse_mfma.py -x -q -m gpu --timeout 1400 > $O/tests_coarse.txt 2>&1; echo "coarse tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/tests_coarse.txt | cut -c1-600 | tail -12
LANCE_HIP_GRAPH=0 LANCE_HIP_COARSE_STATS=1 timeout 900 python bench.py --config c5 --n 20000000 --steps 6 --warmup 1 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c5_20m.json 2> $O/c5_20m.err; echo "rc=$?"
grep "coarse\]" $O/c5_20m.err | sort | uniq -c | sort -rn | head -8 | cut -c1-200
python -c "
import json; j=json.loads(open('$O/c5_20m.json').read().strip().splitlines()[-1]); print('C5 20M', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
timeout 900 python bench.py --config c5 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c5_100m.json 2> $O/c5_100m.err; echo "c5 rc=$?"
python -c "
import json; j=json.loads(open('$O/c5_100m.json').read().strip().splitlines()[-1]); print('C5 100M', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
timeout 600 python bench.py --config c4 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c4_100m.json 2> $O/c4_100m.err; echo "c4 rc=$?"
python -c "
import json; j=json.loads(open('$O/c4_100m.json').read().strip().splitlines()[-1]); print('C4', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
