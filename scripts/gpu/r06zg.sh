# GPU call r06zg: one full round of workgroups in the find_partitions sweeps (slice count rounded down): probe, coarse parity, C5 / C4 at 100M rows, C2
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zg; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/probe_coarse.py > $O/probe.txt 2>&1; echo "probe rc=$?"; grep -E "^nlist" $O/probe.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_zz_gpu_coarse_mfma.py -x -q -m gpu --timeout 1400 > $O/tests_coarse.txt 2>&1; echo "coarse tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/tests_coarse.txt | cut -c1-600 | tail -12
timeout 900 python bench.py --config c5 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c5_100m.json 2> $O/c5_100m.err; echo "c5 rc=$?"
python -c "
import json; j=json.loads(open('$O/c5_100m.json').read().strip().splitlines()[-1]); print('C5 100M', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
timeout 900 python bench.py --config c5 --n 100000000 --nprobes 32 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c5_100m_np32.json 2> $O/c5_100m_np32.err; echo "c5 np32 rc=$?"
python -c "
import json; j=json.loads(open('$O/c5_100m_np32.json').read().strip().splitlines()[-1]); print('C5 100M np32', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
timeout 600 python bench.py --config c4 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c4_100m.json 2> $O/c4_100m.err; echo "c4 rc=$?"
python -c "
import json; j=json.loads(open('$O/c4_100m.json').read().strip().splitlines()[-1]); print('C4', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c2.json 2> $O/c2.err; echo "c2 rc=$?"
python -c "
import json; j=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print('C2', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
