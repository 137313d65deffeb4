# GPU call r06zi: the centroids' bf16 planes as constants of the index (find_partitions over thousands of lists): parity, C5 / C4 at 100M rows
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zi; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_zz_gpu_coarse_mfma.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_threads.py tests/test_zz_gpu_graph.py tests/test_gpu_parity.py -x -q -m gpu --timeout 1400 > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/tests.txt | cut -c1-600 | tail -12
timeout 900 python bench.py --config c5 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c5_100m.json 2> $O/c5_100m.err; echo "c5 rc=$?"
python -c "
import json; j=json.loads(open('$O/c5_100m.json').read().strip().splitlines()[-1]); print('C5 100M', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['build_stages_ms'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-900
timeout 600 python bench.py --config c4 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c4_100m.json 2> $O/c4_100m.err; echo "c4 rc=$?"
python -c "
import json; j=json.loads(open('$O/c4_100m.json').read().strip().splitlines()[-1]); print('C4', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
