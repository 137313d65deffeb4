# GPU call r06w: the concurrent hierarchical trainer (engine-context pool on one GPU) and lance_hip_kmeans_split's f32-sample signature:
# the tests that cover them, then the C4 / C5 builds at 100M rows for the train_ivf stage time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06w; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_zz_gpu_two_ranks.py tests/test_zz_gpu_fullconfig.py tests/test_gpu_parity.py tests/test_zz_gpu_xform_fused.py -x -q -m gpu --timeout 900 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | cut -c1-300 | tail -8
t0=$(date +%s)
timeout 600 python bench.py --config c4 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c4_100m.json 2> $O/c4_100m.err; echo "c4 rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/c4_100m.json').read().strip().splitlines()[-1]); print('C4', j['value'], j['recall_at_10'], j['build_sec'], j.get('build_stages_ms'))" 2>&1 | cut -c1-900; tail -3 $O/c4_100m.err | cut -c1-300
t0=$(date +%s)
timeout 900 python bench.py --config c5 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c5_100m.json 2> $O/c5_100m.err; echo "c5 rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/c5_100m.json').read().strip().splitlines()[-1]); print('C5', j['value'], j['recall_at_10'], j['build_sec'], j.get('build_stages_ms'))" 2>&1 | cut -c1-900; tail -3 $O/c5_100m.err | cut -c1-300
for c in 1 2 4 16; do LANCE_HIP_HIER_CONTEXTS=$c timeout 300 python scripts/probe_hier.py > $O/hier_$c.txt 2>&1; tail -2 $O/hier_$c.txt | cut -c1-300; done
