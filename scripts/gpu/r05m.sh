# GPU call r05m: PQ training on per-sub-quantiser contiguous slices (parity + build A/B); fuzz incl. the long-row flat batches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05m; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_flat_wide.py tests/test_abi.py -m gpu -q --timeout 800 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | cut -c1-300 | tail -8; grep -E "^E  " $O/tests.txt | head -8 | cut -c1-300
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), 'build', j['build_sec'], j['roofline_build']['build_stages_ms'])" | cut -c1-400; }
B="--steps 50 --warmup 5 --no-pmc --no-cpu-baseline --no-grid"
timeout 300 python bench.py $B > $O/bench_slices.json 2> $O/bench_slices.err; brief $O/bench_slices.json SLICES
LANCE_HIP_PQ_NO_SLICES=1 timeout 300 python bench.py $B > $O/bench_rowmajor.json 2> $O/bench_rowmajor.err; brief $O/bench_rowmajor.json ROWMAJOR
timeout 300 python bench.py $B > $O/bench_slices2.json 2> $O/bench_slices2.err; brief $O/bench_slices2.json SLICES
timeout 120 python scripts/probe_build_steps.py 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('pq_train', j['pq: pq_train'], j['pq_train kernel ms by stage'], 'ivf', j['ivf: kmeans_train'])"
timeout 400 python tests/fuzz_parity.py 240 5207 --log $O/fuzz.txt > $O/fuzz.out 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.out | cut -c1-300
