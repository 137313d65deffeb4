# GPU call r05g: bound pass v2 (staging loads together, codes one chunk ahead) parity + bench; merge kernel phase experiments (variant build with
# early returns, results wrong by design); engine-context count and scan-grid knobs on the current kernel mix
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_zz_gpu_msbound.py tests/test_zz_gpu_mscan.py tests/test_zz_gpu_refine_u8.py tests/test_gpu_parity.py tests/test_zz_gpu_fullconfig.py -m gpu -q --timeout 800 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -6 $O/tests.txt | cut -c1-300
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), j['ms_per_step'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])" | cut -c1-600; }
B="--steps 200 --warmup 10 --no-pmc --no-cpu-baseline --no-grid"
timeout 300 python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo "rc=$?"; brief $O/bench_new.json NEW
timeout 300 python bench.py $B --streams 4 > $O/bench_s4.json 2> $O/bench_s4.err; brief $O/bench_s4.json S4
timeout 300 python bench.py $B --streams 2 > $O/bench_s2.json 2> $O/bench_s2.err; brief $O/bench_s2.json S2
LANCE_HIP_MS_GRID=224 timeout 300 python bench.py $B > $O/bench_g224.json 2> $O/bench_g224.err; brief $O/bench_g224.json GRID224
LANCE_HIP_MS_GRID=192 timeout 300 python bench.py $B > $O/bench_g192.json 2> $O/bench_g192.err; brief $O/bench_g192.json GRID192
for n in 1 2 3 4; do
  LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qmdbg.so LANCE_HIP_QM_DBG=$n timeout 200 python bench.py --steps 100 --warmup 5 --no-pmc --no-cpu-baseline --no-grid --streams 1 > $O/qm$n.json 2> $O/qm$n.err
  python -c "
import json; j=json.loads(open('$O/qm$n.json').read().strip().splitlines()[-1]); print('QM_DBG=$n merge ms', j['kernel_ms_per_step']['ivfpq_merge'])"
done
timeout 200 python tests/fuzz_parity.py 60 5205 --log $O/fuzz.txt > $O/fuzz.out 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.out | cut -c1-300
