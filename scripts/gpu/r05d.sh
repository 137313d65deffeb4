# GPU call r05d: refine from the lossless u8 copy (new tests + the suites that refine), bench A/B against the f32 column, first run of the
# bench's latency / recall-grid additions
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_zz_gpu_refine_u8.py tests/test_zz_gpu_mscan.py tests/test_zz_gpu_graph.py tests/test_gpu_parity.py tests/test_abi.py tests/test_zz_gpu_threads.py -m gpu -q --timeout 600 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -15 $O/tests.txt | cut -c1-300
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['exact_replays_last_step'], j['kernel_ms_per_step'], j.get('refine_source','')[:40])" | cut -c1-700; }
B="--steps 200 --warmup 10 --no-pmc --no-cpu-baseline"
timeout 300 python bench.py $B > $O/bench_u8.json 2> $O/bench_u8.err; echo "rc=$?"; brief $O/bench_u8.json U8; tail -3 $O/bench_u8.err | cut -c1-300
LANCE_HIP_NO_RAW_COMPACT=1 timeout 300 python bench.py $B --no-grid > $O/bench_f32.json 2> $O/bench_f32.err; brief $O/bench_f32.json F32
python -c "
import json; j=json.loads(open('$O/bench_u8.json').read().strip().splitlines()[-1]); print('latency', j.get('latency')); print('grid', j.get('recall_grid')); print('tail', j.get('roofline_tail')); print('stages', j['roofline_build']['build_stages_ms'])" | cut -c1-1800
