# GPU call r06x: PQ codebook training E-step on the transform's PQ phase (xf_pqtrain_kernel): parity tests, then the C2 build stages and kernel stats
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06x; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_xform_fused.py -x -q -m gpu --timeout 900 > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | cut -c1-300 | tail -8
timeout 600 python bench.py --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('BENCH', j['value'], j['recall_at_10'], j['build_sec'], j['build_stages_ms'])" 2>&1 | cut -c1-600
LANCE_HIP_NO_XF_TRAIN=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench_old.json 2> $O/bench_old.err; echo "bench old rc=$?"
python -c "
import json; j=json.loads(open('$O/bench_old.json').read().strip().splitlines()[-1]); print('BENCH old', j['value'], j['recall_at_10'], j['build_sec'], j['build_stages_ms'])" 2>&1 | cut -c1-600
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-grid --no-extras > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/bench_kernel_stats.csv; grep -E "xf_pqtrain|pq_mfma|xf_pq_prep|ma_top3|ma_finalize|group_|kmeans_|fill_" $O/bench_kernel_stats.csv | cut -c1-200; fi
timeout 600 python tests/fuzz_parity.py 150 6201 --log $O/fuzz.txt --watchdog 300 > $O/fuzz_out.txt 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_out.txt | cut -c1-300
