# GPU call r06y: f32 assign / IVF k-means E-step on phases 1-3 of the transform kernel (xf_kernel<.., ASSIGN>) on top of r06x: full GPU suite, C2 build stages, kernel stats, fuzz
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06y; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_zz_gpu_xform_fused.py -x -q -m gpu --timeout 900 > $O/tests_xf.txt 2>&1; echo "xf tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/tests_xf.txt | cut -c1-400 | tail -12
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 --deselect tests/test_zz_gpu_xform_fused.py > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -8
timeout 600 python bench.py --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('BENCH', j['value'], j['recall_at_10'], j['build_sec'], j['build_stages_ms'], j['roofline_build']['estep_ivf'])" 2>&1 | cut -c1-1500
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-grid --no-extras > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/bench_kernel_stats.csv; grep -E "xf_|pq_mfma|ma_|group_|kmeans_|fill_" $O/bench_kernel_stats.csv | cut -c1-200; fi
timeout 700 python tests/fuzz_parity.py 200 6202 --log $O/fuzz.txt --watchdog 300 > $O/fuzz_out.txt 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_out.txt | cut -c1-300
