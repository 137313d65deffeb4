# GPU call r03f: (1) where does fuzz case 22/0 hang? (watchdog stack dumps)  (2) tiled scan with the residuals in LDS  (3) flat MFMA with 4 accumulator blocks
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03f; mkdir -p $O
timeout 120 python tests/fuzz_parity.py 60 22 --case 0 --debug --watchdog 25 > $O/fuzz_22_0.txt 2>&1; echo "rc=$?"; grep -v "^  File \"/usr" $O/fuzz_22_0.txt | tail -40 | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_fullconfig.py -m gpu -x -q --timeout 200 -k "tiled or c3" > $O/pytest_tiled.log 2>&1; echo "tiled rc=$?"; tail -4 $O/pytest_tiled.log | cut -c1-300
timeout 300 python scripts/probe_c3_search.py > $O/c3_probe.json 2> $O/c3_probe.err; cat $O/c3_probe.json | cut -c1-2500
LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qtprof.so LANCE_HIP_QT_PROF=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_probe_prof.json 2> $O/c3_probe_prof.err; grep "qt prof" $O/c3_probe_prof.err | tail -3 | cut -c1-200
LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_fm4acc.so timeout 300 python scripts/probe_flat_batch.py > $O/flat_batch_4acc.txt 2>&1; tail -5 $O/flat_batch_4acc.txt | cut -c1-400
for st in 2 3 4 6; do timeout 200 python bench.py --no-cpu-baseline --streams $st > $O/bench_streams$st.json 2> /dev/null; python -c "import json,sys; j=json.loads(open('$O/bench_streams$st.json').read().strip().splitlines()[-1]); print($st, j['value'], j['ms_per_step'], j['build_sec'])"; done
