# GPU call r03e: pipelined tiled table build (512 lanes, 6-8 codewords in flight per lane) -- parity, C3 breakdown, phase clocks; then fuzz run 2
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_fullconfig.py -m gpu -x -q --timeout 200 -k "tiled or c3" > $O/pytest_tiled.log 2>&1; echo "tiled rc=$?"; tail -4 $O/pytest_tiled.log | cut -c1-300
LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_probe.json 2> $O/c3_probe.err; cat $O/c3_probe.json | cut -c1-2500; grep qscan $O/c3_probe.err | tail -3 | cut -c1-300
LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qtprof.so LANCE_HIP_QT_PROF=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_probe_prof.json 2> $O/c3_probe_prof.err; grep "qt prof" $O/c3_probe_prof.err | tail -3 | cut -c1-200
for s in 21 22 23; do OMP_NUM_THREADS=5 timeout 700 python tests/fuzz_parity.py 540 $s --log $O/fuzz_seed$s.log > $O/fuzz_seed$s.out 2>&1 & done
wait
for s in 21 22 23; do tail -n 3 $O/fuzz_seed$s.log | cut -c1-500; done
