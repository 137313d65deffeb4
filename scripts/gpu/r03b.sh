# GPU call r03b: diagnose the fuzz findings of r03a (--debug replays), C3 per-kernel breakdown, the restructured flat MFMA epilogue
# (parity + timing + SQ counters), C99 device harness, C5 golden, HBM-traffic counter passes of the bench
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03b; mkdir -p $O
# 1. fuzz replays with diagnostics
for sc in "11 4" "13 77" "11 59" "13 46"; do set -- $sc; timeout 240 python tests/fuzz_parity.py 200 $1 --case $2 --debug > $O/fuzz_debug_$1_$2.txt 2>&1; grep -E "debug|MISMATCH|fuzz " $O/fuzz_debug_$1_$2.txt | cut -c1-420; done
# 2. parity of what changed since r03a
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_abi.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_wide_mfma.py -m gpu -q --timeout 400 -k "flat or c99 or c5 or wide or cosine" > $O/pytest_changed.log 2>&1; echo "changed rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|^E  " $O/pytest_changed.log | cut -c1-300 | tail -15
# 3. C3 breakdown
LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_probe.json 2> $O/c3_probe.err; cat $O/c3_probe.json | cut -c1-2500; grep qscan $O/c3_probe.err | tail -4 | cut -c1-300
LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qtprof.so LANCE_HIP_QT_PROF=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_probe_prof.json 2> $O/c3_probe_prof.err; grep "qt prof" $O/c3_probe_prof.err | tail -6 | cut -c1-200
# 4. flat MFMA: timing + equality with the exact filter, then SQ counters
timeout 300 python scripts/probe_flat_batch.py > $O/flat_batch.txt 2>&1; tail -5 $O/flat_batch.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_flat -- python $GRAFT_REPO_ROOT/scripts/probe_flat_batch.py mfma > $GRAFT_REPO_ROOT/$O/pmc_flat.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/pmc_sq_summary.py $O/pmc_flat $O/flat_pmc_sq.json flat_ | cut -c1-400
# 5. HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate passes (guide), plus the kernel trace
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/pmc_tcc_summary.py $O/pmc_fetch $O/pmc_write $O/bench_pmc_tcc.json "python bench.py --steps 5 --no-cpu-baseline" | cut -c1-300 | head -14
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_flat
timeout 200 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-1500
