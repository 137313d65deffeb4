# GPU call r02v: round-2 validation -- full -m gpu suite, smoke, bench with the CPU baseline, rocprofv3 kernel stats, SQ counters
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02v; mkdir -p $O
cd $R
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import oracle; print('usable cpus', oracle.usable_cpus(), 'omp threads', oracle.num_threads())"
timeout 560 python -m pytest tests -m gpu -x -q --timeout 240 --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -16 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py --steps 20 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 2500 $O/bench_n1.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 10 --no-cpu-baseline > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv; head -12 $O/bench_kernel_stats.csv | cut -c1-200
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_lds -- python $R/bench.py --steps 5 --no-cpu-baseline > $O/pmc_lds.log 2>&1
python $R/scripts/pmc_sq_summary.py $O/pmc_lds $O/pmc_lds_summary.json ivfpq_q | tail -8
rm -rf $O/*/*/*.db $O/*/*/*kernel_trace.csv $O/*/*/*counter_collection.csv 2>/dev/null
du -sh $O
