# GPU call r06zzu: validation of the final round-6 tree (two-pass flat merge + host flag word, raw scan queue entries): full -m gpu suite, smoke(), the bench line in the
# driver's form (PMC traffic + CPU leg + extras), rocprofv3 kernel stats of the same command, SQ counters of the scan kernel, fuzz, the SURVEY 8(d) grid incl. C3
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzu; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-300
t0=$(date +%s)
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); r=j['roofline']; print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['build_sec_pcie_inclusive'], j['kernel_ms_per_step']); print('ROOF', r['kernel'][:40], r['bound'], r['achieved'], r['frac'], r['frac_algorithmic'], r['peak_measured'], r['traffic'], r['avg_launch_ms']); print('CPU', j['cpu_baseline'])"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 50 --warmup 5 --no-pmc --no-cpu-baseline --no-grid --no-extras > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/bench_kernel_stats.csv; head -8 $O/bench_kernel_stats.csv | cut -c1-170; fi
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmc_a -- python $R/bench.py --steps 10 --warmup 2 --streams 1 --no-pmc --no-cpu-baseline --no-grid --no-extras > $R/$O/pmc_a.log 2>&1); echo "pmc a rc=$?"
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_b -- python $R/bench.py --steps 10 --warmup 2 --streams 1 --no-pmc --no-cpu-baseline --no-grid --no-extras > $R/$O/pmc_b.log 2>&1); echo "pmc b rc=$?"
for kn in ivfpq_mscan_kernel; do timeout 100 python scripts/pmc_sq_summary.py /tmp/pmc_a $O/pmc_a_$kn.json $kn | cut -c1-500; timeout 100 python scripts/pmc_sq_summary.py /tmp/pmc_b $O/pmc_b_$kn.json $kn | cut -c1-500; done
timeout 600 python tests/fuzz_parity.py 250 6401 --log $O/fuzz.txt --watchdog 300 > $O/fuzz_out.txt 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz_out.txt | cut -c1-300
timeout 400 python tests/fuzz_dot_flow.py 150 7101 > $O/fuzz_dot.txt 2>&1; echo "dot fuzz rc=$?"; grep -v amdgpu.ids $O/fuzz_dot.txt | tail -2 | cut -c1-400
timeout 900 python scripts/measure_grid.py --c3 > $O/grid.json 2> $O/grid.err; echo "grid rc=$?"; tail -c 400 $O/grid.json
