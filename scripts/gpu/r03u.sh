# GPU call r03u: round-3 closing validation and measurement pass on the final tree
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03u; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout 400 > $O/gpu_suite.txt 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -- python $R/bench.py --steps 5 --no-cpu-baseline > $R/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -- python $R/bench.py --steps 5 --no-cpu-baseline > $R/$O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/$O/pmc_lds -- python $R/bench.py --steps 5 --no-cpu-baseline > $R/$O/pmc_lds.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/bench.py --steps 20 --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R
python scripts/pmc_tcc_summary.py $O/pmc_fetch $O/pmc_write $O/bench_pmc_tcc.json "python bench.py --steps 5 --no-cpu-baseline" | cut -c1-220 | head -8
python scripts/pmc_sq_summary.py $O/pmc_lds $O/scan_pmc_sq.json ivfpq_q | cut -c1-400
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/bench_kernel_stats.csv; head -12 $O/bench_kernel_stats.csv | cut -c1-150
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_lds $O/prof
cp $O/bench_pmc_tcc.json profiles/r03_bench_pmc_tcc.json
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -1 $O/bench_n1.json | cut -c1-1500
LANCE_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --steps 5 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -1 $O/bench_force_dist.json | cut -c1-500
timeout 300 python bench.py --config c4 --n 8000000 --no-cpu-baseline --steps 5 > $O/bench_c4_8M.json 2> $O/bench_c4.err; tail -1 $O/bench_c4_8M.json | cut -c1-700
timeout 600 python scripts/measure_grid.py --c3 > $O/grid.json 2> $O/grid.err; python -c "
import json; j=json.load(open('$O/grid.json'))
print('c1', {k: round(v['ms'],3) for k,v in j['c1_flat']['by_batch_size'].items()}, j['c1_roofline_batch_10k']['frac'], j['c1_roofline_batch_10k']['frac_executed'])
print('c2 build', j['c2_build']['sec'], [(g['nprobes'], g['refine_factor'], round(g['recall_at_10'],3), int(g['qps'])) for g in j['c2_grid']])
print('c3', j['c3']['build_sec'], j['c3']['stages_ms'], j['c3']['flat_1000q_ms'], [(g['nprobes'], g['refine_factor'], round(g['recall_at_10'],3), int(g['qps'])) for g in j['c3']['grid']], j['c3'].get('roofline'))
"
timeout 200 python scripts/probe_c3_search.py > $O/c3.json 2>/dev/null; python -c "import json; j=json.load(open('$O/c3.json')); print('c3', [(k, v['wall_ms_per_batch'], v['kernel_ms_per_batch']) for k,v in j.items() if k.startswith('nprobes')])"
LANCE_HIP_QMERGE_BS=256 timeout 200 python scripts/probe_c3_search.py > $O/c3_qm256.json 2>/dev/null; python -c "import json; j=json.load(open('$O/c3_qm256.json')); print('c3 qm256', [(k, v['wall_ms_per_batch'], v['kernel_ms_per_batch']['ivfpq_merge']) for k,v in j.items() if k.startswith('nprobes')])"
(timeout 200 python tests/fuzz_parity.py 150 71 --log $O/fuzz_71.txt > /dev/null 2>&1 &
 timeout 200 python tests/fuzz_parity.py 150 72 --log $O/fuzz_72.txt > /dev/null 2>&1 &
 timeout 200 python tests/fuzz_parity.py 150 73 --log $O/fuzz_73.txt > /dev/null 2>&1 &
 wait)
grep -hE "SKIP|MISMATCH|ERROR|fuzz " $O/fuzz_71.txt $O/fuzz_72.txt $O/fuzz_73.txt | cut -c1-400
