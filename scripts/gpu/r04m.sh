# GPU call r04m: rows-on-lanes scan (v4: 16 waves per CU, limit folded into the accumulator start, synchronous flush): parity, bench A/B against v3, fuzz, SQ counters
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04m; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q --timeout 600 > $O/mscan_tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -4 $O/mscan_tests.txt | cut -c1-300
for i in 1 2; do timeout 200 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_mscan.py -m gpu -q -k "loose_bounds or many_ties or tiny" --timeout 300 2>&1 | tail -1; done
timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_ms3.json 2> $O/bench_ms3.err; python -c "
import json; j=json.loads(open('$O/bench_ms3.json').read().strip().splitlines()[-1]); print('MSCAN4', j['value'], j['ms_per_step'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])"
LANCE_HIP_MS_V3=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_ms3b.json 2> $O/bench_ms3b.err; python -c "
import json; j=json.loads(open('$O/bench_ms3b.json').read().strip().splitlines()[-1]); print('MSCAN3', j['value'], j['ms_per_step'], j['kernel_ms_per_step'])"
t0=$(date +%s)
timeout 200 python tests/fuzz_parity.py 100 4505 --log $O/fuzz_4505.txt > $O/fuzz_4505.out 2>&1; echo "fuzz rc=$? $(( $(date +%s)-t0 )) s"; tail -2 $O/fuzz_4505.out | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/$O/pmc_a -- python $R/bench.py --steps 5 --no-cpu-baseline --no-pmc > $R/$O/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_b -- python $R/bench.py --steps 5 --no-cpu-baseline --no-pmc > $R/$O/pmc_b.log 2>&1
cd $R
python scripts/pmc_sq_summary.py $O/pmc_a $O/mscan4_pmc_a.json mscan4 | cut -c1-600
python scripts/pmc_sq_summary.py $O/pmc_b $O/mscan4_pmc_b.json mscan4 | cut -c1-600
rm -rf $O/pmc_a $O/pmc_b
