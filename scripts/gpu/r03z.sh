# GPU call r03z: SQ counters of the 8-query scan (is it VALU-bound as the A/B suggests?)
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03z; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
LANCE_HIP_Q8=1 timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/$O/pmc_q8 -- python $R/bench.py --steps 5 --no-cpu-baseline > $R/$O/pmc_q8.log 2>&1
LANCE_HIP_Q8=1 timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_q8b -- python $R/bench.py --steps 5 --no-cpu-baseline > $R/$O/pmc_q8b.log 2>&1
cd $R
python scripts/pmc_sq_summary.py $O/pmc_q8 $O/q8_pmc_sq.json ivfpq_q | cut -c1-500
python scripts/pmc_sq_summary.py $O/pmc_q8b $O/q8_pmc_sq_b.json ivfpq_q | cut -c1-500
rm -rf $O/pmc_q8 $O/pmc_q8b
cd /tmp
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_baseb -- python $R/bench.py --steps 5 --no-cpu-baseline > $R/$O/pmc_baseb.log 2>&1
cd $R
python scripts/pmc_sq_summary.py $O/pmc_baseb $O/base_pmc_sq_b.json ivfpq_q | cut -c1-500
rm -rf $O/pmc_baseb
