# GPU call r05q: SQ counters of the long-row flat filter (what its waves wait for)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$O/pmc_a -- python $R/scripts/probe_flat_wide.py child 1000000 1536 1000 l2 1 > $R/$O/pmc_a.log 2>&1; echo "pmc a rc=$?"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/$O/pmc_b -- python $R/scripts/probe_flat_wide.py child 1000000 1536 1000 l2 1 > $R/$O/pmc_b.log 2>&1; echo "pmc b rc=$?"
cd $R
python scripts/pmc_sq_summary.py $O/pmc_a $O/flat_wide_pmc_a.json "flat_filter_mfma_wide" | cut -c1-700
python scripts/pmc_sq_summary.py $O/pmc_b $O/flat_wide_pmc_b.json "flat_filter_mfma_wide" | cut -c1-700
tail -2 $O/pmc_b.log | cut -c1-300
rm -rf $O/pmc_a $O/pmc_b
