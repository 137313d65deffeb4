# GPU call r06h: batched row loads; SQ counters of the fused transform kernel at C2
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz_gpu_xform_fused.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | head -8 | cut -c1-300
LANCE_HIP_XF_PROF=1 timeout 600 python scripts/probe_xform.py c2 2>&1 | grep -v amdgpu | grep -E "xf prof" | tail -1 | cut -c1-420
OUT=$O/probe_fused.json timeout 600 python scripts/probe_xform.py c2 2>&1 | grep -v amdgpu | cut -c1-420
(cd /tmp && PYTHONPATH=$R timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$O/pmc_a -- python $R/scripts/probe_xform.py c2 > $R/$O/pmc_a.log 2>&1); echo "pmc a rc=$?"
(cd /tmp && PYTHONPATH=$R timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_b -- python $R/scripts/probe_xform.py c2 > $R/$O/pmc_b.log 2>&1); echo "pmc b rc=$?"
timeout 100 python scripts/pmc_sq_summary.py $O/pmc_a $O/xf_pmc_a.json "xf_kernel" | cut -c1-700
timeout 100 python scripts/pmc_sq_summary.py $O/pmc_b $O/xf_pmc_b.json "xf_kernel" | cut -c1-700
rm -rf $O/pmc_a $O/pmc_b
