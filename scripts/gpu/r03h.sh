# GPU call r03h: where does the tiled table build spend its time?  three ablated variants (no arithmetic / no codeword loads / no residual loads)
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03h; mkdir -p $O
for a in 1 2 3; do LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qtabl$a.so timeout 200 python scripts/probe_c3_search.py > $O/c3_abl$a.json 2> /dev/null; python -c "import json; j=json.load(open('$O/c3_abl$a.json')); print($a, j['nprobes10_refine10']['kernel_ms_per_batch'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_c3 -- python $GRAFT_REPO_ROOT/scripts/probe_c3_search.py > $GRAFT_REPO_ROOT/$O/pmc_c3.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/pmc_sq_summary.py $O/pmc_c3 $O/c3_pmc_sq.json ivfpq_q | cut -c1-420; rm -rf $O/pmc_c3
