# GPU call r04q: rotated tile loop (next tile's operand requested behind the MFMAs); MFMA counters of the build's matrix-core kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04q; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 300 python -m pytest tests/test_zz_gpu_mscan.py tests/test_gpu_pm_scan.py -m gpu -q --timeout 600 > $O/mscan_tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; tail -2 $O/mscan_tests.txt | cut -c1-300
B="python bench.py --no-pmc --no-cpu-baseline"
timeout 200 $B > $O/b0.json 2>$O/b0.err; python -c "
import json; j=json.loads(open('$O/b0.json').read().strip().splitlines()[-1]); print('DEF', j['value'], j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step'])"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/$O/pmc_m -- python $R/bench.py --steps 5 --no-cpu-baseline --no-pmc > $R/$O/pmc_m.log 2>&1
cd $R
python scripts/pmc_sq_summary.py $O/pmc_m $O/mfma_kernels_pmc.json "lh::" | grep -i "ma_top3\|pq_mfma\|mscan\|flat_filter" | cut -c1-500
rm -rf $O/pmc_m
