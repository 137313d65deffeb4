# GPU call r02o: 4-workgroups-per-CU scan variant, MFMA counters for the assign / flat kernels
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O
cd $R
run() { # name, env... -- args
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --no-cpu-baseline --streams 3 > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['kernel_ms_per_step'])" || tail -15 $O/bench_$name.err
}
run w6 A=1
run w8u1 LANCE_HIP_LIB=$R/build/variants/liblance_hip_w8u1.so
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma" | head -20 > $O/mfma_counters.txt; cat $O/mfma_counters.txt | cut -c1-200
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_mfma_assign -- python $R/scripts/probe_assign.py mfma > $O/pmc_mfma_assign.log 2>&1
python $R/scripts/pmc_sq_summary.py $O/pmc_mfma_assign $O/pmc_mfma_assign_summary.json ma_ | tail -6
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_mfma_flat -- python $R/scripts/probe_flat_batch.py mfma > $O/pmc_mfma_flat.log 2>&1
python $R/scripts/pmc_sq_summary.py $O/pmc_mfma_flat $O/pmc_mfma_flat_summary.json flat_filter_mfma | tail -4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_flat -- python $R/scripts/probe_flat_batch.py mfma > $O/prof_flat.log 2>&1
find $O/prof_flat -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {} | cut -c1-220
rm -rf $O/*/*/*.db $O/*/*/*kernel_trace.csv 2>/dev/null
du -sh $O
