# GPU call r02c: quantised flow after the overflow-rescan / ILP / direct-mode fixes
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['exact_replays_last_step'], r['kernel_ms_per_step'])" || tail -5 $O/bench_$name.err
}
run q A=1
run old LANCE_HIP_NO_QSCAN=1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_lds -- python $R/bench.py --steps 5 --no-cpu-baseline > $O/pmc_lds.log 2>&1
python $R/scripts/pmc_sq_summary.py $O/pmc_lds $O/pmc_lds_summary.json ivfpq_ | tail -8
rm -rf $O/pmc_lds/*/*.db 2>/dev/null
du -sh $O
