set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O
cd $R
timeout 200 python tests/fuzz_parity.py 100 11 > $O/fuzz_11.log 2>&1; tail -3 $O/fuzz_11.log
timeout 200 python tests/fuzz_parity.py 80 12 > $O/fuzz_12.log 2>&1; tail -3 $O/fuzz_12.log
