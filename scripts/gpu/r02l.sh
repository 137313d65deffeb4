set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02l; mkdir -p $O
cd $R
timeout 900 python scripts/scale_probe.py 20000000 f16 > $O/scale_f16_20M.txt 2>&1; tail -12 $O/scale_f16_20M.txt
timeout 1200 python scripts/scale_probe.py 100000000 f16 > $O/scale_f16_100M.txt 2>&1; tail -12 $O/scale_f16_100M.txt
