# GPU call r02za: flat scan reading f16 / int8 rows natively -- full suite + the C4-shaped 100M-row f16 run
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02za; mkdir -p $O
cd $R
timeout 200 python -m pytest tests -m gpu -x -q --timeout 150 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-300
timeout 150 python scripts/scale_probe.py 100000000 f16 > $O/scale_c4_100M_f16.txt 2>&1; tail -8 $O/scale_c4_100M_f16.txt | cut -c1-200
