# GPU call r03c: bisect the flaky query-major mismatch found by the fuzz (rows missing for a few late queries of large batches)
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03c; mkdir -p $O
for i in 1 2 3; do timeout 240 python tests/fuzz_parity.py 200 11 --case 4 --debug > $O/fuzz_debug_11_4_run$i.txt 2>&1; grep -E "debug|MISMATCH|fuzz " $O/fuzz_debug_11_4_run$i.txt | cut -c1-500; done
for sc in "13 4" "13 86" "13 63"; do set -- $sc; timeout 240 python tests/fuzz_parity.py 200 $1 --case $2 --debug > $O/fuzz_debug_$1_$2.txt 2>&1; grep -E "debug|MISMATCH|fuzz " $O/fuzz_debug_$1_$2.txt | cut -c1-500; done
