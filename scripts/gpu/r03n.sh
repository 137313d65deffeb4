# GPU call r03n: closing-fuzz findings (f16 dot) replayed with diagnostics; tile-shape variants of the tiled scan
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03n; mkdir -p $O
for sc in "41 21" "43 53"; do set -- $sc; timeout 200 python tests/fuzz_parity.py 100 $1 --case $2 > $O/fuzz_$1_$2.txt 2>&1; grep -E "MISMATCH|ERROR|fuzz " $O/fuzz_$1_$2.txt | cut -c1-300; done
for i in 1 2; do timeout 240 python tests/fuzz_parity.py 200 41 --case 45 --debug > $O/fuzz_41_45_run$i.txt 2>&1; grep -E "debug|MISMATCH|fuzz " $O/fuzz_41_45_run$i.txt | cut -c1-600; done
LANCE_HIP_NO_PM=1 timeout 240 python tests/fuzz_parity.py 200 41 --case 45 --debug > $O/fuzz_41_45_nopm.txt 2>&1; grep -E "debug|MISMATCH|fuzz " $O/fuzz_41_45_nopm.txt | cut -c1-400 | head -6
for v in a b c; do LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qtshape$v.so timeout 200 python scripts/probe_c3_search.py > $O/c3_shape$v.json 2>/dev/null; python -c "import json; j=json.load(open('$O/c3_shape$v.json')); print('$v', j['nprobes10_refine10']['wall_ms_per_batch'], j['nprobes10_refine10']['kernel_ms_per_batch'])"; done
