# GPU call r03o: tiled-scan default shape (512 lanes x 4 rows x 3 tiles) through its tests + full-config goldens; more tile shapes;
# case 41/45 after the generator fix; fresh fuzz seeds
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pm_scan.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_fuzz_findings.py -m gpu -q --timeout 400 > $O/tiled_tests.txt 2>&1; echo "rc=$?"; tail -3 $O/tiled_tests.txt | cut -c1-300
timeout 100 python tests/fuzz_parity.py 100 41 --case 45 > $O/fuzz_41_45.txt 2>&1; grep -E "SKIP|MISMATCH|ERROR|fuzz " $O/fuzz_41_45.txt | cut -c1-300
timeout 200 python scripts/probe_c3_search.py > $O/c3_default.json 2>/dev/null; python -c "import json; j=json.load(open('$O/c3_default.json')); print('default', j['nprobes10_refine10']['wall_ms_per_batch'], j['nprobes10_refine10']['kernel_ms_per_batch']); print(j['nprobes10_refine0']['wall_ms_per_batch'], j['nprobes50_refine10']['wall_ms_per_batch'])"
for v in e f g h; do LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qtshape$v.so timeout 200 python scripts/probe_c3_search.py > $O/c3_shape$v.json 2>/dev/null; python -c "import json; j=json.load(open('$O/c3_shape$v.json')); print('$v', j['nprobes10_refine10']['wall_ms_per_batch'], j['nprobes10_refine10']['kernel_ms_per_batch'])"; done
(timeout 260 python tests/fuzz_parity.py 200 51 --log $O/fuzz_51.txt > /dev/null 2>&1 &
 timeout 260 python tests/fuzz_parity.py 200 52 --log $O/fuzz_52.txt > /dev/null 2>&1 &
 wait)
grep -hE "SKIP|MISMATCH|ERROR|fuzz " $O/fuzz_51.txt $O/fuzz_52.txt | cut -c1-400
