# GPU call r03a: first job of round 3 -- new tiled (M = 48/64/96) quantised scan on hardware, full GPU suite incl. the
# full-configuration golden cases, widened fuzz (3 processes), C3 grid, wide-assign-v2 A/B, clean build on the box
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03a; mkdir -p $O
nproc > $O/box.txt; rocm-smi --showproductname 2>/dev/null | head -8 >> $O/box.txt
# 1. the new kernels first
timeout 300 python -m pytest tests/test_gpu_pm_scan.py -m gpu -x -q --timeout 120 -k "tiled" > $O/pytest_tiled.log 2>&1; echo "tiled rc=$?"; tail -5 $O/pytest_tiled.log | cut -c1-300
# 2. the whole suite (not -x: see everything that fails)
timeout 900 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_all.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_all.log | cut -c1-300 | tail -15
# 3. C3 grid (with the clean build running beside it on the host cores)
( mkdir -p /tmp/cb && cd lance_amd/csrc && /usr/bin/time -v make -j16 OBJDIR=/tmp/cb/obj OUT=/tmp/cb/liblance_hip_clean.so > $GRAFT_REPO_ROOT/$O/clean_build.log 2>&1; ls -la /tmp/cb/*.so >> $GRAFT_REPO_ROOT/$O/clean_build.log ) &
timeout 400 python scripts/measure_grid.py --c3 --skip-c2 > $O/grid_c3.json 2> $O/grid_c3.err; echo "grid rc=$?"; grep -E "qps|ms_per_batch|nprobes|refine" $O/grid_c3.json | tr -d '\n' | cut -c1-1500; echo
wait
# 4. wide-assign-v2 variant against the tree's kernels
timeout 200 python scripts/probe_assign_wide.py > $O/assign_wide_main.txt 2>&1; tail -3 $O/assign_wide_main.txt | cut -c1-400
LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_widev2.so timeout 200 python scripts/probe_assign_wide.py > $O/assign_wide_v2.txt 2>&1; tail -3 $O/assign_wide_v2.txt | cut -c1-400
LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_widev2.so timeout 200 python -m pytest tests/test_zz_gpu_wide_mfma.py -m gpu -q --timeout 120 > $O/pytest_widev2.log 2>&1; tail -3 $O/pytest_widev2.log | cut -c1-300
# 5. LDS ceilings incl. the staggered code-major table
timeout 120 python scripts/ubench_probe.py > $O/ubench.json 2> $O/ubench.err; cat $O/ubench.json | cut -c1-1200
# 6. fuzz: three processes, different seeds, 5 oracle threads each
for s in 11 12 13; do OMP_NUM_THREADS=5 timeout 700 python tests/fuzz_parity.py 600 $s --log $O/fuzz_seed$s.log > $O/fuzz_seed$s.out 2>&1 & done
wait
tail -2 $O/fuzz_seed*.log | cut -c1-600
