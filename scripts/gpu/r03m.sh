# GPU call r03m: from-source build ON THE BOX + tests against it; C4-shaped bench mode (fixed data scaling); closing fuzz run on the final tree
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03m; mkdir -p $O
mkdir -p /tmp/cb
( cd lance_amd/csrc && bash -c "time make -j16 OBJDIR=/tmp/cb/obj OUT=/tmp/cb/liblance_hip_clean.so" ) > $O/clean_build.log 2>&1; tail -4 $O/clean_build.log; ls -la /tmp/cb/liblance_hip_clean.so >> $O/clean_build.log
LANCE_HIP_LIB=/tmp/cb/liblance_hip_clean.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pm_scan.py tests/test_zz_gpu_fuzz_findings.py -m gpu -q --timeout 300 > $O/pytest_clean_build.log 2>&1; echo "clean-build tests rc=$?"; tail -2 $O/pytest_clean_build.log | cut -c1-200
grep -c liblance_hip_clean /proc/self/maps; python -c "
import os; os.environ['LANCE_HIP_LIB']='/tmp/cb/liblance_hip_clean.so'
import lance_amd; e=lance_amd.default_engine(); print([l.split()[-1] for l in open('/proc/self/maps') if 'liblance_hip' in l][:1])" >> $O/clean_build.log 2>&1; tail -1 $O/clean_build.log
timeout 300 python bench.py --config c4 --n 8000000 --no-cpu-baseline --steps 5 > $O/bench_c4_8M.json 2> $O/bench_c4.err; python -c "import json; j=json.loads(open('$O/bench_c4_8M.json').read().strip().splitlines()[-1]); print('c4 8M', j['value'], j['ms_per_step'], j['build_sec'], j['recall_at_10'], j['kernel_ms_per_step'])"
for s in 41 42 43; do OMP_NUM_THREADS=5 OMP_WAIT_POLICY=PASSIVE timeout 560 python tests/fuzz_parity.py 460 $s --watchdog 200 --log $O/fuzz_seed$s.log > $O/fuzz_seed$s.out 2> $O/fuzz_seed$s.err & done
wait
for s in 41 42 43; do tail -n 3 $O/fuzz_seed$s.log | cut -c1-500; done
