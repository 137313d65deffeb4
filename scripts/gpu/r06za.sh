# GPU call r06za: find_partitions over thousands of lists on per-group keys instead of the [nq][nlist] surrogate matrix: parity, then C5 / C4 at 100M rows (and the A/B by switch)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06za; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests/test_zz_gpu_coarse_mfma.py -x -q -m gpu --timeout 1400 > $O/tests_coarse.txt 2>&1; echo "coarse tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/tests_coarse.txt | cut -c1-600 | tail -12
timeout 900 python -m pytest tests/test_zz_gpu_fullconfig.py tests/test_gpu_parity.py tests/test_zz_gpu_two_ranks.py -x -q -m gpu --timeout 900 > $O/tests_more.txt 2>&1; echo "more tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_more.txt | cut -c1-300 | tail -6
for sw in on off; do
  if [ $sw = off ]; then export LANCE_HIP_COARSE_GROUPS=0; else unset LANCE_HIP_COARSE_GROUPS; fi
  timeout 900 python bench.py --config c5 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c5_100m_$sw.json 2> $O/c5_100m_$sw.err; echo "c5 $sw rc=$?"
  python -c "
import json; j=json.loads(open('$O/c5_100m_$sw.json').read().strip().splitlines()[-1]); print('C5 $sw', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
done
unset LANCE_HIP_COARSE_GROUPS
timeout 600 python bench.py --config c4 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c4_100m.json 2> $O/c4_100m.err; echo "c4 rc=$?"
python -c "
import json; j=json.loads(open('$O/c4_100m.json').read().strip().splitlines()[-1]); print('C4', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
timeout 600 python tests/fuzz_parity.py 150 6203 --log $O/fuzz.txt --watchdog 300 > $O/fuzz_out.txt 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_out.txt | cut -c1-300
