# GPU call r06zzzz: validation of the final round-6 tree (slice height 3072): full -m gpu suite, smoke(), the bench line in the driver's form, rocprofv3 kernel stats of the same command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzzz; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/gpu_suite.txt | cut -c1-300 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-300
t0=$(date +%s)
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); r=j['roofline']; print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['build_sec_pcie_inclusive'], j['kernel_ms_per_step']); print('ROOF', r['kernel'][:40], r['bound'], r['achieved'], r['frac'], r['frac_algorithmic'], r['peak_measured'], r['traffic'], r['avg_launch_ms']); print('CPU', j['cpu_baseline']['value'], j['cpu_baseline']['ids_equal_gpu'])"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 50 --warmup 5 --no-pmc --no-cpu-baseline --no-grid --no-extras > $R/$O/bench_prof.json 2> $R/$O/bench_prof.err); echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/bench_kernel_stats.csv; head -6 $O/bench_kernel_stats.csv | cut -c1-170; fi
timeout 300 python tests/fuzz_parity.py 100 6501 --log $O/fuzz.txt --watchdog 200 > $O/fuzz_out.txt 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_out.txt | cut -c1-300
