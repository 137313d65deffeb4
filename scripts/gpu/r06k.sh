# GPU call r06k: bench line with the round-6 measurement fields; C3-shaped transform kernel breakdown
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06k; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python -c "
import json; j=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); r=j['roofline']; print('BENCH', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['build_sec_pcie_inclusive'], j['qps_f32_refine_source']); print('ROOF', r['frac'], r['frac_algorithmic'], r['peak_measured'], r['frac_of_peak_measured'], r['traffic'], r['avg_launch_ms']); print('TAIL', json.dumps(j['roofline_tail']['bound_pass'])[:600]); print('ESTEP', j['roofline_build']['estep_ivf']['peak_measured'])" 2>&1 | cut -c1-900
(cd /tmp && PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $R/scripts/probe_xform_c3.py > $R/$O/c3_run.txt 2>&1); echo "rocprof rc=$?"
grep C3_XFORM $O/c3_run.txt | cut -c1-300
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cp "$f" $O/c3_xform_kernel_stats.csv; head -14 $O/c3_xform_kernel_stats.csv | cut -c1-230; fi
