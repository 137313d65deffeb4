# GPU call r02w: single-stream rocprofv3 kernel stats (per-launch durations comparable with bench.py's HIP-event timing) + parity fuzz
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02w; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 > $O/prof1.log 2>&1
find $O/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats_1stream.csv
KT=$(find $O/prof1 -name "*kernel_trace.csv" | head -1)
python - "$KT" > $O/qscan_launches.txt <<'PY'
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for sub in ("ivfpq_qscan_kernel", "ivfpq_qbound_kernel", "ivfpq_qmerge_kernel", "refine_kernel", "q_residual_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if sub in r["Kernel_Name"]]
    big = [x for x in d if x > 0.5 * max(d)] if d else []
    print(sub, "launches", len(d), "full-size launches", len(big), "mean_us_full_size", round(statistics.mean(big), 1) if big else None,
          "min", round(min(big), 1) if big else None, "max", round(max(big), 1) if big else None)
PY
cat $O/qscan_launches.txt
tail -c 600 $O/prof1.log | head -c 600
rm -rf $O/*/*/*.db $O/*/*/*kernel_trace.csv 2>/dev/null
cd $R
timeout 150 python tests/fuzz_parity.py 70 31 > $O/fuzz_31.log 2>&1; tail -2 $O/fuzz_31.log
timeout 100 python tests/fuzz_parity.py 40 32 > $O/fuzz_32.log 2>&1; tail -2 $O/fuzz_32.log
