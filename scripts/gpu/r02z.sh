# GPU call r02z: final round-2 bench line (with the CPU baseline), smoke, and the C4-shaped 100M-row run on the final kernels
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02z; mkdir -p $O
cd $R
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 > $O/bench_n1.json 2> $O/bench_n1.err; python -c "import json; r=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step'],4), r['recall_at_10'], r['kernel_ms_per_step'], r['roofline']['frac'], r['cpu_baseline']['value'], r['cpu_baseline']['build_sec'], r['build_sec'])"
timeout 200 python scripts/scale_probe.py 100000000 f16 > $O/scale_c4_100M_f16.txt 2>&1; tail -8 $O/scale_c4_100M_f16.txt | cut -c1-200
