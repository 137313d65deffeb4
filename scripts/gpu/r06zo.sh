# GPU call r06zo: (1) C2 with the per-group keys taken from 256 lists on (A/B against the default 1024); (2) C5's shape at ONE BILLION rows on the final tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zo; mkdir -p $O
export TMPDIR=/tmp
for sw in default 256; do
  if [ $sw = 256 ]; then export LANCE_HIP_COARSE_GROUPS=256; else unset LANCE_HIP_COARSE_GROUPS; fi
  for rep in 1 2; do
  timeout 600 python bench.py --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c2_$sw_$rep.json 2> $O/c2_$sw.err; python -c "
import json; j=json.loads(open('$O/c2_$sw_$rep.json').read().strip().splitlines()[-1]); print('C2 groups-from $sw', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step']['dist_matrix'], j['kernel_ms_per_step']['select_probes'])" 2>&1 | cut -c1-300
  done
done
unset LANCE_HIP_COARSE_GROUPS
t0=$(date +%s)
timeout 2400 python bench.py --config c5 --n 1000000000 --nprobes 32 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-extras > $O/bench_c5_1B.json 2> $O/bench_c5_1B.err; echo "c5 1B rc=$? $(( $(date +%s)-t0 )) s"
python -c "
import json; j=json.loads(open('$O/bench_c5_1B.json').read().strip().splitlines()[-1]); print('C5-1B', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['build_stages_ms'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-900
tail -3 $O/bench_c5_1B.err | cut -c1-300
