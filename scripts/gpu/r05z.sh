# GPU call r05z: SURVEY 8(d) measurement grid (C1 flat, C2 recall / QPS grid, small batches, IVF_FLAT, C3 build + search + flat) on the final round-5 tree
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05z; mkdir -p $O
export TMPDIR=/tmp
timeout 700 python scripts/measure_grid.py --c3 > $O/grid.json 2> $O/grid.err; echo "grid rc=$?"; tail -3 $O/grid.err | cut -c1-300
python -c "
import json; j=json.load(open('$O/grid.json'))
print('copy', j['peak_measured']); print('c1 single', j['c1_roofline_single_query']['achieved'], j['c1_roofline_single_query']['frac']); print('c1 batch', j['c1_roofline_batch_10k']['achieved'], j['c1_roofline_batch_10k'].get('frac_executed'))
print('c1_flat', j['c1_flat']['by_batch_size']); print('c2_build', j['c2_build']['sec']); print('c2_small', j['c2_small_batches'])
c3=j['c3']; print('c3 build', c3['build_sec'], c3['stages_ms']); print('c3 flat 1000q ms', c3['flat_1000q_ms']); print('c3 grid', c3['grid']); print('c3 roof', c3.get('roofline'))" | cut -c1-1500
