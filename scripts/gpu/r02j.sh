set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02j; mkdir -p $O
cd $R
timeout 300 python scripts/probe_sharded_kmeans.py 2>&1 | grep -v amdgpu.ids | tee $O/sharded_kmeans.log
timeout 400 env LANCE_BENCH_FORCE_DIST=1 python bench.py --steps 10 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err
python -c "import json,sys; r=json.loads(open('$O/bench_dist1.json').read().strip().splitlines()[-1]); print(r['build_sec'], r['multi_gpu'])" || tail -15 $O/bench_dist1.err
