# GPU call r03w: C3-shaped search on 1 / 2 / 3 engine contexts
set -x
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03w; mkdir -p $O
timeout 300 python scripts/probe_c3_streams.py > $O/c3_streams.json 2> $O/c3_streams.err; cat $O/c3_streams.json | cut -c1-1500; tail -3 $O/c3_streams.err | cut -c1-300
