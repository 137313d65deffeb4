# GPU call r05p: C3 build with and without HIP-graph replay of the Lloyd iteration blocks (train_ivf 256 ms in round 4 with graphs on, 341 ms now)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05p; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/probe_c3_build.py 2>/dev/null | tail -1
LANCE_HIP_KMEANS_GRAPH=1 timeout 300 python scripts/probe_c3_build.py 2>/dev/null | tail -1
LANCE_HIP_KMEANS_CHECK=16 timeout 300 python scripts/probe_c3_build.py 2>/dev/null | tail -1
