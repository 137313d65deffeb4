# GPU call r06zq: what the dot metric costs without a quantised flow (C2 shape, one context), and the C1 single-query figure
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zq; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/probe_metrics.py > $O/metrics.txt 2>&1; echo "rc=$?"; cat $O/metrics.txt | cut -c1-300
