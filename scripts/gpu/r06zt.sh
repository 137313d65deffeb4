# GPU call r06zt: where a C2-shaped dot batch spends its time (quantised flow vs exact pair scan; SIFT-like rows as they are / centred)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zt; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/probe_dot_flow.py > $O/dot_probe.txt 2>&1; echo "rc=$?"; cat $O/dot_probe.txt | cut -c1-600
LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_dot_flow.py child 2>&1 | grep "qscan\]" | sort | uniq -c | sort -rn | head -8 | cut -c1-400
