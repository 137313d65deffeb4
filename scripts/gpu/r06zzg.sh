# GPU call r06zzg: SIFT-like rows as they are (uneven lists) through the dot flow with the skew guard off, bound pass over 2 / 3 lists
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzg; mkdir -p $O
export TMPDIR=/tmp
for b in 2 3; do
  echo "lists $b"; LANCE_HIP_DOT_FLOW_SKEW=1e18 LANCE_HIP_DOT_BOUND_LISTS=$b timeout 300 python scripts/probe_dot_flow.py child 2>&1 | grep -v amdgpu.ids | grep -A1 "as-is   dot" | cut -c1-400 | tee -a $O/asis_$b.txt
done
LANCE_HIP_DOT_FLOW_SKEW=1e18 LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_dot_flow.py child 2>&1 | grep "qscan\]" | uniq -c | sed -n 2p | cut -c1-300
