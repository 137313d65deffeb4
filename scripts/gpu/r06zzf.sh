# GPU call r06zzf: A/B of the number of lists the dot bound pass covers
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzf; mkdir -p $O
export TMPDIR=/tmp
for b in 1 2 3 4; do
  echo "lists $b"; LANCE_HIP_DOT_BOUND_LISTS=$b timeout 300 python scripts/probe_dot_flow.py child 2>&1 | grep -v amdgpu.ids | grep -A1 "centred dot\|unit    dot" | cut -c1-400 | tee -a $O/bound_lists_$b.txt
done
