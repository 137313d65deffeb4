# GPU call r06zzd: triage of the dot fuzz failures (f16 columns): the same cases with the quantised flow and on the exact pair scan
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzd; mkdir -p $O
export TMPDIR=/tmp
for c in 25 89 104 192 254 332; do
  timeout 120 python tests/fuzz_dot_flow.py 1 7001 --case $c 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tee -a $O/flow.txt
  LANCE_HIP_NO_DOT_FLOW=1 timeout 120 python tests/fuzz_dot_flow.py 1 7001 --case $c 2>&1 | grep -v amdgpu.ids | grep "dot fuzz\|FAIL" | sed 's/^/   [exact pair scan] /' | cut -c1-300 | tee -a $O/exact.txt
done
