# GPU call r05k: build wall time step by step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05k; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/probe_build_steps.py > $O/build_steps.json 2> $O/build_steps.err; echo "rc=$?"; tail -1 $O/build_steps.json | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read()).items(): print(' ', k, v)"
tail -3 $O/build_steps.err | cut -c1-300
