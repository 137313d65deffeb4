# GPU call r06zzc: the skew-guard test, a fuzz campaign of the dot metric's quantised flow, C3 at 10,000-query batches on three contexts
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzc; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_dot_flow.py -x -q -m gpu --timeout 600 > $O/dot_flow.txt 2>&1; echo "dot flow rc=$?"; tail -3 $O/dot_flow.txt | cut -c1-300
timeout 900 python tests/fuzz_dot_flow.py 600 7001 > $O/fuzz_dot.txt 2>&1; echo "dot fuzz rc=$?"; grep -v amdgpu.ids $O/fuzz_dot.txt | tail -6 | cut -c1-400
timeout 600 python scripts/probe_c3_streams.py 1000000 10000 > $O/c3_streams_10k.json 2> $O/c3_streams_10k.err; echo "c3 rc=$?"; tail -c 900 $O/c3_streams_10k.json
