# GPU call r05l: cosine flat: bf16 plane out of the row-norm pass (tests + C3-shaped timing); PQ E-step rows per workgroup A/B (variant builds) on the build time
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05l; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests/test_zz_gpu_flat_wide.py tests/test_gpu_parity.py -m gpu -q --timeout 800 -k "flat or cosine or wide" > $O/tests.txt 2>&1; echo "tests rc=$? $(( $(date +%s)-t0 )) s"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.txt | cut -c1-300 | tail -8; grep -E "^E  " $O/tests.txt | head -8 | cut -c1-300
timeout 400 python scripts/probe_flat_wide.py > $O/probe.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu $O/probe.txt | cut -c1-330
brief() { python -c "
import json,sys; j=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', round(j['value']), 'build', j['build_sec'], j['roofline_build']['build_stages_ms'])" | cut -c1-400; }
B="--steps 50 --warmup 5 --no-pmc --no-cpu-baseline --no-grid"
timeout 300 python bench.py $B > $O/bench_rg4.json 2> $O/bench_rg4.err; brief $O/bench_rg4.json RG4
LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_pqrg8.so timeout 300 python bench.py $B > $O/bench_rg8.json 2> $O/bench_rg8.err; brief $O/bench_rg8.json RG8
LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_pqrg16.so timeout 300 python bench.py $B > $O/bench_rg16.json 2> $O/bench_rg16.err; brief $O/bench_rg16.json RG16
