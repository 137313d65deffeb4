# GPU call r06zh: per-group records of four tiles leave the sweep together: probe, coarse parity, fullconfig, C5 at 100M rows
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zh; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/probe_coarse.py > $O/probe.txt 2>&1; echo "probe rc=$?"; grep -E "^nlist" $O/probe.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_zz_gpu_coarse_mfma.py tests/test_zz_gpu_fullconfig.py tests/test_zz_gpu_xform_fused.py -x -q -m gpu --timeout 1400 > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $O/tests.txt | cut -c1-600 | tail -12
timeout 900 python bench.py --config c5 --n 100000000 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline --no-grid --no-extras > $O/c5_100m.json 2> $O/c5_100m.err; echo "c5 rc=$?"
python -c "
import json; j=json.loads(open('$O/c5_100m.json').read().strip().splitlines()[-1]); print('C5 100M', j['value'], j['ms_per_step'], j['recall_at_10'], j['build_sec'], j['kernel_ms_per_step'])" 2>&1 | cut -c1-700
