# GPU call r04f: QPT mode 2 (tables before the bound pass), MFMA table v2 (opt-in), A/B timings
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04f; mkdir -p $O
timeout 1200 python -m pytest tests/test_zz_gpu_qpt_variant.py tests/test_zz_gpu_graph.py -m gpu -q --timeout 900 > $O/qpt_tests.txt 2>&1; echo "qpt/graph tests rc=$?"; grep -E "passed|failed|Error|assert" $O/qpt_tests.txt | cut -c1-300 | tail -8
LANCE_HIP_MFMA_TABLE=1 timeout 600 python -m pytest tests/test_gpu_pm_scan.py -m gpu -q -x -k "f32_every or two_class or random_shapes or overflow" > $O/mb_tests.txt 2>&1; echo "mb tests rc=$?"; tail -2 $O/mb_tests.txt | cut -c1-200
(LANCE_HIP_QPT=2 timeout 130 python tests/fuzz_parity.py 90 601 --log $O/fuzz_qpt2_601.txt > /dev/null 2>&1 &
 LANCE_HIP_QPT=2 timeout 130 python tests/fuzz_parity.py 90 602 --log $O/fuzz_qpt2_602.txt > /dev/null 2>&1 &
 wait)
grep -hE "SKIP|MISMATCH|ERROR|fuzz " $O/fuzz_qpt2_601.txt $O/fuzz_qpt2_602.txt | cut -c1-300 | tail -8
LANCE_HIP_QPT=1 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt1.json 2> $O/c3_qpt1.err; tail -1 $O/c3_qpt1.json | cut -c1-1400
LANCE_HIP_QPT=2 timeout 300 python scripts/probe_c3_search.py > $O/c3_qpt2.json 2> $O/c3_qpt2.err; tail -1 $O/c3_qpt2.json | cut -c1-1400; tail -2 $O/c3_qpt2.err
LANCE_HIP_QPT=2 LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_c3_search.py > /dev/null 2> $O/c3_qpt2_stats.err; grep qscan $O/c3_qpt2_stats.err | tail -3
LANCE_HIP_QPT=1 LANCE_HIP_Q_STATS=1 timeout 300 python scripts/probe_c3_search.py > /dev/null 2> $O/c3_qpt1_stats.err; grep qscan $O/c3_qpt1_stats.err | tail -3
LANCE_HIP_MFMA_TABLE=1 timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_mb.json 2> $O/bench_mb.err; python -c "
import json; j=json.loads(open('$O/bench_mb.json').read().strip().splitlines()[-1]); print('MB', j['value'], j['kernel_ms_per_step'])"
timeout 300 python bench.py --no-pmc --no-cpu-baseline > $O/bench_def.json 2> $O/bench_def.err; python -c "
import json; j=json.loads(open('$O/bench_def.json').read().strip().splitlines()[-1]); print('DEF', j['value'], j['kernel_ms_per_step'])"
