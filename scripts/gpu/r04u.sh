# GPU call r04u: merge-kernel variants (codebook entries fetched together: 4 / 8 / 16; waves per SIMD hint 6 / 4) on the C2 bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04u; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-pmc --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/b_$tag.json 2>$O/b_$tag.err; python -c "
import json; j=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(j['value']), j['recall_at_10'], j['exact_replays_last_step'], j['kernel_ms_per_step']['ivfpq_scan_c1'], j['kernel_ms_per_step']['ivfpq_merge'], j['kernel_ms_per_step']['refine'])"; }
run default X=1
run mpf8 LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qm_mpf8.so
run mpf8_w4 LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qm_mpf8_w4.so
run mpf16_w4 LANCE_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/liblance_hip_qm_mpf16_w4.so
run streams2 X=1 -- 2>/dev/null || true
timeout 200 $B --streams 2 > $O/b_s2.json 2>/dev/null; python -c "
import json; j=json.loads(open('$O/b_s2.json').read().strip().splitlines()[-1]); print('streams2', round(j['value']))"
timeout 200 $B --streams 4 > $O/b_s4.json 2>/dev/null; python -c "
import json; j=json.loads(open('$O/b_s4.json').read().strip().splitlines()[-1]); print('streams4', round(j['value']))"
timeout 200 $B --streams 1 > $O/b_s1.json 2>/dev/null; python -c "
import json; j=json.loads(open('$O/b_s1.json').read().strip().splitlines()[-1]); print('streams1', round(j['value']))"
