# GPU call r04x: grouping (histogram + scans) and k-means M-step (stats + accumulate + control) each as ONE launch (last-arriving workgroup
# finishes the job): full suite (training is compared bit for bit), build / search A/B against the multi-launch forms
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04x; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > $O/gpu_suite.txt 2>&1; echo "suite rc=$? $(( $(date +%s)-t0 )) s"; tail -4 $O/gpu_suite.txt | cut -c1-300
B="python bench.py --no-pmc --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/b_$tag.json 2>$O/b_$tag.err; python -c "
import json; j=json.loads(open('$O/b_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(j['value']), j['recall_at_10'], j['build_sec'], j['build_stages_ms'], j['kernel_ms_per_step']['pm_group'])"; }
run fused X=1
run unfused LANCE_HIP_KMEANS_FUSED=0 LANCE_HIP_GROUP_FUSED=0
run fused2 X=1
