set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log | cut -c1-300
timeout 600 python scripts/probe_flat_batch.py 2>&1 | grep -v amdgpu.ids | tee $O/flat_batch.log
