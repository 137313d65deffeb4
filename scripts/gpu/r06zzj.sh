# GPU call r06zzj: find_partitions over a model with infinite centroids (f16 k-means that overflowed): device vs oracle
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python scripts/probe_nonfinite_model.py 89 gpu 2>&1 | grep -v amdgpu.ids | cut -c1-900
