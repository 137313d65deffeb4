# GPU call r04zk: synchronous searches of very large batches go through the library in slices of queries (DeviceIndex.MAX_PAIRS_PER_CALL)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_zz_gpu_mscan.py -m gpu -q -k "slices or every_instantiation" --timeout 300 2>&1 | tail -2 | cut -c1-300
