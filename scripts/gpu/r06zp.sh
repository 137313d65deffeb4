# GPU call r06zp: fuzz campaigns on the final tree: (1) per-group keys forced from 256 lists on and the matrix-core coarse quantiser forced for every shape it takes
# (the fuzz draws at most 600 lists: by default its find_partitions calls never reach the per-group route); (2) the default routes, another seed
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zp; mkdir -p $O
export TMPDIR=/tmp
LANCE_HIP_MFMA_COARSE=1 LANCE_HIP_COARSE_GROUPS=256 timeout 800 python tests/fuzz_parity.py 700 6401 --log $O/fuzz_groups.txt --watchdog 300 > $O/fuzz_groups_out.txt 2>&1; echo "fuzz groups rc=$?"; tail -1 $O/fuzz_groups_out.txt | cut -c1-300
timeout 800 python tests/fuzz_parity.py 700 6402 --log $O/fuzz_default.txt --watchdog 300 > $O/fuzz_default_out.txt 2>&1; echo "fuzz default rc=$?"; tail -1 $O/fuzz_default_out.txt | cut -c1-300
