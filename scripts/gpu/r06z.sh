# GPU call r06z: wall time of the hierarchical IVF trainer against the number of engine contexts (C4 shape: 4096 lists over 1M x 128; C3 shape: 1024 lists over 262k x 1536)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06z; mkdir -p $O
export TMPDIR=/tmp
for c in 1 2 4 8 16; do LANCE_HIP_HIER_CONTEXTS=$c timeout 300 python scripts/probe_hier.py > $O/hier_c4_$c.txt 2>&1; tail -1 $O/hier_c4_$c.txt | cut -c1-300; done
for c in 1 4 8 16; do HIER_K=1024 HIER_D=1536 LANCE_HIP_HIER_CONTEXTS=$c timeout 300 python scripts/probe_hier.py > $O/hier_c3_$c.txt 2>&1; tail -1 $O/hier_c3_$c.txt | cut -c1-300; done
