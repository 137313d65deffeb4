# GPU call r03zz: last seconds of the round's budget: a fuzz burst on the final tree (default path)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03zz; mkdir -p $O
(timeout 40 python tests/fuzz_parity.py 28 91 --log $O/fuzz_91.txt > /dev/null 2>&1 &
 timeout 40 python tests/fuzz_parity.py 28 92 --log $O/fuzz_92.txt > /dev/null 2>&1 &
 wait)
grep -hE "SKIP|MISMATCH|ERROR|fuzz " $O/fuzz_91.txt $O/fuzz_92.txt | cut -c1-300
