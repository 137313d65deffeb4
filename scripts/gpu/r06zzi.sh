# GPU call r06zzi: the magnitude fuzz (1e-3 .. 1e4, f16 columns, prefilters) on the L2 and cosine flows; sign of the device's default NaN
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06zzi; mkdir -p $O
export TMPDIR=/tmp
python -c "
import torch
x=torch.tensor([float('inf')],device='cuda'); print('inf*0 on the device:', hex((x*0).view(torch.int32).item() & 0xffffffff), ' inf-inf:', hex((x-x).view(torch.int32).item() & 0xffffffff))" 2>&1 | grep -v amdgpu.ids
timeout 500 python tests/fuzz_dot_flow.py 300 7101 --metric l2 > $O/fuzz_l2.txt 2>&1; echo "l2 fuzz rc=$?"; grep -v amdgpu.ids $O/fuzz_l2.txt | tail -12 | cut -c1-600
timeout 500 python tests/fuzz_dot_flow.py 300 7102 --metric cosine > $O/fuzz_cos.txt 2>&1; echo "cosine fuzz rc=$?"; grep -v amdgpu.ids $O/fuzz_cos.txt | tail -12 | cut -c1-600
