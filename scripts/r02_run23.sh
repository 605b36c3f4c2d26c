#!/bin/bash
# MSDA kernel parity + a timing at the UPN encoder geometry
OUT=$(pwd)/gpurun_out/r02_run23; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 600 > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
python - <<'P' 2>&1 | grep -v amdgpu
import torch, time, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from vlm_fo1_amd import ops
shapes=[(100,167),(50,84),(25,42),(13,21),(7,11)]
start=[0]
for h,w in shapes[:-1]: start.append(start[-1]+h*w)
S=sum(h*w for h,w in shapes)
sh=torch.tensor(shapes).cuda(); ls=torch.tensor(start).cuda()
for name,Lq,dt in (("encoder fp32",S,torch.float32),("decoder fp32",900,torch.float32),("encoder bf16",S,torch.bfloat16)):
    v=(torch.rand(1,S,8,32)*2-1).to(dt).cuda(); loc=torch.rand(1,Lq,8,5,4,2).cuda(); w=torch.softmax(torch.rand(1,Lq,8,20),-1).view(1,Lq,8,5,4).cuda()
    for _ in range(3): ops.ms_deform_attn(v,sh,ls,loc,w)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): ops.ms_deform_attn(v,sh,ls,loc,w)
    torch.cuda.synchronize(); us=(time.perf_counter()-t)/20*1e6
    esz=v.element_size(); gathered=Lq*8*20*4*32*esz
    print(f"msda {name}: Lq={Lq} {us:.1f} us; taps gathered {gathered/1e6:.0f} MB -> {gathered/us/1e3:.0f} GB/s (L2-resident value {S*256*esz/1e6:.1f} MB)")
P
