#!/usr/bin/env python3
"""The U-Net's Winograd-domain GEMMs ([16 x rows x K] . [16 x K x N], csrc/conv_wino4.hip -> advmix_conv_direct_gemm_batched)
against the vendor library's strided-batched fp32 GEMM (torch.bmm -> rocBLAS / hipBLASLt) on the same shapes: where a plain
library GEMM would beat conv_direct's main loop (EXPERIMENTS M8).  The in-step times are profiles/r06k_per_shape_1lane.csv's."""
import torch, time
dev='cuda:0'
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
# (rows, K, N) of the U-Net's Winograd-domain GEMMs at B = 32 (per-shape table: conv_direct 1x1 over 16 images)
for rows,K,N,us in ((11264,256,256,218.0),(12032,256,256,235.8),(3200,512,512,240.0),(2816,512,512,202.6),(1024,1024,1024,287.3),(768,1024,1024,221.8),(384,1024,2048,217.0),(256,2048,1024,160.5),(12032,128,256,130.5),(3200,256,512,126.1),(1024,512,1024,146.3)):
    a=torch.randn(16,rows,K,device=dev); b=torch.randn(16,N,K,device=dev)
    us_b=t(lambda: torch.bmm(a,b.transpose(1,2)))
    fl=2*16*rows*K*N
    print('rows %6d K %5d N %5d: torch.bmm (rocBLAS / hipBLASLt) %7.1f us = %6.1f TF/s | conv_direct in the step %7.1f us = %6.1f TF/s' % (rows,K,N,us_b,fl/us_b/1e6,us,fl/us/1e6))
