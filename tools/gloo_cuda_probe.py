"""Does this torch build all-reduce CUDA tensors over gloo?  (two ranks on ONE GPU: a way to run the data-parallel path at
world size 2 on a 1-GPU box)  usage: python tools/gloo_cuda_probe.py"""
import os, sys, subprocess
if 'RANK' not in os.environ:
    ps = [subprocess.Popen([sys.executable, __file__], env=dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                                                                 MASTER_PORT='29577')) for r in range(2)]
    sys.exit(max(p.wait() for p in ps))
import torch, torch.distributed as dist
r = int(os.environ['RANK'])
dist.init_process_group('gloo', rank=r, world_size=2)
t = torch.full((1000,), float(r + 1), device='cuda:0')
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    dist.all_reduce(t)
torch.cuda.current_stream().wait_stream(s)
print('rank', r, 'sum ok', bool((t == 3).all()), flush=True)
dist.destroy_process_group()
