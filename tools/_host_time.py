import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from advmix_amd.graph import AdvMixGraphRunner
dev = torch.device('cuda:0')
cfg, D, G, T, crit, optD, optG = bench.build_models('hrnet_w32', dev)
args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
views, tgt, tw = bench.synth(32, 17, 256, 192, dev, 1234)
r = AdvMixGraphRunner(args, D, G, T, crit, optD, optG, views, tgt, tw)
print('graphs', r.seq.n_graphs)
for _ in range(5):
    r.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
if r.seq.tape is not None:
    r.seq.tape.check()
print('host issue time per step %.2f ms; wall per step %.2f ms' % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
