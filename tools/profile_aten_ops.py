import sys, types, torch
sys.path.insert(0, '/root/repo')
import bench
from advmix_amd.core.function import advmix_step
from advmix_amd.core.evaluate import accuracy
dev = torch.device('cuda:0')
cfg, D, G, T, crit, optD, optG = bench.build_models('hrnet_w32', dev)
args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
views, tgt, tw = bench.synth(32, 17, 256, 192, dev, 1234)
for _ in range(2):
    advmix_step(args, D, G, T, crit, optD, optG, views, tgt, tw, None)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    advmix_step(args, D, G, T, crit, optD, optG, views, tgt, tw, None)
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
rows = [(e.key, e.count, str(e.input_shapes)[:90]) for e in ev if e.key.startswith('aten::') and e.count >= 5]
rows.sort(key=lambda r: -r[1])
for r in rows[:40]:
    print('%6d  %-28s %s' % (r[1], r[0], r[2]))
