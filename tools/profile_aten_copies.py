#!/usr/bin/env python3
"""Which torch calls of one eager AdvMix step turn into device-to-device copies (hipMemcpyAsync = rocclr copyBuffer
launches)?  Uses torch.profiler over one step and lists aten::copy_ / clone / contiguous call sites."""
import os, sys, types, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from advmix_amd.core.function import advmix_step
dev = torch.device('cuda:0')
cfg, D, G, T, crit, optD, optG = bench.build_models('hrnet_w32', dev)
args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
views, tgt, tw = bench.synth(32, 17, 256, 192, dev, 1234)
advmix_step(args, D, G, T, crit, optD, optG, views, tgt, tw)
torch.cuda.synchronize()
sites = collections.Counter()
orig_copy = torch.Tensor.copy_
orig_clone = torch.Tensor.clone
orig_contig = torch.Tensor.contiguous


def where():
    st = traceback.extract_stack(limit=6)[:-2]
    return ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-3:]))


def copy_(self, *a, **k):
    sites['copy_ ' + where()] += 1
    return orig_copy(self, *a, **k)


def clone(self, *a, **k):
    sites['clone ' + where()] += 1
    return orig_clone(self, *a, **k)


def contiguous(self, *a, **k):
    r = orig_contig(self, *a, **k)
    if r.data_ptr() != self.data_ptr():
        sites['contiguous(copy) ' + where()] += 1
    return r


torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = copy_, clone, contiguous
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
    advmix_step(args, D, G, T, crit, optD, optG, views, tgt, tw)
    torch.cuda.synchronize()
torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = orig_copy, orig_clone, orig_contig
print('python-level call sites:')
for k, v in sites.most_common(15):
    print('  %5d  %s' % (v, k))
ops = collections.Counter()
for e in prof.events():
    if e.name.startswith('aten::') or 'Memcpy' in e.name or 'memcpy' in e.name:
        ops[e.name] += 1
print('aten ops of one step:')
for k, v in ops.most_common(25):
    print('  %5d  %s' % (v, k))
