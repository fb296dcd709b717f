"""Times advmix_norm_bwd (partial sums -> finalize -> apply) at the HRNet-W32 B=32 shapes.
Round-1 record: a two-launch variant (fp64 atomic slots instead of the partial/finalize pair) measured
45.6 vs 23.8 us at rows 98304 x C 32 and 182.6 vs 131.9 us at C 256 - every block's atomics land at the
end of the same wave of blocks and serialise per address at the memory side - and was dropped."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from advmix_amd._lib import call, lib

d = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr() if t is not None else None
for (B, C, H, W) in [(32, 32, 64, 48), (32, 64, 32, 24), (32, 128, 16, 12), (32, 256, 8, 6), (32, 256, 64, 48)]:
    rows = B * H * W
    x = torch.randn(rows, C, device=d); dy = torch.randn(rows, C, device=d); y = torch.relu(x)
    mean = x.mean(0).contiguous(); invstd = (x.var(0, unbiased=False) + 1e-5).rsqrt().contiguous()
    gamma = torch.ones(C, device=d); dx = torch.empty_like(x); dg = torch.zeros(C, device=d); db = torch.zeros(C, device=d)
    ws3 = torch.zeros(lib.advmix_norm_ws_bytes(1, C) // 4 + 16, device=d)
    res = {}
    for name in ('three',):
        def run():
            call('advmix_norm_bwd', p(dy), p(y), C, p(x), p(mean), p(invstd), p(gamma), p(dx), None, p(dg), p(db),
                 1, rows, C, 1, p(ws3), st)
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            run()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 50 * 1e3
    gb = rows * C * 4 * 7 / 1e9      # partial: dy, y, x; apply: dy, y, x, dx
    print('rows %7d C %4d  %7.1f us  (%.0f GB/s over 7 tensor passes)' % (rows, C, res['three'], gb / res['three'] * 1e6),
          flush=True)
