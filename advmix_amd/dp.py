"""Data parallelism for the AdvMix step: one process per GPU, full replicas, and exactly two
gradient all-reduces per step (D grads before optimizer.step(), G grads before
optimizer_G.step()) on the optimizers' flat fp32 gradient buffers - RCCL over xGMI on the
GPU (backend "nccl"), gloo in the CPU tests.

Replaces the reference's single-process nn.DataParallel (tools/train.py:69,106,109), which
re-broadcasts ~110 MiB of weights and scatters/gathers activations through GPU 0 four times
per step.  BatchNorm statistics stay per replica, as they are per shard under DataParallel.
The loss is a local-batch mean, so averaging gradients over equal shards reproduces the
reference's global-batch mean (lib/core/function.py:151-153)."""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, bucket_mb=64, group=None, force=False):
        self.group = group
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.force = force          # run the collectives even with one rank (single-GPU test of the path)
        self._side = None

    def _side_stream(self, device):
        if self._side is None and device.type == 'cuda':
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def all_reduce_mean(self, flat):
        """Average ``flat`` (1-D fp32) over ranks in place.  On CUDA the bucketed collectives
        run on a side HIP stream (so bucket k+1 overlaps the scaling of bucket k) and the
        current stream waits for them before the optimizer reads the buffer."""
        if self.world == 1 and not self.force:
            return
        n = flat.numel()
        if flat.is_cuda:
            cur = torch.cuda.current_stream(flat.device)
            side = self._side_stream(flat.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for lo in range(0, n, self.bucket_elems):
                    chunk = flat[lo:min(n, lo + self.bucket_elems)]
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.group)
            cur.wait_stream(side)
        else:
            for lo in range(0, n, self.bucket_elems):
                chunk = flat[lo:min(n, lo + self.bucket_elems)]
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
                chunk.div_(self.world)

    def sync(self, optimizer):
        self.all_reduce_mean(optimizer.flat_grads)
