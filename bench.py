#!/usr/bin/env python3
"""Headline benchmark: images/sec of one full AdvMix train step (HRNet-W32 256x192, B=32 per
GPU, synthetic inputs resident in HBM, random-init weights), fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hrnet_w32|resnet50|hrnet_w48]

N > 1: either the driver's `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, or plain
`python bench.py --gpus N`, which starts its own N ranks (advmix_amd/launch.py: one fresh process per GPU before
anything touches the GPU; exits non-zero if fewer than N GPUs are visible - it never silently runs one rank).
One rank per GPU, RCCL; per-GPU work is fixed (weak scaling); `rccl_ranks` in the line is dist.get_world_size().  A "step" is the
body of train_advmix's batch loop (lib/core/function.py:137-171): G fwd, softmax-mix, D step
(heat-map + KD loss, backward, Adam), G step through the updated frozen D (backward, Adam),
loss.item() and the PCK accuracy read-out - nothing is skipped inside the timed region.

The JSON line also carries
  roofline     - the step's MFMA conv family timed live with HIP events (tools/bench_roofline.py): algorithmic FLOPs per
                 launch / launch time vs the 157.3 TFLOP/s fp32 matrix peak, the member the step spends most time in as
                 ``dominant``, and the step's kernel-time shares by class from the committed per-shape profile
  cpu_baseline - the CPU oracle (a restatement of the reference step, pinned to it by golden
                 vectors) timed on this box's host cores on a bounded sample (B=4, a few steps).
The other measurements of this CLI (--path validate | inputs | nms | rendezvous | replicas, --through-loop, --dump-shapes)
live in tools/bench_paths.py.
"""
import argparse
import json
import os
import sys
import time
import types

# Two processes SHARING one GPU only (ADVMIX_BENCH_SHARE_GPU=1, a functional run of the N-rank path on a one-GPU box): the HIP
# runtime's captured-packet graph launches computed garbage there; read by the runtime when it comes
# up, i.e. before ``import torch``.  A real one-process-per-GPU job does not set it (advmix_amd/launch.py decides the same way).
if int(os.environ.get('WORLD_SIZE', '1')) > 1 and os.environ.get('ADVMIX_BENCH_SHARE_GPU') == '1':
    os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'tools')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from bench_common import (FP32_MFMA_PEAK_TFLOPS, HRNET_STAGES, NO_ORACLE, WORKLOADS, build_models, cpu_baseline,   # noqa: E402,F401
                          hrnet_extra, synth, _event_time, _ranks)

SYNC_METRICS = os.environ.get('ADVMIX_SYNC_METRICS') == '1'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)          # discard >= 10 warm-up steps, average >= 50
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='hrnet_w32', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=32, help='images per GPU (TRAIN.BATCH_SIZE_PER_GPU)')
    ap.add_argument('--exec', dest='exec_mode', default='graph', choices=['graph', 'eager'])
    ap.add_argument('--path', default='train', choices=['train', 'validate', 'inputs', 'nms', 'rendezvous', 'replicas'],
                    help='train = the headline AdvMix step; validate = the validate() batch body (SURVEY 8 f1); '
                         'inputs = the device input pipeline (SURVEY 8 f2); rendezvous = start the ranks, one all-reduce, '
                         'report (launcher self-test; ADVMIX_BENCH_BACKEND=gloo runs it without GPUs)')
    ap.add_argument('--through-loop', action='store_true',
                    help='time core.function.train_advmix itself (the drop-in entry point): pinned host batches, H2D copies, '
                         'graph replay, loss.item(), accuracy - the reference loop body lib/core/function.py:107-197')
    ap.add_argument('--dump-shapes', default=None, metavar='CSV',
                    help='log kernel template, grid, shape and FLOPs of every MFMA launch of ONE eager step (for '
                         'tools/kernel_shapes.py) and exit')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-verify', action='store_true', help='skip the data-parallel self-verification steps (N > 1 ranks)')
    ap.add_argument('--no-through-loop', action='store_true', help='skip the extra train_advmix-with-H2D measurement')
    ap.add_argument('--no-roofline', action='store_true')
    a = ap.parse_args()
    if a.dump_shapes:
        os.environ['ADVMIX_TRACE_SHAPES'] = os.path.abspath(a.dump_shapes)

    import torch.distributed as dist
    backend = os.environ.get('ADVMIX_BENCH_BACKEND', 'nccl')      # 'gloo' only for --path rendezvous (CPU launcher test)
    # ADVMIX_BENCH_SHARE_GPU=1: a FUNCTIONAL run of the N-rank path on a box with fewer GPUs - every rank uses cuda:0 and the
    # gradient exchange goes over gloo (RCCL refuses two ranks on one device).  The line says so; it is not a scaling number.
    share_gpu = os.environ.get('ADVMIX_BENCH_SHARE_GPU') == '1'
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # One command, N ranks (the reference's multi-GPU entry is one command too: GPUS in the YAML ->
        # nn.DataParallel, tools/train.py:69,106,109).  This parent has NOT touched the GPU; it starts one fresh
        # process per GPU, relays rank 0's JSON line and fails loudly rather than run fewer ranks than asked for.
        from advmix_amd.launch import spawn_ranks
        need = not (a.path in ('rendezvous', 'replicas') and backend == 'gloo') and not share_gpu
        raise SystemExit(spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], a.gpus, need_gpus=need,
                                     share_gpu=share_gpu))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit('WORLD_SIZE %d != --gpus %d' % (world, a.gpus))
    import bench_paths as BP
    if a.path == 'rendezvous':
        return BP.rendezvous(a, backend, rank, world, local)
    if a.path == 'replicas':
        return BP.replicas_selftest(a, backend, rank, world, local)
    if backend != 'nccl':
        raise SystemExit('ADVMIX_BENCH_BACKEND=%s is only for --path rendezvous' % backend)
    if share_gpu:
        local, backend = 0, 'gloo'
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    force_sync = os.environ.get('ADVMIX_FORCE_SYNC') == '1'     # exercise the RCCL path with a single rank
    if world > 1 or force_sync:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29555')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if a.path in ('validate', 'inputs', 'nms'):
        line = {'validate': BP.bench_validate, 'inputs': BP.bench_inputs, 'nms': BP.bench_nms}[a.path](a, device, rank, world)
        if world > 1 or force_sync:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            sys.stdout.flush()
            print(json.dumps(line), flush=True)
        return

    from advmix_amd.core.function import advmix_step
    from advmix_amd.core import function as F_
    from advmix_amd.core.evaluate import PendingAccuracy
    from advmix_amd.dp import GradSync
    from advmix_amd.graph import AdvMixGraphRunner

    # Everything from here on runs on a CREATED stream (as core.function's loops do): graph replays, exchanges, barriers and
    # the verification collectives inherit it - none is ever issued with the NULL stream current.
    main_stream = torch.cuda.Stream(device=device)
    main_stream.wait_stream(torch.cuda.current_stream(device))
    stream_ctx = torch.cuda.stream(main_stream)
    stream_ctx.__enter__()
    net, extra, J, H, W, downs, gflop_img = WORKLOADS[a.workload]
    cfg, D, G, T, crit, optD, optG = build_models(a.workload, device)
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    views, tgt, tw = synth(a.batch, J, H, W, device, 1234 + rank)
    sync = GradSync(force=force_sync) if (world > 1 or force_sync) else None
    if sync is not None:
        sync.broadcast_state([D, G, T], [optD, optG])          # every replica starts from rank 0's weights / Adam state
    if a.dump_shapes:
        return BP.dump_shapes(a.dump_shapes, lambda: advmix_step(args, D, G, T, crit, optD, optG, views, tgt, tw, sync))
    loop_note = None
    if world > 1 and a.exec_mode == 'graph' and not F_.DP_GRAPH:
        a.exec_mode = 'eager'                               # core.function.DP_GRAPH (ADVMIX_DP_GRAPH=0): the line says which

    def make_step(holder=None):
        """The execution the timed region uses: the HIP-graph runner (seven graphs with data parallelism) or the eager step."""
        if a.exec_mode == 'graph':
            t_cap = time.perf_counter()
            runner = AdvMixGraphRunner(args, D, G, T, crit, optD, optG, views, tgt, tw, sync)
            if holder is not None:
                holder['runner'], holder['capture_s'] = runner, time.perf_counter() - t_cap
            return lambda: runner.step() + (runner.target,)
        return lambda: advmix_step(args, D, G, T, crit, optD, optG, views, tgt, tw, sync) + (tgt,)

    def loop(n_warm, n_timed):
        return BP.through_loop(a, cfg, args, (D, G, T), crit, (optD, optG), sync, rank, world, n_warm, n_timed)

    def max_over_ranks(x):
        if world > 1:
            tmax = torch.tensor([x], device=device, dtype=torch.float64)
            with sync.off_null(tmax):                       # (no collective with the NULL stream current, dp.GradSync.off_null)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            return float(tmax.item())
        return x

    verification = None
    exchange = None
    dt_loop = None
    hold = {}
    if a.through_loop:
        dt, lv, acc = loop(a.warmup, a.steps)
        loop_note = 'core.function.train_advmix over pinned host batches (H2D inside the timed region)'
    else:
        step = make_step(hold)
        if sync is not None and not a.no_verify:
            # before anything is timed: the N-rank execution proves itself - three steps through THIS runner
            verification = BP.verify_data_parallel(step, args, (D, G, T), crit, (optD, optG), (views, tgt, tw), sync)
            if a.exec_mode == 'graph' and (not verification[0] or os.environ.get('ADVMIX_BENCH_FAIL_GRAPH_VERIFY') == '1'):
                # the replayed graphs did not prove themselves on this machine: the same step WITHOUT graphs, from rank 0's
                # state again, verified the same way - the line says so (config.exec 'eager', dp_verification.graph_attempt)
                graph_attempt = dict(verification[1], verified=verification[0])
                hold.clear()
                del step
                torch.cuda.synchronize()
                sync.broadcast_state([D, G, T], [optD, optG])
                a.exec_mode = 'eager'
                step = make_step()
                verification = BP.verify_data_parallel(step, args, (D, G, T), crit, (optD, optG), (views, tgt, tw), sync)
                verification[1]['graph_attempt'] = graph_attempt

        def launch():
            loss_D, out, target = step()
            return PendingAccuracy(out, target, loss_D)     # function.py:167-168, device half enqueued

        def run_steps(n):
            """n steps; loss.item() and accuracy() of every step are read - one step late, while the next one runs (the
            loop mirror core.function.train_advmix does the same), the last one before returning."""
            pend, lv, acc = None, float('nan'), 0.0
            for _ in range(n):
                cur = launch()
                if SYNC_METRICS:                            # A/B switch: read every step's numbers before the next launch
                    cur.get()
                if pend is not None:
                    _, acc, _, _, lv = pend.get()
                pend = cur
            if pend is not None:
                _, acc, _, _, lv = pend.get()
            return lv, acc

        run_steps(a.warmup)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if sync is not None:
            sync.meter = True                               # (two events per finish(): what the exchange makes the step wait)
        t0 = time.perf_counter()
        lv, acc = run_steps(a.steps)
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0                   # this rank's own clock, before it waits for the slowest one
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if sync is not None:
            sync.meter = False
            exchange = BP.rank_diagnosis(sync, dt_own, a.steps, device)
        if not a.no_through_loop and sync is None:
            # (one rank: a second capture in one process is fine there; with the data-parallel runner it is not attempted)
            # SURVEY 8 d1's step includes the H2D of step 1: the same workload through train_advmix itself, beside the
            # resident-input figure (never instead of it)
            capture_s = hold.get('capture_s')
            hold.clear()
            hold['capture_s'] = capture_s
            del step
            torch.cuda.empty_cache()
            n_loop = 50 if a.steps >= 20 else a.steps
            dt_loop = (max_over_ranks(loop(10, n_loop)[0]), n_loop)
    dt = max_over_ranks(dt)
    replicas = sync.replicas_state([optD, optG]) if sync is not None else None
    from advmix_amd._lib import lib as _hiplib
    variant = _hiplib.advmix_build_flags()                  # non-zero: a tools/build_variant.sh library (ADVMIX_SO=...)
    if not (lv == lv) and not variant:
        raise SystemExit('loss is NaN')

    line = None
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = a.batch * world * a.steps / dt
        line = {
            'metric': 'images/sec AdvMix train step (HRNet-W32 256x192)' if a.workload == 'hrnet_w32'
            else 'images/sec AdvMix train step (%s)' % a.workload,
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'rccl_ranks': _ranks(), 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic (N(0,1) views, Gaussian targets), random-init weights',
            'config': {'workload': '%s_%dx%d_advmix' % (a.workload, H, W), 'batch_per_gpu': a.batch,
                       'global_batch': a.batch * world, 'generator': 'UnetGenerator(9,3,%d)' % downs,
                       'parallelism': 'dp%d' % world, 'exec': 'hipgraph' if a.exec_mode == 'graph' else 'eager',
                       'entry': loop_note or 'graph.AdvMixGraphRunner.step (inputs resident in HBM)',
                       'step_gflop_per_image': gflop_img},
        }
        if a.workload in NO_ORACLE:
            line['config']['parity'] = NO_ORACLE[a.workload]
        if share_gpu:
            line['config']['shared_gpu'] = ('all %d ranks on cuda:0, gradient exchange over gloo: a functional run of the '
                                            'data-parallel path, NOT a scaling measurement' % world)
        line.update({
            'step_tflops_per_gpu': round(value / world * gflop_img / 1e3, 2),
            'step_frac_of_fp32_mfma_peak': round(value / world * gflop_img / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
            'last_loss_D': round(lv, 6) if lv == lv else None,
        })
        if hold.get('capture_s') is not None:
            line['graph_capture_s'] = round(hold['capture_s'], 2)      # warm-up steps + capture of the step's graphs, this rank
        if dt_loop is not None:
            line['value_through_loop'] = round(a.batch * world * dt_loop[1] / dt_loop[0], 2)
            line['through_loop'] = {'entry': 'core.function.train_advmix over pinned host batches: H2D copies, graph replay, '
                                             'loss.item(), accuracy inside the timed region (SURVEY 8 d1)',
                                    'steps': dt_loop[1], 'warmup': 10, 'ms_per_step': round(dt_loop[0] / dt_loop[1] * 1e3, 3)}
        if sync is not None:
            line['replicas_identical'], line['all_finite'] = replicas['identical'], replicas['finite'] and (lv == lv)
            line['grad_exchange_verified'] = verification[0] if verification is not None else None
            line['dp_verification'] = dict(verification[1], exec=line['config']['exec'],
                                           transport=dist.get_backend()) if verification is not None else 'skipped (--no-verify)'
            if exchange is not None:
                line['ranks'] = exchange                    # per-rank step time, exchange wait, bytes: why an efficiency is what it is
        if variant:
            line['INVALID_variant_build_flags'] = variant   # measurement build: never a benchmark result
        if not a.no_roofline:
            from bench_roofline import time_conv_family
            widths = HRNET_STAGES.get('hrnet_w48' if a.workload == 'hrnet_w48' else 'hrnet_w32')
            line['roofline'] = time_conv_family(a.batch, device, family=tuple(
                (c, (H // 4) >> i, (W // 4) >> i) for i, c in enumerate(widths)), workload=a.workload)
        if world == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(a.workload)

    failed = sync is not None and (not replicas['identical'] or not replicas['finite']
                                   or (verification is not None and not verification[0]))
    if world > 1 or force_sync:
        dist.barrier()
    torch.cuda.synchronize()
    stream_ctx.__exit__(None, None, None)
    if world > 1 or force_sync:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so the JSON line is the LAST line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    if failed:
        raise SystemExit(5)                                 # the line says which of the three checks failed


if __name__ == '__main__':
    main()
