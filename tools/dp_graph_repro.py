"""Reproducer / bisection harness for the round-3 open bug: the seven-graph data-parallel runner corrupting exchanged
gradients with two ranks.

    python tools/dp_graph_repro.py --out gpurun_out/r04a/dp_repro.jsonl [--only NAME ...] [--runs 3] [--replays 6]

The parent NEVER touches the GPU; every run is a pair (or one) of fresh child processes.  Variants (table VARIANTS below):
two ranks on cuda:0 over gloo with the runner's segments replayed by THIS file's loop, so that each ordering ingredient
can be switched alone (fence launch, host sync, exchange on the replay stream, packet capture off, finite checks before
and after every exchange); and ONE process with a transport that changes the data (device-only, or gloo's host round
trip restated) against the eager step from the same state, bit for bit in deterministic mode.
"""
import argparse
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

# name -> (kind, env, options)
NULL = {'ADVMIX_REPLAY_STREAM': 'null'}                     # round 3's behaviour: replays on the caller's (NULL) stream
VARIANTS = {
    # --- round 4, first pass (gpurun_out/r04a): replays on the NULL stream, two ranks over gloo on one GPU ---
    'r2_l1_nofence':  ('two', dict(NULL, ADVMIX_LANES='1'), {'custom': 1}),
    'r2_l1_pkt0':     ('two', dict(NULL, ADVMIX_LANES='1', DEBUG_CLR_GRAPH_PACKET_CAPTURE='0'), {'custom': 1}),
    'r2_l1_inline':   ('two', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'inline': 1}),
    'r2_l1_hostsync': ('two', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'hostsync': 1}),
    'r2_l1_check':    ('two', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'check': 1}),
    'r2_l1_ownstream': ('two', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'ownstream': 1}),
    'r2_l4_nofence':  ('two', dict(NULL, ADVMIX_LANES='4'), {'custom': 1}),
    'r2_l4_pkt0':     ('two', dict(NULL, ADVMIX_LANES='4', DEBUG_CLR_GRAPH_PACKET_CAPTURE='0'), {'custom': 1}),
    'p1_l1_devmul':   ('one', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'transport': 'devmul'}),
    'p1_l1_hostrt':   ('one', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'transport': 'hostrt'}),
    'p1_l4_hostrt':   ('one', dict(NULL, ADVMIX_LANES='4'), {'custom': 1, 'transport': 'hostrt'}),
    # --- second pass: what in the two-rank rig is the trigger? (all on the NULL stream) ---
    'p1_l1_hostrt_hiprio': ('one', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'transport': 'hostrt', 'hiprio': 1}),   # gloo's streams are high-priority
    'p1_l1_gloo1':    ('one', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'transport': 'gloo1'}),        # gloo's own machinery, one rank
    'p2_l1_devmul':   ('one2', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'transport': 'devmul'}),      # two INDEPENDENT processes on the GPU
    'p1_l1_noise':    ('one+noise', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'transport': 'devmul'}), # one process + a stranger's kernels
    # --- the product as shipped (replays on the runner's own stream, graph.AdvMixGraphRunner.step itself) ---
    'r2_l1_product':  ('two', {'ADVMIX_LANES': '1'}, {}),
    'r2_l4_product':  ('two', {'ADVMIX_LANES': '4'}, {}),
    'p1_l4_product':  ('one', {'ADVMIX_LANES': '4'}, {'transport': 'hostrt'}),
    # --- third pass (r04c): the own-stream product step STILL failed (0/4, 1/4) although the custom loop run wholly under a
    #     non-null stream had passed (3/3).  What is left of the NULL stream there: the per-step hand-off events and the
    #     checks between the steps - gloo all_gathers issued with the NULL stream current.
    'r2_l1_product_chk':  ('two', {'ADVMIX_LANES': '1'}, {'chk_stream': 1}),                 # hand-offs stay, gloo never sees NULL
    'r2_l1_null_chk':     ('two', dict(NULL, ADVMIX_LANES='1'), {'chk_stream': 1}),          # NULL replays, gloo never sees NULL
    'r2_l1_loopstream':   ('two', {'ADVMIX_LANES': '1'}, {'ownstream': 1}),                  # everything under one non-null stream
    'r2_l4_loopstream':   ('two', {'ADVMIX_LANES': '4'}, {'ownstream': 1}),
    'p1_l1_gloo1_nullchk': ('one', dict(NULL, ADVMIX_LANES='1'), {'custom': 1, 'transport': 'gloo1', 'null_gather': 1}),
    # --- fourth pass (r04e): the two-rank TEST still failed with every collective off the NULL stream (5 of 5).  What it
    #     does on the NULL stream between two steps that the passing variant above does not: loss.item(), and the kernels
    #     of GradSync.replicas_state's fold (its all_gather is off the NULL stream).
    'r2_l1_product_item': ('two', {'ADVMIX_LANES': '1'}, {'chk_stream': 1, 'item': 1}),
    'r2_l1_product_fold': ('two', {'ADVMIX_LANES': '1'}, {'chk_stream': 1, 'fold': 1}),
}


# ------------------------------------------------------------------------------------------------------- workers
def _setup(salt):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_models_gpu as T
    return T._tiny_setup(salt=salt, lr=1e-3)


def _param_map(opt, model):
    """[(flat offset, numel, name)] sorted by offset: which parameter an index of the flat buffers belongs to."""
    names = {id(p): n for n, p in model.named_parameters()}
    rows = [(p._flat_off, p.numel(), names.get(id(p), '?')) for p in opt._params()]
    return sorted(rows)


def _bad_report(flat, pmap, lo=0, hi=None):
    """Which parameters hold non-finite or absurd (> 1e20) values in flat[lo:hi]."""
    import torch
    hi = flat.numel() if hi is None else hi
    v = flat[lo:hi]
    bad = (~torch.isfinite(v)) | (v.abs() > 1e20)
    n = int(bad.sum())
    if n == 0:
        return None
    idx = (bad.nonzero().flatten() + lo).tolist()
    hits = {}
    for i in idx[:20000]:
        for off, numel, name in pmap:
            if off <= i < off + numel:
                hits[name] = hits.get(name, 0) + 1
                break
    return {'n_bad': n, 'first': idx[:8], 'max_abs': float(v[torch.isfinite(v)].abs().max()) if bool(torch.isfinite(v).any()) else None,
            'params': sorted(hits.items(), key=lambda kv: -kv[1])[:12]}


def _replay_loop(runner, sync, opt, log, pmaps, step_no):
    """graph.AdvMixGraphRunner.step with every ordering ingredient switchable (options in OPT)."""
    import torch
    if not OPT.get('custom'):
        runner.step()                                       # the product's own step
        return
    runner.opt.sync_hyper()
    runner.optG.sync_hyper()
    for si, seg in enumerate(runner.segments):
        g, red = seg[0], seg[1]
        if len(seg) > 2 and seg[2]:
            sync.finish()
        runner.seq.replay(g)
        if red is None:
            continue
        o, lo, hi = red
        if OPT.get('hostsync'):
            torch.cuda.current_stream().synchronize()
        if OPT.get('check'):
            torch.cuda.synchronize()
            r = _bad_report(o.flat_grads, pmaps[id(o)], lo, hi)
            if r is not None:
                log.append({'step': step_no, 'seg': si, 'when': 'pre-exchange', 'range': [lo, hi], **r})
        if OPT.get('inline'):                               # the exchange on the replay stream itself: no event crossing
            chunk = o.flat_grads[lo:hi]
            for b in range(0, hi - lo, sync.bucket_elems):
                sync._mean_(chunk[b:b + sync.bucket_elems])
        else:
            sync.reduce_async(o.flat_grads, lo, hi)
        if OPT.get('check'):
            torch.cuda.synchronize()
            r = _bad_report(o.flat_grads, pmaps[id(o)], lo, hi)
            if r is not None:
                log.append({'step': step_no, 'seg': si, 'when': 'post-exchange', 'range': [lo, hi], **r})


def worker_two():
    import types
    import torch
    import torch.distributed as dist
    from oracle.synth import synth_batch
    from advmix_amd.dp import GradSync
    from advmix_amd.graph import AdvMixGraphRunner
    rank = int(os.environ['RANK'])
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % os.environ['MASTER_PORT'], rank=rank, world_size=2)
    torch.cuda.set_device(0)

    def same(t):
        got = [torch.zeros_like(t), torch.zeros_like(t)]
        dist.all_gather(got, t.contiguous())
        return bool(torch.equal(got[0], got[1]))
    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    B, J, H, W = 2, 5, 64, 64
    v, t, w = synth_batch('hrnet_tiny.it%d' % rank, B, J, H, W)
    data = ([x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
    cfg, D, G, T_, crit, oD, oG, _ = _setup(10 + 7 * rank)
    sync = GradSync(bucket_mb=0.25)
    sync.broadcast_state([D, G, T_], [oD, oG])
    pmaps = {id(oD): _param_map(oD, D), id(oG): _param_map(oG, G)}
    log, res = [], {'rank': rank, 'ok': True, 'first_bad_step': None}
    runner = AdvMixGraphRunner(args, D, G, T_, crit, oD, oG, *data, sync)
    ctx = torch.cuda.stream(torch.cuda.Stream()) if OPT.get('ownstream') else None
    if ctx is not None:
        ctx.__enter__()
    chk = torch.cuda.Stream() if OPT.get('chk_stream') else None
    for k in range(REPLAYS):
        _replay_loop(runner, sync, None, log, pmaps, k)
        if OPT.get('item'):
            float(runner.loss_D)                            # a synchronous read-back with the NULL stream current
        if OPT.get('fold'):
            sync.replicas_state([oD, oG])                   # fold kernels on the NULL stream, the all_gather off it
        torch.cuda.synchronize()
        cctx = torch.cuda.stream(chk) if chk is not None else None    # the checks (gloo all_gathers) off the NULL stream
        if cctx is not None:
            cctx.__enter__()
        fin = all(bool(torch.isfinite(x.float()).all()) for x in oD.flat_state() + oG.flat_state())
        eq = all(same(x) for x in oD.flat_state()) and all(same(x) for x in oG.flat_state())
        torch.cuda.synchronize()
        if cctx is not None:
            cctx.__exit__(None, None, None)
        if not (fin and eq):
            res['ok'] = False
            res['first_bad_step'] = k
            res['finite'], res['replicas_equal'] = fin, eq
            res['bad_D_grads'] = _bad_report(oD.flat_grads, pmaps[id(oD)])
            res['bad_G_grads'] = _bad_report(oG.flat_grads, pmaps[id(oG)])
            break
    if ctx is not None:
        ctx.__exit__(None, None, None)
    res['log'] = log[:40]
    print('DPREPRO ' + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def worker_one():
    """ONE process: seven-graph runner vs the eager step from the same state, deterministic mode, the exchange replaced by
    a transport that CHANGES the data (x 0.5) on the side stream."""
    import types
    import torch
    from oracle.synth import synth_batch
    from advmix_amd import ops
    from advmix_amd.core.function import advmix_step
    from advmix_amd.dp import GradSync
    from advmix_amd.graph import AdvMixGraphRunner
    torch.cuda.set_device(0)
    transport = OPT.get('transport', 'devmul')
    if transport == 'gloo1':
        import torch.distributed as dist
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % os.environ['MASTER_PORT'], rank=0, world_size=1)
    pool = [torch.cuda.Stream(priority=-1 if OPT.get('hiprio') else 0) for _ in range(3)]
    count = [0]

    class FakeSync(GradSync):
        def __init__(self):
            super().__init__(bucket_mb=0.25, force=True)

        def _mean_(self, t):
            if transport == 'devmul':
                t.mul_(0.5)
                return
            if transport == 'gloo1':                        # the real thing with one rank: pinned round trip on gloo's streams
                import torch.distributed as dist
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                t.mul_(0.5)
                return
            # gloo's AsyncAllreduceCUDAWork restated: an internal stream waits for an event recorded on the caller's
            # stream, copies to a pinned buffer, the HOST waits for that stream and reduces, copies back asynchronously,
            # the caller's stream waits for the copy's event
            cur = torch.cuda.current_stream()
            s = pool[count[0] % len(pool)]
            count[0] += 1
            ev = torch.cuda.Event()
            ev.record(cur)
            s.wait_event(ev)
            with torch.cuda.stream(s):
                tmp = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                tmp.copy_(t, non_blocking=True)
            s.synchronize()
            tmp.mul_(0.5)
            with torch.cuda.stream(s):
                t.copy_(tmp, non_blocking=True)
                ev2 = torch.cuda.Event()
                ev2.record(s)
            cur.wait_event(ev2)

    args = types.SimpleNamespace(alpha=0.1, adv_loss_weight=1.0)
    B, J, H, W = 2, 5, 64, 64
    v, t, w = synth_batch('hrnet_tiny.it0', B, J, H, W)
    data = ([x.cuda().contiguous() for x in v], t.cuda(), w.cuda())
    ops.set_deterministic(True)
    cfg, D, G, T_, crit, oD, oG, _ = _setup(10)
    cfg, D2, G2, T2, crit2, oD2, oG2, _ = _setup(10)
    sync, sync2 = FakeSync(), FakeSync()
    pmaps = {id(oD): _param_map(oD, D), id(oG): _param_map(oG, G)}
    runner = AdvMixGraphRunner(args, D, G, T_, crit, oD, oG, *data, sync)
    log, res = [], {'rank': 0, 'ok': True, 'first_bad_step': None, 'diffs': []}
    for k in range(REPLAYS):
        _replay_loop(runner, sync, None, log, pmaps, k)
        advmix_step(args, D2, G2, T2, crit2, oD2, oG2, *data, sync2)
        torch.cuda.synchronize()
        if OPT.get('null_gather'):                          # what the two-rank worker's checks do: gloo all_gathers, NULL stream current
            import torch.distributed as dist
            for x in oD.flat_state() + oG.flat_state():
                got = [torch.zeros_like(x)]
                dist.all_gather(got, x.contiguous())
            torch.cuda.synchronize()
        fin = all(bool(torch.isfinite(x.float()).all()) for x in oD.flat_state() + oG.flat_state())
        dD = float((oD.flat_params - oD2.flat_params).abs().max())
        dG = float((oG.flat_params - oG2.flat_params).abs().max())
        gD = float((oD.flat_grads - oD2.flat_grads).abs().max())
        gG = float((oG.flat_grads - oG2.flat_grads).abs().max())
        res['diffs'].append([dD, dG, gD, gG])
        if not fin or not (dD < 1e-2 and dG < 1e-2):
            res['ok'] = False
            res['first_bad_step'] = k
            res['finite'] = fin
            res['bad_D_grads'] = _bad_report(oD.flat_grads, pmaps[id(oD)])
            res['bad_G_grads'] = _bad_report(oG.flat_grads, pmaps[id(oG)])
            break
    res['log'] = log[:40]
    print('DPREPRO ' + json.dumps(res), flush=True)


# --------------------------------------------------------------------------------------------------------- parent
def run_variant(name, run_no, replays, port, timeout=420):
    kind, env_extra, opt = VARIANTS[name]
    n = 2 if kind in ('two', 'one2') else 1
    procs = []
    t0 = time.time()
    noise = None
    if kind == 'one+noise':                                 # a stranger on the same GPU: back-to-back matmuls until told to stop
        noise = subprocess.Popen([sys.executable, '-c',
                                  'import torch, time, os\nx = torch.randn(4096, 4096, device="cuda")\nt0 = time.time()\n'
                                  'while time.time() - t0 < 150 and not os.path.exists(%r):\n    y = x @ x\n    torch.cuda.synchronize()\n'
                                  % ('/tmp/dprepro_stop_%d' % port)])
        time.sleep(8)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port + (r if kind == 'one2' else 0)),
                   DPREPRO_OPT=json.dumps(opt), DPREPRO_REPLAYS=str(replays), **env_extra)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker', 'two' if kind == 'two' else 'one'],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = {'variant': name, 'run': run_no, 'ranks': [], 'rc': []}
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            se += '\nTIMEOUT'
        out['rc'].append(p.returncode)
        got = [json.loads(l[8:]) for l in so.splitlines() if l.startswith('DPREPRO ')]
        out['ranks'].append(got[0] if got else {'ok': False, 'stderr_tail': se[-1500:]})
    if noise is not None:
        open('/tmp/dprepro_stop_%d' % port, 'w').close()
        noise.wait(timeout=60)
    out['ok'] = all(r.get('ok') for r in out['ranks']) and all(c == 0 for c in out['rc'])
    out['seconds'] = round(time.time() - t0, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--worker', default=None)
    ap.add_argument('--out', default='gpurun_out/dp_repro.jsonl')
    ap.add_argument('--only', nargs='*', default=None)
    ap.add_argument('--runs', type=int, default=3)
    ap.add_argument('--replays', type=int, default=6)
    a = ap.parse_args()
    if a.worker:
        global OPT, REPLAYS
        OPT = json.loads(os.environ.get('DPREPRO_OPT', '{}'))
        REPLAYS = int(os.environ.get('DPREPRO_REPLAYS', '6'))
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        (worker_two if a.worker == 'two' else worker_one)()
        return
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    names = a.only or list(VARIANTS)
    port = 29700
    summary = {}
    with open(a.out, 'a') as f:
        for name in names:
            for r in range(a.runs):
                port += 3
                res = run_variant(name, r, a.replays, port)
                f.write(json.dumps(res) + '\n')
                f.flush()
                s = summary.setdefault(name, [0, 0])
                s[0] += 1 if res['ok'] else 0
                s[1] += 1
                print('%-18s run %d: %s (%.0f s)%s' % (name, r, 'ok' if res['ok'] else 'FAIL', res['seconds'],
                      '' if res['ok'] else '  ' + json.dumps(res['ranks'][0])[:600]), flush=True)
    print('SUMMARY ' + json.dumps({k: '%d/%d ok' % tuple(v) for k, v in summary.items()}))


OPT, REPLAYS = {}, 6
if __name__ == '__main__':
    main()
