"""One command, N ranks: start one fresh process per GPU and relay rank 0's output.

Replaces the reference's single-command multi-GPU entry - `GPUS: (0,...,7)` in the experiment YAMLs
(experiments/*:7) handed to `torch.nn.DataParallel(model, device_ids=cfg.GPUS)` in tools/train.py:69,106,109 -
for the one-process-per-GPU design (dp.py): the parent never touches the GPU (no HIP call, no `torch.cuda.*` call at all:
GPUs are counted from the kernel driver's sysfs topology), starts N children with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits for them and exits with the first
failing child's code.  Nothing is re-exec'ed: a process that has initialised the GPU must not be replaced.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


KFD_NODES = '/sys/class/kfd/kfd/topology/nodes'


def _kfd_gpu_nodes(root=KFD_NODES, dev_dir='/dev/dri'):
    """GPU nodes of the amdkfd topology, in node order: those with ``simd_count > 0`` (CPU nodes have none) whose render
    node this process may open - the ROCr runtime skips a GPU it cannot open (a container that was handed a subset)."""
    try:
        nodes = sorted((n for n in os.listdir(root) if n.isdigit()), key=int)
    except OSError:
        return None
    gpus = []
    for n in nodes:
        props = {}
        try:
            with open(os.path.join(root, n, 'properties')) as f:
                for ln in f:
                    k, _, v = ln.partition(' ')
                    props[k] = v.strip()
        except OSError:
            continue
        if int(props.get('simd_count', '0') or 0) <= 0:
            continue
        minor = props.get('drm_render_minor')
        if minor not in (None, '', '0', '-1') and dev_dir:
            dev = os.path.join(dev_dir, 'renderD%s' % minor)
            if os.path.exists(dev_dir) and not os.access(dev, os.R_OK | os.W_OK):
                continue
        gpus.append(n)
    return gpus


def _visible_filter(n_phys, env=None):
    """How many of ``n_phys`` devices survive ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (HIP
    reads the latter as an alias).  Entries are indices (or GPU-xxxx UUIDs, which are taken at face value); like the
    runtimes, the list ends at the first entry that does not name a device."""
    env = os.environ if env is None else env
    n = n_phys
    for keys in (('ROCR_VISIBLE_DEVICES',), ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')):
        val = next((env[k] for k in keys if k in env), None)
        if val is None:
            continue
        seen, cnt = set(), 0
        for tok in (t.strip() for t in val.split(',')):
            if tok.upper().startswith('GPU-'):
                pass
            elif not tok.lstrip('-').isdigit() or not (0 <= int(tok) < n) or tok in seen:
                break
            seen.add(tok)
            cnt += 1
        n = min(n, cnt)
    return n


def visible_gpus():
    """Number of GPUs this process could hand to children - WITHOUT initialising the HIP runtime here: on this torch
    build ``torch.cuda.device_count()`` is ``hipGetDeviceCount``, which brings HIP / HSA up in the calling process, and
    the children are forked from it.  The amdkfd sysfs topology says the same thing without a runtime; when the driver
    exposes no topology (no sysfs in the sandbox) a throw-away CHILD process asks torch - the parent still never does."""
    nodes = _kfd_gpu_nodes()
    if nodes is not None:
        return _visible_filter(len(nodes))
    try:
        out = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                             capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        return 0


def ranks_share_a_device(n, visible, share_gpu=False):
    """True when the ranks of this job cannot each have a GPU of their own: the caller says so (``share_gpu``: a functional
    run of the N-rank path on a smaller box) or fewer than ``n`` devices are visible while ranks are still started
    (``need_gpus=False``).  LOCAL_RANK r binds device r (bench.py, tools/train.py recipe), so with ``visible >= n`` every
    rank has its own."""
    return bool(share_gpu) or (visible is not None and 0 < visible < n)


def rank_env(r, n, port, shared_device, base=None, cores=None):
    """Environment of rank ``r`` of ``n``.
      * rendezvous variables, dmabuf IPC (the only kind this driver has; a user's value wins);
      * a bounded CPU budget (round 5): OMP / MKL threads = cores / n, at least 1 - N ranks each capturing ~3,500 launches
        into HIP graphs and each owning a torch intra-op pool of ALL cores oversubscribe the host N times over
        (``cores``: os.cpu_count() unless given; a user's OMP_NUM_THREADS wins);
      * DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 ONLY when ranks share a device (``shared_device``): with two processes on ONE GPU,
        replays served from the runtime's captured AQL packets computed garbage after NULL-stream traffic between them
        (tools/dp_graph_repro.py; 0 failures with the switch).  Every ingredient of that failure needs
        the second process on the SAME device, so a one-process-per-GPU job does not carry an undocumented runtime debug
        switch; ADVMIX_GRAPH_PACKET_CAPTURE_OFF=1 forces it on for A/B runs."""
    env = dict(os.environ if base is None else base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    threads = str(max(1, (cores or os.cpu_count() or 1) // n))
    env.setdefault('OMP_NUM_THREADS', threads)
    env.setdefault('MKL_NUM_THREADS', threads)
    if shared_device or env.get('ADVMIX_GRAPH_PACKET_CAPTURE_OFF') == '1':
        env.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
    return env


def spawn_ranks(argv, n, need_gpus=True, timeout_s=None, env_extra=None, share_gpu=False):
    """Run ``argv`` (a full command line) as ranks 0..n-1 of one node.  Rank 0 inherits stdout (its JSON line is the
    job's output); the other ranks' stdout goes to stderr.  Returns the job's exit code: 0 only if EVERY rank exited 0.
    Raises SystemExit with a non-zero code if fewer than ``n`` GPUs are visible (``need_gpus``) - never runs fewer
    ranks than asked for.  Children are FRESH processes (nothing that has touched the GPU is ever re-exec'ed)."""
    if n < 2:
        raise ValueError('spawn_ranks is for n >= 2 ranks')
    have = None
    if need_gpus or share_gpu:
        have = visible_gpus()
    if need_gpus and have < n:
        sys.stderr.write('launch: --gpus %d asked for but only %d GPU(s) visible - refusing to run fewer ranks\n'
                         % (n, have))
        raise SystemExit(2)
    shared = ranks_share_a_device(n, have, share_gpu)
    port = os.environ.get('MASTER_PORT') or str(free_port())
    procs = []
    try:
        for r in range(n):
            env = rank_env(r, n, port, shared)
            env.update(env_extra or {})
            # ranks > 0: stdout onto the parent's stderr by DESCRIPTOR (sys.stderr may be a capture object with no fileno)
            procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else 2))
    except OSError as e:
        for p in procs:
            p.kill()
        sys.stderr.write('launch: could not start rank %d: %s\n' % (len(procs), e))
        raise SystemExit(3)
    t0 = time.time()
    code = 0
    live = list(procs)

    def _stop(signum, frame):                               # the parent is being stopped (driver timeout, ^C): take the
        for q in procs:                                     # ranks along - an orphaned rank would keep its GPU
            if q.poll() is None:
                q.terminate()
        raise SystemExit(128 + signum)
    old = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            old[sg] = signal.signal(sg, _stop)
        except ValueError:                                  # not the main thread: leave the handlers alone
            pass
    try:
        return _wait(procs, live, code, t0, timeout_s)
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)


def _wait(procs, live, code, t0, timeout_s):
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                sys.stderr.write('launch: rank %d exited with %d - stopping the others\n' % (procs.index(p), rc))
                for q in live:
                    q.terminate()
        if timeout_s is not None and time.time() - t0 > timeout_s and live:
            sys.stderr.write('launch: timeout after %.0f s\n' % timeout_s)
            for q in live:
                q.kill()
            code = code or 124
        time.sleep(0.05)
    return code
