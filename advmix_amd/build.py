"""Build libadvmix_hip.so (gfx950 only) with hipcc.  `python -m advmix_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SO = os.path.join(HERE, 'libadvmix_hip.so')
SOURCES = ['conv_mfma.hip', 'conv_direct.hip', 'wgrad_direct.hip', 'wgrad_lds.hip', 'norm.hip', 'pointwise.hip', 'advmix_ops.hip', 'postproc.hip', 'inputpipe.hip', 'nms.hip']
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-munsafe-fp-atomics', '-std=c++17',
         '-Wno-unused-result']


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = [os.path.join(CSRC, 'common.h'), os.path.join(HERE, '..', 'include', 'advmix_hip.h')]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on ' + s)
    if force or procs or _stale(SO, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', SO] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == '__main__':
    build(force='--force' in sys.argv)
