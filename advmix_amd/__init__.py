"""advmix_amd - MI355X-native AdvMix training step (hand-written HIP kernels for gfx950
behind the reference's Python entry points).  See DESIGN.md / INTEGRATION.md.

Importing the package does not load the HIP library; ``advmix_amd.ops`` (and everything
that computes) does, and raises if ``libadvmix_hip.so`` has not been built.
"""
import os as _os

# ADVMIX_TAPE=1 (off by default): the launch lanes of a step replay as graphs of their own and hand work to each other with
# kernels that WAIT on a counter (ops.Tape).  Every lane then needs its own hardware queue; the HIP runtime reads
# GPU_MAX_HW_QUEUES (default 4) when it is loaded, so this only helps a process that imports advmix_amd BEFORE torch -
# otherwise export GPU_MAX_HW_QUEUES=8 in the environment.
if _os.environ.get('ADVMIX_TAPE', '0') == '1':
    _os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.2.0'
