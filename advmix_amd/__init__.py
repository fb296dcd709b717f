"""advmix_amd - MI355X-native AdvMix training step (hand-written HIP kernels for gfx950
behind the reference's Python entry points).  See DESIGN.md / INTEGRATION.md.

Importing the package does not load the HIP library; ``advmix_amd.ops`` (and everything
that computes) does, and raises if ``libadvmix_hip.so`` has not been built.
"""
__version__ = '0.2.0'
