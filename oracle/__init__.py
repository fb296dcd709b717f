"""CPU oracle for the AdvMix hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``advmix_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

The oracle is a from-scratch, functional (state-dict driven) restatement of the
reference's per-batch training step in plain PyTorch CPU fp32 plus a small C
file for box NMS.  Every function cites the reference file:line it follows.

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, imported in the build
container by ``oracle/gen_golden.py`` (fixtures in ``tests/golden/``).  The
convolution / normalisation arithmetic itself lives in PyTorch (reference pins
``torch>=1.0.0``; ground truth here is torch 2.10.0 CPU fp32).
"""
