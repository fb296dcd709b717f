import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _effective_cpus():
    """CPUs this process may really use (affinity mask AND cgroup quota): the GPU boxes show 256 CPUs to a 16-CPU
    cgroup, and an OpenMP team sized for 256 spends its life being throttled - the CPU oracle then runs 3-6x slower."""
    from oracle.cpu_bench import effective_cpus
    return effective_cpus()


def pytest_sessionstart(session):
    try:
        import torch
        n = _effective_cpus()
        os.environ.setdefault('OMP_NUM_THREADS', str(n))
        torch.set_num_threads(n)
    except Exception:                                      # noqa: BLE001 - a sizing hint only
        pass
