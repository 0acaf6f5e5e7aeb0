import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    # The oracle's torch-CPU convolutions size their OpenMP teams by the VISIBLE core count (256 on the GPU boxes) while the
    # container may only use a quota of them (cgroup cpu.max: 16): the spinning surplus gets the whole process throttled.
    try:
        import torch

        n = min(os.cpu_count() or 1, 32)
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                n = max(1, min(n, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
        torch.set_num_threads(n)
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def bottomup_model_dir():
    return os.path.join(GOLDEN, "models", "minimal_instance.UNet.bottomup")
