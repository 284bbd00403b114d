import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_rgba(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name)).convert("RGBA"))


def sorted_palette(name):
    """cli/src/args.rs:197-216 parse_palette: pixels of the palette image, sorted, unique."""
    px = load_rgba(name).reshape(-1, 4)
    return np.array(sorted(set(map(tuple, px))), np.uint8)


@pytest.fixture(scope="session")
def tokyo():
    return load_rgba("tokyo.png")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch


@pytest.fixture(scope="session")
def processor(torch_cuda):
    import kmeans_gpu_amd as kg
    p = kg.ImageProcessor()
    yield p
    p.close()


def set_strategy(strategy):
    """kmg_options.strategy of every live processor of this process and of the ones created later (kmeans_gpu_amd.set_strategy):
    "auto" | "scan" (alias "brute") | "table" [+ "mask_words"].  Results are identical either way -- which is what the tests that
    call this assert.  Reset to "auto" after every test (below)."""
    import kmeans_gpu_amd as kg
    kg.set_strategy(strategy)


@pytest.fixture(autouse=True)
def _strategy_back_to_auto():
    yield
    if "kmeans_gpu_amd" in sys.modules:
        sys.modules["kmeans_gpu_amd"].set_strategy("auto")
