import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("wmi_models"))


@pytest.fixture(scope="session")
def model_path(model_dir):
    """Returns a function arch -> path of a seeded synthetic ggml file (written once per session)."""
    from handy_amd import ggml_file as gf
    cache = {}

    def get(arch="micro.en", seed=0, ftype=1, decisive=1.0, style="plain"):
        key = (arch, seed, ftype, decisive, style)
        if key not in cache:
            p = os.path.join(model_dir, f"{arch}_s{seed}_f{ftype}_d{decisive}_{style}.bin")
            gf.write_synthetic_model(p, arch, seed=seed, ftype=ftype, decisive=decisive, style=style)
            cache[key] = p
        return cache[key]

    return get


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as wo
    wo.build()
    return wo


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "hf_golden.npz"))
