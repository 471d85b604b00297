import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import sk_oracle
    sk_oracle.lib()
    return sk_oracle


@pytest.fixture(scope="session")
def sk():
    """The product package; building it here is allowed (hipcc cross-compiles without a GPU)."""
    # no torch here: the GPU tests run as a Julia caller would, with the library's own HIP runtime only
    import __graft_entry__
    lib = os.path.join(ROOT, "spectralkernels.jl_amd", "libspectralkernels_hip.so")
    if not os.path.exists(lib):
        __graft_entry__.build()
    import spectralkernels_jl_amd
    return spectralkernels_jl_amd
