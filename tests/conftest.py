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
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def rz():
    import reze_engine_amd
    return reze_engine_amd


@pytest.fixture(scope="session")
def rzv(rz):
    """The tools-only build that carries EVERY kernel variant (make -C reze-engine_amd/csrc variants): rest geometry through
    LDS, plain morph loads, 4 morphs in flight, the register-resident crowd kernel. The product ships only the variants a
    plan can select; the parity tests reach the others through this library. Same C ABI, bound next to the product's."""
    import subprocess
    import types
    if not os.path.exists(rz.capi.VARIANTS_LIB_PATH):      # test infrastructure, unlike the product: build it on demand (hipcc, ~90 s)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "reze-engine_amd", "csrc"), "variants"])
    lib = rz.capi.load(rz.capi.VARIANTS_LIB_PATH)
    return types.SimpleNamespace(DeformContext=lambda device=0: rz.DeformContext(device, lib=lib), lib=lib, capi=rz.capi)
