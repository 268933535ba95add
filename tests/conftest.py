import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def api(pkg):
    return pkg.api


@pytest.fixture(scope="session")
def oracle(pkg):
    """CPU oracle backend (test infrastructure)."""
    if not os.path.exists(graft.ORACLE_LIB):
        graft.build_oracle()
    return graft.oracle_backend()


@pytest.fixture(scope="session")
def emulated(pkg):
    """The product kernel sources compiled for the host (tests/hip_emu) - CPU-side check of the kernels themselves."""
    path = graft.build_emulated()
    b = pkg.api.Backend(path, "nrdhip_", "cpu")
    b.check_abi()
    return b


@pytest.fixture(scope="session")
def hip(pkg):
    """The product: libnrdhip.so on cuda:0. Fails (does not skip) when the library or the device is missing."""
    return pkg.hip_backend("cuda:0")


# ---- the NRD_UPSTREAM_FORMULAS=0 build flavour (the cheaper forms of ledger rows 1, 2, 7, 13 that rounds 1-3 had frozen; oracle/README.md) ----
@pytest.fixture(scope="session")
def oracle_frozen(pkg):
    if not os.path.exists(graft.ORACLE_LIB_FROZEN):
        graft.build_oracle()
    return graft.oracle_backend("frozen")


@pytest.fixture(scope="session")
def emulated_frozen(pkg):
    path = graft.build_emulated(flavour="frozen")
    b = pkg.api.Backend(path, "nrdhip_", "cpu")
    b.check_abi()
    return b


@pytest.fixture(scope="session")
def hip_frozen(pkg):
    return pkg.hip_backend("cuda:0", flavour="frozen")


# ---- the NRD_HW_TRANSCENDENTALS=1 build flavour (v_rcp_f32 / v_sqrt_f32 / v_exp_f32 in the weight arithmetic of the spatial filters) ----
@pytest.fixture(scope="session")
def oracle_hwt(pkg):
    if not os.path.exists(graft.ORACLE_LIB_HWT):
        graft.build_oracle()
    return graft.oracle_backend("hwt")


@pytest.fixture(scope="session")
def emulated_hwt(pkg):
    path = graft.build_emulated(flavour="hwt")
    b = pkg.api.Backend(path, "nrdhip_", "cpu")
    b.check_abi()
    return b


@pytest.fixture(scope="session")
def hip_hwt(pkg):
    return pkg.hip_backend("cuda:0", flavour="hwt")
