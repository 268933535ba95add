"""nrd-sample_amd: MI355X-native NRD denoiser dispatch backend (REBLUR / RELAX / SIGMA / REFERENCE).

Layout:  csrc/      HIP kernels + the C-ABI (libnrdhip.so) + the C++ nrd:: API on top of it
         api.py     ctypes twin of nrd::Integration over the C-ABI
         synth.py   synthetic G-buffer / noisy radiance generator in the sample's encodings
         harness.py the sample's per-frame call sequence (Sample::RenderFrame's NRD part), headless

The directory name carries a hyphen, so import it through ``__graft_entry__.load_package()``
(registers it as ``nrd_sample_amd``).
"""
import os

from . import api, harness, sample_tests, synth  # noqa: F401

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
HIP_LIB = os.path.join(PKG_DIR, "csrc", "libnrdhip.so")
# the same kernels built with the cheaper forms of four formulas that rounds 1-3 had frozen (csrc/nrd_device.h NRD_UPSTREAM_FORMULAS = 0)
HIP_LIB_FROZEN = os.path.join(PKG_DIR, "csrc", "libnrdhip_frozen.so")
# the same kernels with v_rcp_f32 / v_sqrt_f32 / v_exp_f32 in the weight arithmetic of the spatial filters (csrc/nrd_device.h NRD_HW_TRANSCENDENTALS = 1)
HIP_LIB_HWT = os.path.join(PKG_DIR, "csrc", "libnrdhip_hwt.so")
HIP_LIBS = {None: HIP_LIB, "frozen": HIP_LIB_FROZEN, "hwt": HIP_LIB_HWT}


def hip_backend(device="cuda:0", flavour=None):
    """The product path. Fails loudly when the HIP library is missing or no GPU is visible - there is no CPU fallback.
    ``flavour="frozen"``: libnrdhip_frozen.so (bench.py's config.frozen_formulas leg, tests); ``flavour="hwt"``: libnrdhip_hwt.so."""
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("nrd-sample_amd: no HIP device visible; the denoiser passes are HIP kernels and have no CPU fallback")
    b = api.Backend(HIP_LIBS[flavour], "nrdhip_", device)
    b.check_abi()
    return b


def hip_library_symbols():
    """Load libnrdhip.so without touching a device (CPU-side ABI checks)."""
    b = api.Backend(HIP_LIB, "nrdhip_", "cuda:0")
    b.check_abi()
    return b
