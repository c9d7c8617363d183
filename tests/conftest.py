import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_collection_finish(session):
    """PyTorch-ROCm ships its own HIP runtime; librtw_hip links the system one.  With two runtimes in one process
    the one that initialises second sees no device, and the tests that hand torch streams / tensors to the library
    need both: let torch go first whenever GPU tests are about to run."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:      # no torch / no GPU: the tests that need it fail on their own
            pass


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/librtw_oracle.so), built on demand.  Checker only."""
    import rtw_oracle
    rtw_oracle.build()
    rtw_oracle.lib()
    return rtw_oracle


@pytest.fixture(scope="session")
def rtw():
    import rtw_amd
    return rtw_amd


#: the numerics modes of the ray-sphere test the device implements (include/rtw_hip.h RTW_FLAG_NUMERICS_*); "reference" is the default
NUMERICS_MODES = ["reference", "contract", "reference_fma2"]      # (the oracle alone also knows "reference_fma": no mode of the library since ABI 4)


def current_numerics():
    from rtw_amd import _capi
    return _capi.numerics_name()


@pytest.fixture(params=NUMERICS_MODES)
def numerics(request):
    """Runs the test once per numerics mode: the mode becomes the default of BOTH sides -- the oracle (render, pixel_samples and
    the unit-level helpers) and the product's Python mirror (render, DeviceRenderer, the unit ops of test_gpu_units.run_unit) --
    and load_golden() returns that mode's expected image and counters."""
    import rtw_oracle
    from rtw_amd import _capi
    rtw_oracle.build()
    prev_o = rtw_oracle.set_numerics(request.param)
    prev_r = _capi.set_default_numerics(request.param)
    yield request.param
    rtw_oracle.set_numerics(prev_o)
    _capi.set_default_numerics(prev_r)


all_numerics = pytest.mark.usefixtures("numerics")


def load_golden(name, numerics=None):
    """-> dict with flat scene, camera dict, params and the expected image(s) / counters of one numerics mode
    (default: the mode that is current, see the `numerics` fixture)"""
    mode = numerics or current_numerics()
    suf = "" if mode == "reference" else "_" + mode
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    flat = {k[len("scene_"):]: z[k] for k in z.files if k.startswith("scene_")}
    flat["n"] = int(flat["n"])
    cam = {k[len("cam_"):]: z[k] for k in z.files if k.startswith("cam_")}
    out = dict(flat=flat, cam=cam, image=z["image" + suf], numerics=mode)
    for k in ("width", "height", "spp", "depth", "seed", "n_chunks"):
        out[k] = int(z[k])
    for k in ("segments", "rng_draws"):
        out[k] = int(z[k + suf])
    if "image_reference_order" + suf in z.files:
        out["image_reference_order"] = z["image_reference_order" + suf]
    return out


class CamObj:
    """camera dict -> object with attributes (what rtw_amd._capi.make_camera expects)"""

    def __init__(self, d):
        self.__dict__.update(d)

    @property
    def elem_type(self):
        return self.origin.dtype.type


GOLDEN_CASES = [
    "cfg1_2spheres_96x54_16spp_d4_f32",
    "smoke_2spheres_96x54_16spp_d16_f64",
    "cfg2_random_320x180_64spp_d16_f32",
    "random_64x36_8spp_d50_f64",
    "diel_bubble_96x54_8spp_d16_f32",
    "metal4_96x54_8spp_d16_f32",
    "diel_plus_96x54_8spp_d16_f32",
    "blue_red_96x54_8spp_d16_f64",
]
