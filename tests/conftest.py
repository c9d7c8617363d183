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


def load_golden(name):
    """-> dict with flat scene, camera dict, params and expected image(s)"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    flat = {k[len("scene_"):]: z[k] for k in z.files if k.startswith("scene_")}
    flat["n"] = int(flat["n"])
    cam = {k[len("cam_"):]: z[k] for k in z.files if k.startswith("cam_")}
    out = dict(flat=flat, cam=cam, image=z["image"])
    for k in ("width", "height", "spp", "depth", "seed", "n_chunks", "segments", "rng_draws"):
        out[k] = int(z[k])
    if "image_reference_order" in z.files:
        out["image_reference_order"] = z["image_reference_order"]
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
