#!/usr/bin/env python3
"""image_diff.py A B  -- difference report of two images (.npy float arrays [H,W,3] or P6 .ppm)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from rtw_amd import imageio  # noqa: E402


def load(p):
    return np.load(p) if p.endswith(".npy") else imageio.load_ppm(p).astype(np.float64) / 255.0


if __name__ == "__main__":
    print(json.dumps(imageio.diff_report(load(sys.argv[1]), load(sys.argv[2])), indent=1))
