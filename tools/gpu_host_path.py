#!/usr/bin/env python3
"""Time the host-buffer entry point (rtw_render_f32: scene upload + render + image D2H, blocking)
at the headline config -- the PCIe-inclusive rate quoted in DESIGN.md.  Needs an MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtw_amd as R
T = np.float32
R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T)
R.render(scene, cam, 1920, 8, depth=50)                      # warm-up: context, workspace allocation
for cull in (False, True):
    ts = []
    for _ in range(3):
        t = time.perf_counter(); img = R.render(scene, cam, 1920, 1000, depth=50, group_cull=cull); ts.append(time.perf_counter() - t)
    st = R.last_stats()
    best = min(ts)
    print(f"group_cull={cull}: host-path wall {best*1e3:.1f} ms = {1920*1080*1000/best/1e6:.1f} Msamples/s; "
          f"device total_ms {st['total_ms']:.1f} kernel_ms {st['kernel_ms']:.1f}; overhead {(best*1e3-st['total_ms']):.1f} ms "
          f"({(best*1e3/st['total_ms']-1)*100:.2f} %)")
