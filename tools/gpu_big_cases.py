import sys, os, time
sys.path[:0] = ["/root/repo", "/root/repo/oracle"]
import numpy as np, rtw_amd as R, rtw_oracle as O
T = np.float32
R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T)
t = time.time(); img = R.render(scene, cam, 7680, 1, depth=50); print("8K render", img.shape, time.time() - t, R.last_stats()["samples"])
ref, _ = O.render(R.flatten_scene(scene, T), cam, 7680, 4320, 1, T=T, max_depth=50, seed=1, n_chunks=1)
print("8K bit-exact vs oracle:", np.array_equal(img, ref))
img = R.render(scene, cam, 64, 5000, depth=1000); ref, _ = O.render(R.flatten_scene(scene, T), cam, 64, 36, 5000, T=T, max_depth=1000, seed=1)
print("5000 spp / depth 1000 bit-exact:", np.array_equal(img, ref), R.last_stats()["n_chunks"])
T = np.float64
R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T)
img = R.render(scene, cam, 200, 300, depth=50, devices=[0, 0]); ref, _ = O.render(R.flatten_scene(scene, T), cam, 200, 112, 300, T=T, max_depth=50, seed=1)
print("f64 300 spp 2 shards bit-exact:", np.array_equal(img, ref))
