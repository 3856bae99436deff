"""Stats build (make -C avbd-demo3d_amd/csrc stats; AVBD_LIB=libavbd_hip_stats.so): time stamps of every tile visit of the last fast step —
how long a visit waits for the inputs of its second pass, and how long it takes from their arrival to its publish.  Usage: AVBD_LIB=libavbd_hip_stats.so python tools/fast_times.py <scene> <steps>"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H

scene, steps = sys.argv[1], int(sys.argv[2])
opts = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[3:])
g = H.HipWorld(); g.scene(scene); g.w.set_option("sweep", 1)
for k, v in opts.items():
    g.w.set_option(k, v)
g.step(steps); g.w.synchronize()
info = g.w.fast_plan_info()
its = int(g.params()["iterations"])
T = min(info["tiles"], 2048)
buf = np.zeros(2048 * 64 * 4, np.uint32)
n = g.pkg.lib().avbd_debug_fast_times(buf.ctypes.data_as(C.POINTER(C.c_uint)), len(buf))
t = buf.reshape(2048, 64, 4)[:T, :its].astype(np.int64)
t0 = t[:, 0, 0].min()
t = (t - t0) * 0.01          # us
print(scene, info, "launch span", float(t[:, :, 3].max()), "us")
passA = t[:, :, 1] - t[:, :, 0]; wait = t[:, :, 2] - t[:, :, 1]; post = t[:, :, 3] - t[:, :, 2]
print(f"  visit start -> first pass done: mean {passA.mean():.2f} us (p90 {np.percentile(passA, 90):.2f})")
print(f"  wait for second-pass inputs   : mean {wait.mean():.2f} us (p10 {np.percentile(wait, 10):.2f}, p90 {np.percentile(wait, 90):.2f})")
print(f"  inputs in -> publish          : mean {post.mean():.2f} us (p10 {np.percentile(post, 10):.2f}, p90 {np.percentile(post, 90):.2f})")
per_it = np.diff(t[:, :, 3].max(axis=0))
print(f"  per iteration (last publish to last publish): mean {per_it.mean():.2f} us -> per colour {per_it.mean() / max(1, info['colors']):.2f} us")
