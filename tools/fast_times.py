"""Stats build (make -C avbd-demo3d_amd/csrc stats; AVBD_LIB=libavbd_hip_stats.so): time stamps of every tile visit of the last fast step —
how long a visit waits for the inputs of its second pass, and how long it takes from their arrival to its publish.  Usage: AVBD_LIB=libavbd_hip_stats.so python tools/fast_times.py <scene> <steps>"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H

scene, steps = sys.argv[1], int(sys.argv[2])
post_mode = "--post" in sys.argv
opts = dict((a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[3:] if "=" in a)
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
t0 = t[:, 0, 3].min() if post_mode else t[:, 0, 0].min()
t = (t - t0) * 0.01          # us
print(scene, info, "launch span", float(t[:, :, 3].max()), "us")
if post_mode:      # make -C avbd-demo3d_amd/csrc stats FASTSTATS=2: inputs in (0), rows done (1), solved (2), published (3)
    col = buf.reshape(2048, 64, 4)[:T, 63, 0].astype(int)
    for c in range(max(1, info["colors"])):
        m = col == c
        if not m.any():
            continue
        a, b, d = (t[m][:, 1:, 1] - t[m][:, 1:, 0]), (t[m][:, 1:, 2] - t[m][:, 1:, 1]), (t[m][:, 1:, 3] - t[m][:, 1:, 2])
        print(f"  colour {c}: {int(m.sum()):4d} tiles; inputs in -> rows done {np.median(a):.2f} us, -> sums and solve done {np.median(b):.2f} us, -> published {np.median(d):.2f} us")
    sys.exit(0)
passA = t[:, :, 1] - t[:, :, 0]; wait = t[:, :, 2] - t[:, :, 1]; post = t[:, :, 3] - t[:, :, 2]
print(f"  visit start -> first pass done: mean {passA.mean():.2f} us (p90 {np.percentile(passA, 90):.2f})")
print(f"  wait for second-pass inputs   : mean {wait.mean():.2f} us (p10 {np.percentile(wait, 10):.2f}, p90 {np.percentile(wait, 90):.2f})")
print(f"  inputs in -> publish          : mean {post.mean():.2f} us (p10 {np.percentile(post, 10):.2f}, p90 {np.percentile(post, 90):.2f})")
per_it = np.diff(t[:, :, 3].max(axis=0))
print(f"  per iteration (last publish to last publish): mean {per_it.mean():.2f} us -> per colour {per_it.mean() / max(1, info['colors']):.2f} us")
# per colour (the stats build leaves a tile's colour in the last iteration slot)
nc = max(1, info["colors"])
col = buf.reshape(2048, 64, 4)[:T, 63, 0].astype(int)
last = np.stack([t[col == c][:, :, 3].max(axis=0) if np.any(col == c) else np.full(its, np.nan) for c in range(nc)])        # (colour, iteration): last publish
for c in range(nc):
    m = col == c
    if not m.any():
        continue
    prev = last[c - 1] if c else np.concatenate([[np.nan], last[nc - 1][:-1]])
    hop = last[c] - prev
    print(f"  colour {c}: {int(m.sum()):4d} tiles; start -> first pass done {passA[m].mean():.2f} us, wait for the second pass's inputs {wait[m].mean():.2f} us, inputs in -> publish {post[m].mean():.2f} us (p90 {np.percentile(post[m], 90):.2f}); "
          f"last publish behind the previous colour's last publish {np.nanmedian(hop):.2f} us")
