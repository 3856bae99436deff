import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
g = H.HipWorld(); g.scene(sys.argv[1] if len(sys.argv) > 1 else "Stress1000"); g.step(420); g.w.synchronize()
acc = np.zeros(3)
for k in range(20):
    g.step(1); g.w.synchronize()
    b = np.zeros(4, np.uint64)
    g.pkg.lib().avbd_debug_front_times(g.w.h, b.ctypes.data_as(C.c_void_p))
    acc += np.diff(b.astype(np.int64)) / 100.0
for n, v in zip(["loads", "collide+initialize", "decay+append+stores"], acc / 20): print(f"{n:22s} {v:6.2f} us")
# every workgroup of the last launch: when it started and how long it lived (persistent-manifold workgroups first, then the broadphase ones)
NB = 8192
blk = np.zeros(16 * NB, np.uint64)
g.pkg.lib().avbd_debug_front_blocks(g.w.h, blk.ctypes.data_as(C.c_void_p), NB)
blk = blk.reshape(NB, 16).astype(np.int64)
used = blk[:, 0] > 0
if used.any():
    t0 = blk[used, 0].min()
    st = (blk[used, 0] - t0) / 100.0
    en = (np.where(blk[used, 15] > 0, blk[used, 15], blk[used, 0]) - t0) / 100.0
    life = en - st
    n = int(used.sum())
    print(f"{n} workgroups; last start {st.max():.2f} us, last end {en.max():.2f} us")
    for lo in range(0, n, max(1, n // 12)):
        hi = min(n, lo + max(1, n // 12))
        print(f"  workgroups {lo:5d}..{hi - 1:5d}: start {st[lo:hi].min():6.2f}..{st[lo:hi].max():6.2f} us  life mean {life[lo:hi].mean():5.2f} max {life[lo:hi].max():5.2f} us")
    # the first pair of the first group of every workgroup that had one, stage by stage (face contacts: every stamp present)
    order = [0, 1, 4, 5, 6, 7, 8, 9, 10, 2, 3, 15]
    names = ["loads", "SAT axes", "axis chosen", "incident face", "four clips", "contacts chosen", "contacts stored", "features matched", "per-contact init", "decay+append+stores", "to workgroup end"]
    pp = blk[used][:, order]
    ok = (pp > 0).all(axis=1)
    if ok.any():
        d = np.diff(pp[ok], axis=1) / 100.0
        print(f"first pair of group 0 in {int(ok.sum())} workgroups (face contacts), us per stage, mean (max):")
        for k, nm in enumerate(names): print(f"  {nm:22s} {d[:, k].mean():6.2f} ({d[:, k].max():6.2f})")
