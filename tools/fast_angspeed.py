"""How much the tail statistics of a chaotic pile depend on the (equally valid) order of the sweep: Stress1000, 600 steps in the fast mode with several colouring seeds /
forms, max angular speed over the last 100 steps and the distribution of |angVel| at step 600, next to the exact mode and the reference's diagnostics.  Usage: python tools/fast_angspeed.py"""
import os, sys
import numpy as np
sys.path.insert(0, "tests")
import harness as H
G = "tests/golden"
diags = np.load(os.path.join(G, "stress1000_diag_600.npy")).view(H.DIAG_DT)
for opts in ({"sweep": 0}, {}, {"fast_seed": 1}, {"fast_seed": 2}, {"fast_seed": 3}, {"fast_seed": 4}, {"fast_seed": 5}, {"fast_seed": 6}, {"fast_seed": 7}, {"fast_recolor": 1}, {"fast_recolor": 8}):
    g = H.HipWorld(); g.scene("Stress1000"); g.w.set_option("sweep", 1)
    for k, v in opts.items(): g.w.set_option(k, v)
    rec = []; cols = []
    for s in range(600):
        g.step(); d = g.diag()
        rec.append((float(d["maxLinearSpeed"]), float(d["maxAngularSpeed"]), int(d["activeContacts"]), int(d["activeManifolds"])))
        if s >= 300: cols.append(g.w.fast_plan_info()["colors"])
    rec = np.array(rec)
    b = g.bodies(); w = np.linalg.norm(b["angVel"], axis=1); k = int(np.argmax(w))
    print(opts, "max ang 500..600", rec[500:, 1].max(), "ref", diags["maxAngularSpeed"][500:].max(), "at end body", k, w[k], "pos", b["pos"][k], "linvel", b["linVel"][k], "colours hist", np.bincount(cols), "contacts", rec[599, 2], int(diags["activeContacts"][599]))
    rc = rec[300:, 2] / diags["activeContacts"][300:]; rm = rec[300:, 3] / diags["activeManifolds"][300:]
    print("   contacts / reference from step 300: [%.3f, %.3f], manifolds [%.3f, %.3f]; first contact step %d; max linear speed 500..600 %.2f (ref %.2f)" % (rc.min(), rc.max(), rm.min(), rm.max(), int(np.nonzero(rec[:, 2] > 0)[0][0]) + 1, rec[500:, 0].max(), diags["maxLinearSpeed"][500:].max()))
    print("   ang speed every 50 steps:", np.round(rec[::50, 1], 2))
    dyn = b["invMass"] > 0
    print("   |angVel| at step 600: p50 %.3f p90 %.3f p99 %.3f max %.3f; bodies above 1 rad/s: %d; |linVel| p99 %.3f" % (*np.percentile(w[dyn], [50, 90, 99]), w[dyn].max(), int((w[dyn] > 1.0).sum()), np.percentile(np.linalg.norm(b["linVel"][dyn], axis=1), 99)))
print("ref ang every 50:", np.round(diags["maxAngularSpeed"][::50], 2))
