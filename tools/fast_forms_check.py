"""Which settings of the fast sweep's same-bits options leave the bits of the default run, and from which step on.  Usage: python tools/fast_forms_check.py <scene> <steps>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H
scene, steps = sys.argv[1], int(sys.argv[2])
cfgs = [{}, {"fast_one_xcd": 0}, {"fast_stamps": 0}, {"fast_form": 2}, {"fast_form": 3}, {"fast_form": 1, "fast_blocks": 3}, {"fast_form": 2, "fast_blocks": 2, "fast_stamps": 0}]
ws = []
for o in cfgs:
    g = H.HipWorld(); g.scene(scene); g.w.set_option("sweep", 1)
    for k, v in o.items(): g.w.set_option(k, v)
    ws.append(g)
first = [None] * len(cfgs)
for s in range(steps):
    for g in ws: g.step()
    b0 = ws[0].bodies()
    for i, g in enumerate(ws[1:], 1):
        if first[i] is None and not (H.same_bits(b0["pos"], g.bodies()["pos"]) and H.same_bits(b0["quat"], g.bodies()["quat"])):
            d = np.abs(b0["pos"] - g.bodies()["pos"]).max(axis=1)
            first[i] = (s + 1, int(np.argmax(d)), float(d.max()), int((d > 0).sum()), ws[0].w.fast_plan_info(), g.w.fast_plan_info())
for o, f in zip(cfgs, first): print(o, "differs first at (step, body, |dpos|, bodies, plan default, plan this):" if f else "same bits", f or "")
