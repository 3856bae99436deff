"""Walks the critical dependency chain of one step of k_iterate_groups from per-task time stamps (-DAVBD_DF_STATS build)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import harness as H
scene = sys.argv[1] if len(sys.argv) > 1 else "StressN:20"
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 410
g = H.HipWorld(); g.scene(scene); g.step(pre); g.w.synchronize()
f = g.forces(); b = g.bodies()
g.step(1); g.w.synchronize()                       # the step whose stamps are read (topology of the step before is close enough for the walk)
IT = int(g.params()["iterations"]); n = len(b)
buf = np.zeros((IT, n, 4), np.uint32)
rc = g.pkg.lib().avbd_debug_group_task_times(g.w.h, buf.ctypes.data_as(C.c_void_p), IT)
assert rc == 0, rc
seen, pub, end, wave = (buf[..., k].astype(np.int64) for k in range(4))
dyn = b["invMass"] > 0
nb = [[] for _ in range(n)]
for a_, b_, nc in zip(f["bodyA"], f["bodyB"], f["numContacts"]):
    a_, b_ = int(a_), int(b_)
    if nc > 0 and a_ >= 0 and dyn[a_] and dyn[b_]:
        nb[a_].append(b_); nb[b_].append(a_)
t0 = seen[0][dyn].min()
last = np.unravel_index(np.argmax(np.where(dyn[None, :], pub, 0)), pub.shape)
it, body = int(last[0]), int(last[1])
print("kernel span us %.1f, last publish at %.1f us (body %d, iteration %d)" % ((end[:, dyn].max() - t0) / 100, (pub[it, body] - t0) / 100, body, it))
hops = []; hand = []; comp = []; kinds = []
while True:
    # the dependency that arrived last
    best = None
    for o in nb[body]:
        if o > body: cand = (pub[it, o], o, it, "pose")
        elif it > 0: cand = (end[it - 1, o], o, it - 1, "dual")
        else: continue
        if best is None or cand[0] > best[0]: best = cand
    own = (end[it - 1, body], body, it - 1, "self") if it > 0 else None
    if own is not None and (best is None or own[0] > best[0]): best = own
    if best is None: break
    hand.append((seen[it, body] - best[0]) / 100.0); comp.append((pub[it, body] - seen[it, body]) / 100.0); kinds.append(best[3])
    body, it = best[1], best[2]
hand, comp = np.array(hand), np.array(comp)
print("hops on the chain:", len(hand), {k: kinds.count(k) for k in set(kinds)})
print("per hop (us): producer ready -> consumer sees its inputs %.2f (median %.2f), inputs seen -> pose published %.2f" % (hand.mean(), np.median(hand), comp.mean()))
for k in set(kinds):
    sel = np.array([x == k for x in kinds])
    print("  after a %-5s dependency: wait %.2f, compute %.2f (%d hops)" % (k, hand[sel].mean(), comp[sel].mean(), sel.sum()))
dual = (end - pub)[:, dyn] / 100.0
print("pose published -> task end (dual, park of the next first half): mean %.2f us" % dual.mean())
# ---- where the long waits come from: what the consumer's wave did between the producer's publish and the moment it saw its inputs
print("wait histogram (us):", {f"<{e}": int((hand < e).sum()) for e in (2, 5, 10, 20, 40, 80, 160, 1e9)})
order = np.argsort(-hand)[:12]
flat_wave = wave.reshape(-1); flat_seen = seen.reshape(-1); flat_end = end.reshape(-1)
chain = []
it, body = int(last[0]), int(last[1])
# (re-walk to keep the (body, it, dep time) of every hop)
while True:
    best = None
    for o in nb[body]:
        if o > body: cand = (pub[it, o], o, it, "pose")
        elif it > 0: cand = (end[it - 1, o], o, it - 1, "dual")
        else: continue
        if best is None or cand[0] > best[0]: best = cand
    own = (end[it - 1, body], body, it - 1, "self") if it > 0 else None
    if own is not None and (best is None or own[0] > best[0]): best = own
    if best is None: break
    chain.append((body, it, int(best[0]), best[3], best[1]))
    body, it = best[1], best[2]
for k in order:
    b_, i_, tdep, kind, dep = chain[k]
    wv_ = int(wave[i_, b_])
    mine = (flat_wave == wv_) & (flat_seen > 0)
    between = mine & (flat_seen >= tdep) & (flat_seen < seen[i_, b_])
    running = mine & (flat_seen < tdep) & (flat_end > tdep)
    its = np.nonzero(between)[0] // n
    print("  wait %6.1f us: body %5d it %2d after %s of %5d | its wave %4d started %2d other tasks meanwhile (iterations %s), was inside a task at the publish: %s"
          % (hand[k], b_, i_, kind, dep, wv_, int(between.sum()), sorted(set(its.tolist()))[:6], bool(running.any())))
# ---- which neighbour the chain runs through: creation-index distance between a body and the later-created partner it waited for
import collections
dist = collections.Counter(int(dep) - int(b_) for (b_, i_, tdep, kind, dep) in chain if kind == "pose")
print("index distance of the awaited partner on the chain (pose hops):", dict(sorted(dist.items(), key=lambda kv: -kv[1])[:8]))
same_wg = sum(1 for (b_, i_, tdep, kind, dep) in chain if kind == "pose" and int(wave[i_, b_]) // 4 == int(wave[i_, dep]) // 4)
same_wave = sum(1 for (b_, i_, tdep, kind, dep) in chain if kind == "pose" and int(wave[i_, b_]) == int(wave[i_, dep]))
print("pose hops whose two bodies share a workgroup: %d, a wave: %d, of %d" % (same_wg, same_wave, sum(1 for x in chain if x[3] == "pose")))
