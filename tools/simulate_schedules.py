"""Event-driven model of the resident schedule with several bodies per wave (not product, not a test).

Input: contact graphs dumped on the GPU box by tools/dump_contact_graph.py (gpurun_out/graphs/<tag>.npz).
Model: a wave prepares a task (PREP), waits for the records of its partners (a partner's pose is visible HAND after its
publish, its dual DUAL later), publishes POST after the last input, and is busy TAIL more (dual + parking).  The constants
are the stage times measured with tools/debug_resident.py on an MI355X.  Calibration: Stress1000 one body per wave 741 us
(measured k_iterate ~700), 8 000 cubes in order 2.67 ms (measured 2.36-2.7), 64 000 cubes in order 11.2 ms (measured 11.2).

    python tools/simulate_schedules.py s20 2048      # tag, resident waves

What it showed: with several bodies per wave the sweep is bound by the waves being serial resources,
not by the dependency chain (8 000 cubes: 0.9 ms with a wave per body, 1.7 ms with two bodies per wave, 2.7 ms with four);
static orders by any potential (round, dependency level) deadlock because contacts span up to 40 levels; an out-of-order
pick that examines a body after its inputs arrived loses to the in-order sweep unless a wave owns many bodies.
"""
import os, sys, heapq, numpy as np, time
def load(tag):
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "graphs", f"{tag}.npz"))
    a, b, nc, inv = z["a"], z["b"], z["nc"], z["inv"]
    dyn = inv > 0
    ids = np.nonzero(dyn)[0][::-1]           # sweep order: descending creation index
    slot = -np.ones(len(inv), int); slot[ids] = np.arange(len(ids))
    nb = [[] for _ in range(len(ids))]
    for x, y, c in zip(a, b, nc):
        if c <= 0 or x < 0 or y < 0 or not (dyn[x] and dyn[y]): continue
        sx, sy = slot[x], slot[y]
        nb[sx].append(sy); nb[sy].append(sx)
    return len(ids), nb

PREP, POST, TAIL, HAND, DUAL = 4.8, 2.6, 1.7, 0.9, 1.1

def static_sim(nd, nb, W, IT, order_fn):
    """per-wave static task lists, blocking waits with the head prepared while waiting"""
    lists = []
    for w in range(min(W, nd)):
        ks = list(range(w, nd, W))
        lists.append(order_fn(w, ks, IT))
    P = {}          # (slot,it) -> publish time
    waiting = {}    # (slot,it) -> waves blocked on it
    head = [0] * len(lists); free = [0.0] * len(lists)
    ready = list(range(len(lists)))
    done = 0; total = nd * IT; tmax = 0.0
    def try_run(w):
        nonlocal done, tmax
        while head[w] < len(lists[w]):
            s, it = lists[w][head[w]]
            r = 0.0
            for o in nb[s]:
                if o < s: key = (o, it); extra = HAND
                elif it > 0: key = (o, it - 1); extra = HAND + DUAL
                else: continue
                t = P.get(key)
                if t is None:
                    waiting.setdefault(key, []).append(w); return
                if t + extra > r: r = t + extra
            pub = max(free[w] + PREP, r) + POST
            P[(s, it)] = pub; free[w] = pub + TAIL; head[w] += 1; done += 1
            if free[w] > tmax: tmax = free[w]
            for w2 in waiting.pop((s, it), []): ready.append(w2)
    while ready:
        try_run(ready.pop())
    assert done == total, (done, total, "DEADLOCK")
    return tmax

def inorder(w, ks, IT): return [(s, it) for it in range(IT) for s in ks]
def phi(lmb):
    def f(w, ks, IT):
        t = [(it * lmb + k, k, s, it) for it in range(IT) for k, s in enumerate(ks)]
        t.sort()
        return [(s, it) for _, _, s, it in t]
    return f

def ooo_sim(nd, nb, W, IT, clairvoyant, exam=2.5):
    """event driven: a wave that is free takes, among its bodies' next tasks, the one that is (or will first be) ready.
    clairvoyant: it knows which will be ready first and has it prepared (latency POST after ready);
    otherwise it detects readiness, then examines + prepares (exam + PREP after ready)."""
    nW = min(W, nd)
    nxt = {}                                        # slot -> next iteration
    for s in range(nd): nxt[s] = 0
    P = {}
    # compute ready time lazily: need all deps published; use event simulation over time
    # each wave: state free time; bodies list
    bodies = [list(range(w, nd, W)) for w in range(nW)]
    free = [0.0] * nW
    # dependency counting: remaining deps per task
    def deps(s, it):
        for o in nb[s]:
            if o < s: yield (o, it), HAND
            elif it > 0: yield (o, it - 1), HAND + DUAL
    readyAt = {}                                    # task -> time all deps are visible (only when all published)
    pending = {}                                    # task -> count of unpublished deps
    rev = {}
    def arm(s, it):
        cnt = 0; r = 0.0
        for key, extra in deps(s, it):
            t = P.get(key)
            if t is None: cnt += 1; rev.setdefault(key, []).append((s, it))
            else: r = max(r, t + extra)
        pending[(s, it)] = cnt; readyAt[(s, it)] = r
    for s in range(nd): arm(s, 0)
    ev = [(0.0, w) for w in range(nW)]              # wave wake-up events
    heapq.heapify(ev)
    asleep = {}                                     # wave -> True if waiting for any readiness
    wave_of = lambda s: s % W
    done = 0; tmax = 0.0
    while ev:
        now, w = heapq.heappop(ev)
        if now < free[w]: heapq.heappush(ev, (free[w], w)); continue
        # candidate tasks
        best = None
        for s in bodies[w]:
            it = nxt[s]
            if it >= IT: continue
            if pending[(s, it)] == 0:
                r = readyAt[(s, it)]
                key = (max(r, now), it, s)
                if best is None or key < best: best = key
        if best is None:
            if all(nxt[s] >= IT for s in bodies[w]): continue
            asleep[w] = True; continue
        r, it, s = best
        if r > now and not clairvoyant:
            # something may become ready earlier than r? all deps published already, r is known future time: just wait
            pass
        if clairvoyant: pub = max(now + PREP, r) + POST
        else: pub = max(now, r) + exam + PREP + POST
        P[(s, it)] = pub; free[w] = pub + TAIL; nxt[s] = it + 1; done += 1
        if free[w] > tmax: tmax = free[w]
        if it + 1 < IT: arm(s, it + 1)
        for t in rev.pop((s, it), []):
            pending[t] -= 1; readyAt[t] = max(readyAt[t], pub + (HAND if t[1] == it and t[0] > s else HAND + DUAL))
            if pending[t] == 0:
                w2 = wave_of(t[0])
                if asleep.pop(w2, None): heapq.heappush(ev, (max(pub, free[w2]), w2))
        heapq.heappush(ev, (free[w], w))
    assert done == nd * IT, (done, nd * IT)
    return tmax

if __name__ == "__main__":
    tag = sys.argv[1]; W = int(sys.argv[2])
    nd, nb = load(tag)
    R = max((abs(o - s) + W - 1) // W for s in range(nd) for o in nb[s]) if nd else 0
    Rfloor = max(abs(o // W - s // W) for s in range(nd) for o in nb[s])
    print(tag, "nd", nd, "W", W, "K", (nd + W - 1) // W, "max round distance", Rfloor, "(ceil", R, ")")
    IT = 20
    t0 = time.time()
    print(" in-order           %.0f us" % static_sim(nd, nb, W, IT, inorder))
    for lmb in (Rfloor + 1, 2, 3, 4):
        try: print(" phi lambda=%d       %.0f us" % (lmb, static_sim(nd, nb, W, IT, phi(lmb))))
        except AssertionError as e: print(" phi lambda=%d DEADLOCK" % lmb)
    print(" ooo clairvoyant    %.0f us" % ooo_sim(nd, nb, W, IT, True))
    print(" ooo detect+prepare %.0f us" % ooo_sim(nd, nb, W, IT, False))
    print(" (%.0f s)" % (time.time() - t0))
