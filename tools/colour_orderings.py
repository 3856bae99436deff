"""How many colours greedy colourings of a scene's contact graph need under different vertex orders: largest degree first with hashed ties (what k_fast_plan's Jones-Plassmann
rounds compute), smallest-last (degeneracy order), DSATUR, random.  Usage: python tools/colour_orderings.py <scene> <steps>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H
scene, steps = sys.argv[1], int(sys.argv[2])
g = H.HipWorld(); g.scene(scene); g.w.set_option("sweep", 1); g.step(steps)
b, f = g.bodies(), g.forces()
dyn = b["invMass"] > 0
n = len(b)
adj = [set() for _ in range(n)]
for a, c in zip(f["bodyA"], f["bodyB"]):
    a, c = int(a), int(c)
    if a >= 0 and c >= 0 and dyn[a] and dyn[c]:
        adj[a].add(c); adj[c].add(a)
verts = [v for v in range(n) if dyn[v] and adj[v]]
print(scene, "step", steps, "plan", g.w.fast_plan_info(), "vertices with neighbours", len(verts), "edges", sum(len(adj[v]) for v in verts) // 2, "max degree", max(len(adj[v]) for v in verts))
def greedy(order):
    col = {}
    for v in order:
        used = {col[u] for u in adj[v] if u in col}
        c = 0
        while c in used: c += 1
        col[v] = c
    return col
def report(name, col):
    k = max(col.values()) + 1
    print(f"  {name:28s}: {k} colours, class sizes {np.bincount(list(col.values())).tolist()}")
rng = np.random.default_rng(1)
deg = {v: len(adj[v]) for v in verts}
for seed in range(3):
    h = {v: int(rng.integers(1 << 30)) for v in verts}
    report(f"largest degree first (seed {seed})", greedy(sorted(verts, key=lambda v: (-min(deg[v], 127), -h[v]))))
# smallest-last
import heapq
d = dict(deg); removed = set(); order = []
heap = [(d[v], v) for v in verts]; heapq.heapify(heap)
while heap:
    dv, v = heapq.heappop(heap)
    if v in removed or dv != d[v]: continue
    removed.add(v); order.append(v)
    for u in adj[v]:
        if u not in removed:
            d[u] -= 1; heapq.heappush(heap, (d[u], u))
report("smallest last (degeneracy)", greedy(order[::-1]))
# DSATUR
col = {}; sat = {v: set() for v in verts}; left = set(verts)
while left:
    v = max(left, key=lambda x: (len(sat[x]), deg[x]))
    used = sat[v]; c = 0
    while c in used: c += 1
    col[v] = c; left.discard(v)
    for u in adj[v]:
        if u in left: sat[u].add(c)
report("DSATUR", col)
for seed in range(2):
    report(f"random order (seed {seed})", greedy(list(rng.permutation(verts))))
# parallel approximations of the degeneracy order: peel in ROUNDS, colour in reverse removal order (later removed = earlier coloured), ties by hash
def peel(rule):
    d = dict(deg); alive = set(verts); rank = {}; r = 0; k = 0
    while alive:
        if rule == "adg":
            avg = sum(d[v] for v in alive) / len(alive)
            batch = [v for v in alive if d[v] <= 1.1 * avg]
        else:                                   # "level": everything at or below the current threshold; the threshold rises only when nothing is left below it
            batch = [v for v in alive if d[v] <= k]
            if not batch:
                k += 1; continue
        for v in batch: rank[v] = r
        for v in batch:
            alive.discard(v)
        for v in batch:
            for u in adj[v]:
                if u in alive: d[u] -= 1
        r += 1
    return rank, r
for rule in ("adg", "level"):
    rank, rounds = peel(rule)
    for seed in range(2):
        h = {v: int(rng.integers(1 << 30)) for v in verts}
        report(f"peel '{rule}' ({rounds} rounds, seed {seed})", greedy(sorted(verts, key=lambda v: (-rank[v], -h[v]))))
# iterated greedy (Culberson): recolour class by class — the vertices of a class are mutually non-adjacent, so a class is one parallel round; a pass never needs more colours
def ig_pass(col, class_order):
    new = {}
    for c in class_order:
        for v in [v for v in verts if col[v] == c]:
            used = {new[u] for u in adj[v] if u in new}
            k = 0
            while k in used: k += 1
            new[v] = k
    return new
h = {v: int(rng.integers(1 << 30)) for v in verts}
col = greedy(sorted(verts, key=lambda v: (-min(deg[v], 127), -h[v])))
report("LDF start", col)
for p in range(8):
    k = max(col.values()) + 1
    sizes = np.bincount(list(col.values()))
    order = list(range(k))[::-1] if p % 2 == 0 else list(np.argsort(-sizes))      # reverse colour order, then largest class first, alternating
    col = ig_pass(col, order)
    report(f"  iterated greedy pass {p + 1} ({'reverse' if p % 2 == 0 else 'largest first'})", col)
# Jones-Plassmann depth (rounds = longest path of the priority DAG) under the two kinds of priority
def jp_depth(prio):
    depth = {}
    for v in sorted(verts, key=lambda v: -prio[v]):
        depth[v] = 1 + max([depth[u] for u in adj[v] if prio[u] > prio[v]], default=0)
    return max(depth.values())
h = {v: int(rng.integers(1 << 30)) for v in verts}
print("  JP rounds: largest degree first", jp_depth({v: (min(deg[v], 127) << 32) | h[v] for v in verts}), "| level peel", jp_depth({v: (peel("level")[0][v] << 32) | h[v] for v in verts}), "| adg peel", jp_depth({v: (peel("adg")[0][v] << 32) | h[v] for v in verts}))
