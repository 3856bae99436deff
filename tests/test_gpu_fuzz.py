"""Randomised differential test: random piles of boxes (sizes, densities, frictions, poses, velocities), sometimes with joints,
springs and ignore-collision markers, random solver parameters including post-stabilisation — every step compared bit for bit
(bodies, force list with rows and contacts, diagnostics) with the CPU oracle.  Found the missing Force::C store of joints and
springs in post-stabilisation visits."""
import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu


def build(w, seed):
    r = np.random.default_rng(seed)
    n = int(r.integers(3, 160))
    w.add_body((float(r.uniform(10, 60)), 1.0, float(r.uniform(10, 60))), 0.0, float(r.uniform(0.1, 1.0)), (0, -0.5, 0))
    if r.random() < 0.4:
        w.add_body((float(r.uniform(2, 8)), float(r.uniform(0.5, 3)), float(r.uniform(2, 8))), 0.0, 0.6, (float(r.uniform(-3, 3)), 1.0, float(r.uniform(-3, 3))))
    for k in range(n):
        size = tuple(float(x) for x in r.uniform(0.3, 1.8, 3))
        pos = (float(r.uniform(-4, 4)), float(r.uniform(0.4, 9)), float(r.uniform(-4, 4)))
        q = r.normal(size=4); q = tuple(float(x) for x in q / np.linalg.norm(q))
        lin = tuple(float(x) for x in r.uniform(-3, 3, 3)); ang = tuple(float(x) for x in r.uniform(-5, 5, 3))
        w.add_body(size, float(r.uniform(0.2, 5.0)), float(r.uniform(0.0, 1.2)), pos, q, lin, ang)
    if r.random() < 0.5:
        for _ in range(int(r.integers(1, 6))):
            a, b = int(r.integers(1, n)), int(r.integers(1, n))
            if a != b:
                kind = r.integers(0, 3)
                if kind == 0: w.add_joint(a, b, (0.4, 0, 0), (-0.4, 0, 0))
                elif kind == 1: w.add_spring(a, b, (0, 0.2, 0), (0, -0.2, 0), float(r.uniform(50, 5000)))
                else: w.add_ignore(a, b)
    w.set_params(iterations=int(r.integers(1, 31)), alpha=float(r.uniform(0.5, 1.0)), beta=float(r.uniform(1e3, 2e5)), gamma=float(r.uniform(0.9, 1.0)),
                 postStabilize=int(r.integers(0, 2)), gravity=(float(r.uniform(-2, 2)), float(r.uniform(-12, -5)), float(r.uniform(-2, 2))))
    return n


@pytest.mark.parametrize("first", [100, 120, 140, 160])
def test_random_scenes_bit_exact(first):
    for seed in range(first, first + 20):
        o, g = H.CpuWorld("orc"), H.HipWorld()
        for w in (o, g):
            build(w, seed)
        for s in range(18):
            o.step(); g.step()
            H.assert_same_world(o, g, f"seed {seed} step {s + 1}")
