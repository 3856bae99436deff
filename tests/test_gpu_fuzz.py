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


def build_jointed(w, seed, big=False):
    r = np.random.default_rng(seed)
    n = int(r.integers(20, 160)) if not big else int(r.integers(1100, 1900))
    w.add_body((80, 1.0, 80), 0.0, 0.5, (0, -0.5, 0))
    span = 4 if not big else 14
    for k in range(n):
        size = tuple(float(x) for x in r.uniform(0.4, 1.4, 3))
        pos = (float(r.uniform(-span, span)), float(r.uniform(0.4, 7)), float(r.uniform(-span, span)))
        q = r.normal(size=4); q = tuple(float(x) for x in q / np.linalg.norm(q))
        w.add_body(size, float(r.uniform(0.3, 3.0)), float(r.uniform(0.1, 1.0)), pos, q, tuple(float(x) for x in r.uniform(-2, 2, 3)),
                   tuple(float(x) for x in r.uniform(-3, 3, 3)))
    for _ in range(int(r.integers(3, 25))):
        a, b = int(r.integers(1, n)), int(r.integers(1, n))
        if a == b:
            continue
        kind = r.integers(0, 4)
        if kind == 0: w.add_joint(a, b, (0.4, 0, 0), (-0.4, 0, 0))
        elif kind == 1: w.add_joint(-1, b, (float(r.uniform(-3, 3)), 3.0, float(r.uniform(-3, 3))), (0, 0.3, 0))      # world anchor
        elif kind == 2: w.add_joint(a, b, (0, 0.3, 0), (0, -0.3, 0), lin_stiff=float(r.uniform(1e3, 1e5)), ang_stiff=H.FLT_MAX)  # soft rows
        else: w.add_ignore(a, b)
    w.set_params(iterations=int(r.integers(1, 25)), postStabilize=int(r.integers(0, 2)), beta=float(r.uniform(1e4, 2e5)))


@pytest.mark.parametrize("first", [300, 320])
def test_random_jointed_scenes_run_the_resident_schedule_bit_exact(first):
    """Joints (hard, soft, world-anchored) and ignore-collision markers but no springs: the resident schedule takes them
    (a lane holds one half = three rows of a joint); every step compared with the oracle."""
    for seed in range(first, first + 20):
        o, g = H.CpuWorld("orc"), H.HipWorld()
        for w in (o, g):
            build_jointed(w, seed)
        for s in range(18):
            o.step(); g.step()
            H.assert_same_world(o, g, f"seed {seed} step {s + 1}")
        assert int(g.w.stats()["schedule"]) == 2 and int(g.w.stats()["numJoints"]) > 0


def test_jointed_scene_with_several_bodies_per_wave(monkeypatch):
    monkeypatch.setenv("AVBD_GROUPS", "0")                    # (since round 5 the group schedule takes jointed scenes: tests/test_gpu_groups.py)
    for seed in (9001, 9003):
        o, g = H.CpuWorld("orc"), H.HipWorld()
        for w in (o, g):
            build_jointed(w, seed, big=True)
        for s in range(10):
            o.step(); g.step()
            H.assert_same_world_fast(o, g, f"seed {seed} step {s + 1}", check_rows=True)
        assert int(g.w.stats()["schedule"]) == 3


@pytest.mark.parametrize("lite", ["0", "1"])
def test_random_jointed_scenes_out_of_order_sweep(monkeypatch, lite):
    """Few resident waves (AVBD_ITER_BLOCKS=1: 4 or 8 waves) so that a wave owns up to 40 bodies, swept out of order
    (AVBD_RESIDENT_OOO=1: whichever body has its inputs published, oldest iteration first), with one or two workgroups per CU.
    The order of execution changes, the data flow does not: every step bit for bit."""
    monkeypatch.setenv("AVBD_ITER_BLOCKS", "1")
    monkeypatch.setenv("AVBD_GROUPS", "0")                    # (the parked sweep of the resident schedule is what this test is about)
    monkeypatch.setenv("AVBD_RESIDENT_OOO", "1")
    monkeypatch.setenv("AVBD_RESIDENT_FORCE_LITE", lite)
    monkeypatch.setenv("AVBD_RESIDENT_LITE_FROM", "0")
    for seed in range(400, 412):
        o, g = H.CpuWorld("orc"), H.HipWorld()
        for w in (o, g):
            build_jointed(w, seed)
        for s in range(14):
            o.step(); g.step()
            H.assert_same_world(o, g, f"seed {seed} step {s + 1}")
        assert int(g.w.stats()["schedule"]) == 3


def test_out_of_order_sweep_in_the_per_iteration_launches_of_a_sharded_run(monkeypatch):
    """The one-launch-per-iteration schedule of sharded runs (here with one rank: RCCL communicator, all-gather after every
    iteration) combined with several bodies per wave swept out of order: parked state, stamped records and watched records carry
    over from launch to launch.  Bit for bit against the oracle."""
    monkeypatch.setenv("AVBD_ITER_BLOCKS", "1")
    monkeypatch.setenv("AVBD_RESIDENT_OOO", "1")
    monkeypatch.setenv("AVBD_FORCE_SHARDED", "1")
    pkg = H.load_product()
    for seed in range(420, 428):
        o, g = H.CpuWorld("orc"), H.HipWorld()
        for w in (o, g):
            build_jointed(w, seed)
        b = g.bodies()
        g.w.set_ownership(1, 0, np.where(b["invMass"] > 0, 0, -1).astype(np.int32))
        g.w.comm_init(1, 0, pkg.comm_unique_id())
        for s in range(10):
            o.step(); g.step()
            H.assert_same_world(o, g, f"seed {seed} step {s + 1}")
        assert int(g.w.stats()["schedule"]) == 3


def build_contacts_only(w, seed):
    """random boxes, contacts only (what the group schedule runs), dense enough that bodies get many partners"""
    r = np.random.default_rng(seed)
    n = int(r.integers(40, 400))
    w.add_body((float(r.uniform(10, 30)), 1.0, float(r.uniform(10, 30))), 0.0, float(r.uniform(0.1, 1.0)), (0, -0.5, 0))
    span = float(r.uniform(1.5, 4.0))
    for k in range(n):
        size = tuple(float(x) for x in r.uniform(0.4, 1.5, 3))
        pos = (float(r.uniform(-span, span)), float(r.uniform(0.4, 6)), float(r.uniform(-span, span)))
        q = r.normal(size=4); q = tuple(float(x) for x in q / np.linalg.norm(q))
        w.add_body(size, float(r.uniform(0.3, 4.0)), float(r.uniform(0.0, 1.0)), pos, q, tuple(float(x) for x in r.uniform(-2, 2, 3)), tuple(float(x) for x in r.uniform(-3, 3, 3)))
    w.set_params(iterations=int(r.integers(1, 25)), alpha=float(r.uniform(0.5, 1.0)), beta=float(r.uniform(1e3, 2e5)), gamma=float(r.uniform(0.9, 1.0)),
                 postStabilize=int(r.integers(0, 2)))
    return n


@pytest.mark.parametrize("first", [300, 312, 324])
def test_random_piles_in_the_several_bodies_loops_of_the_group_schedule(first):
    """The loops of k_iterate_groups<true, 0> — pools (round 4), the task in hand, the plain loop, with and without arrival counters — on random
    piles: a handful of waves (`groups_waves`) makes a small scene a several-bodies-per-sub-group scene; wide boxes bring the 32- and 64-lane classes."""
    variants = [dict(groups_waves=8), dict(groups_waves=8, group_pool=0), dict(groups_waves=16, group_pool=0, group_dep=0), dict(groups_waves=4, group_pool_min_rounds=2),
                dict(groups_waves=12, group_spec=1, group_spec_rounds=16, group_pool=0), dict(groups_waves=8, groups_rounds=16)]
    ran4 = 0
    for seed in range(first, first + 12):
        opt = variants[seed % len(variants)]
        o, g = H.CpuWorld("orc"), H.HipWorld()
        for w in (o, g):
            build_contacts_only(w, seed)
        g.w.set_option("groups", 1)
        for k, v in opt.items():
            g.w.set_option(k, v)
        for s in range(14):
            o.step(); g.step()
            H.assert_same_world(o, g, f"seed {seed} step {s + 1} with {opt}")
            ran4 += int(g.w.stats()["schedule"]) == 4
    assert ran4 >= 100, ran4          # (a few waves decline the densest piles — more than 16 bodies per sub-group: those steps take the resident schedule)
