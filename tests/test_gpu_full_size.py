"""BASELINE.json configs 3-5 at their full sizes on ONE MI355X, exact mode: 8 000 cubes, 64 000 cubes, 64 000 cubes + 10 000
weld joints.  The CPU oracle needs ~0.5 s/step at 8 k and ~20 s/step at 64 k (O(n^2) broadphase), so the direct bit-for-bit
comparisons run a handful of steps of the contact-rich packed variant; size-independent properties (determinism across
runs, nothing non-finite, nothing through the ground) cover longer runs."""
import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu


def add_config5_joints(w, count=10000):
    """Joint(cube[2k], cube[2k+1], (0.5,0,0), (-0.5,0,0)) for the first `count` x-adjacent pairs
    (body 0 is the ground, cubes follow in creation order, x is the innermost loop of the scene builder)."""
    for k in range(count):
        w.add_joint(1 + 2 * k, 2 + 2 * k, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))


def test_config3_8000_cubes_bit_exact_vs_oracle():
    o, g = H.CpuWorld("orc"), H.HipWorld()
    for w in (o, g):
        w.scene("Packed:20")
    assert g.w.num_bodies() == 8001
    for s in range(12):
        o.step(); g.step()
        H.assert_same_world_fast(o, g, f"Packed:20 step {s + 1}")
    st = g.w.stats()
    assert int(st["numManifolds"]) > 5000 and int(st["schedule"]) == 4     # group schedule: every body resident in a sub-group of a wavefront


def test_config3_8000_cubes_resident_schedule_with_parked_bodies_bit_exact_vs_oracle(monkeypatch):
    """The same scene through the schedule that scenes beyond 8 192 bodies (and scenes with joints) take: several bodies per
    wavefront, state parked between tasks."""
    monkeypatch.setenv("AVBD_GROUPS", "0")
    o, g = H.CpuWorld("orc"), H.HipWorld()
    for w in (o, g):
        w.scene("Packed:20")
    for s in range(6):
        o.step(); g.step()
        H.assert_same_world_fast(o, g, f"Packed:20 step {s + 1} (parked sweep)")
    assert int(g.w.stats()["schedule"]) == 3


def test_config2_falling_8000_cubes_bit_exact_vs_oracle_on_landing():
    """configs[2] itself (StressN:20, the falling scene, not the packed variant): the reference cannot load a snapshot and its
    O(n^2) broadphase makes 130 steps of free fall expensive on the CPU, so the restatement is started from the device's own
    state shortly before the first contact and both go through the landing of the first layers, bit for bit."""
    g = H.HipWorld(); g.scene("StressN:20"); g.step(131)                 # first contacts appear in step 132
    o = H.CpuWorld("orc"); o.scene("StressN:20")
    o.set_bodies(g.bodies()); o.set_forces(g.forces()); o.set_step_index(g.step_index())
    for s in range(8):
        o.step(); g.step()
        H.assert_same_world_fast(o, g, f"StressN:20 step {132 + s}")
    assert int(g.diag()["activeContacts"]) > 1000


def test_config3_8000_cubes_long_run_is_deterministic_and_finite():
    runs = []
    for _ in range(2):
        g = H.HipWorld(); g.scene("Packed:20"); g.step(120)
        runs.append(g.bodies())
    assert H.same_bits(runs[0], runs[1])
    b = runs[0]
    assert np.isfinite(b["pos"]).all() and np.isfinite(b["quat"]).all() and np.isfinite(b["linVel"]).all()
    assert float(b["pos"][1:, 1].min()) > 0.3          # nothing fell through the ground slab


def test_config5_64000_cubes_with_10k_weld_joints_bit_exact_vs_oracle():
    o, g = H.CpuWorld("orc"), H.HipWorld()
    for w in (o, g):
        w.scene("Packed:40")
        add_config5_joints(w)
    assert g.w.num_bodies() == 64001
    for s in range(2):
        o.step(); g.step()
        H.assert_same_world_fast(o, g, f"Packed:40 + 10k joints, step {s + 1}")
    st = g.w.stats()
    assert int(st["numJoints"]) == 10000 and int(st["numManifolds"]) > 40000


def test_config4_64000_cubes_with_joints_is_deterministic_and_finite():
    """40 steps of the falling 64 000-cube scene with the 10 000 welds: two runs agree bit for bit and nothing is non-finite
    (the reference's weld rows are bug-compatible — SURVEY.md §7-5 — so no claim is made about their residuals)."""
    finals = []
    for _ in range(2):
        g = H.HipWorld(); g.scene("StressN:40"); add_config5_joints(g)
        g.step(40)
        finals.append(g.bodies())
    assert H.same_bits(finals[0], finals[1])
    b = finals[0]
    assert np.isfinite(b["pos"]).all() and np.isfinite(b["quat"]).all() and np.isfinite(b["linVel"]).all()


def test_grid_broadphase_matches_the_n2_loop_on_mixed_sizes(monkeypatch):
    """Above 4096 bodies the broadphase is a hashed grid.  Mixed radii put several bodies on its "big" list
    (ground slab, two long static beams that overlap each other, one large dynamic crate); AVBD_BROADPHASE=n2 is the
    O(n^2) tile kernel.  Both must reproduce the oracle's (O(n^2)) manifold set and state bit for bit."""
    rng = np.random.default_rng(7)
    n = 4600
    size = rng.uniform(0.5, 1.4, (n, 3)).astype(np.float32)
    pos = np.stack([rng.uniform(-30, 30, n), rng.uniform(0.8, 9.0, n), rng.uniform(-30, 30, n)], 1).astype(np.float32)

    def build(w):
        w.add_body((100, 1, 100), 0.0, 0.5, (0, -0.5, 0))
        w.add_body((40, 2, 2), 0.0, 0.6, (0, 1.0, 10))                 # static beam
        w.add_body((2, 2, 40), 0.0, 0.6, (5, 1.5, 0))                  # static beam crossing the first one
        w.add_body((8, 8, 8), 0.3, 0.5, (-12, 4.2, -12))               # big dynamic crate
        for k in range(n):
            w.add_body(tuple(float(x) for x in size[k]), 1.0, 0.5, tuple(float(x) for x in pos[k]))
        w.set_params(iterations=6)

    o = H.CpuWorld("orc"); build(o); o.step(12)
    for mode in ("grid", "n2"):
        monkeypatch.setenv("AVBD_BROADPHASE", mode)
        g = H.HipWorld(); build(g); g.step(12)
        monkeypatch.delenv("AVBD_BROADPHASE")
        H.assert_same_world_fast(o, g, f"mixed sizes, broadphase={mode}")
    assert int(g.w.stats()["numManifolds"]) > 800 and int(g.w.stats()["numCandidates"]) > 0


def test_22000_cubes_several_bodies_per_sub_group_match_the_parked_sweep(monkeypatch):
    """21 952 falling cubes, steps 0..230 (the pile lands from step ~130): the group schedule with several bodies per sub-group
    (three rounds, plan scanned over six tiles of 4096 bodies by six workgroups, out-of-order picks) and the resident schedule's
    parked sweep are both exact, so they must agree bit for bit — bodies and forces — deep into the landing."""
    finals = {}
    for groups in ("default", "0"):
        if groups == "0":
            monkeypatch.setenv("AVBD_GROUPS", "0")
        g = H.HipWorld(); g.scene("StressN:28")
        g.step(230); g.w.synchronize()
        st = g.w.stats()
        assert int(st["schedule"]) == (4 if groups == "default" else 3), st
        assert int(st["numManifolds"]) > 20000
        finals[groups] = (g.bodies(), g.forces())
        if groups == "0":
            monkeypatch.delenv("AVBD_GROUPS")
    assert finals["default"][0].tobytes() == finals["0"][0].tobytes(), "body state differs between the two schedules"
    assert finals["default"][1].tobytes() == finals["0"][1].tobytes(), "force state differs between the two schedules"


def test_64000_cubes_landed_pile_group_schedule_matches_the_parked_sweep_and_the_oracle(monkeypatch):
    """The workload the strong-scaling line of bench.py times (StressN:40 from step 470: eight bodies per sub-group, out-of-order
    picks, header table in LDS), at its own size: 50 steps of the default group schedule and of the resident schedule's parked sweep
    (AVBD_GROUPS=0) from the same device state must agree bit for bit, bodies and forces; and ONE step of the CPU restatement from
    that state (~20 s: its broadphase is O(n^2)) must agree with the device's first step of the window."""
    g0 = H.HipWorld(); g0.scene("StressN:40"); g0.step(470); g0.w.synchronize()
    assert int(g0.w.stats()["schedule"]) == 4 and int(g0.w.stats()["numManifolds"]) > 120000
    b0, f0, s0 = g0.bodies(), g0.forces(), g0.step_index()
    o = H.CpuWorld("orc"); o.scene("StressN:40")
    o.set_bodies(b0); o.set_forces(f0); o.set_step_index(s0)
    o.step()
    g0.step()
    H.assert_same_world_fast(o, g0, "StressN:40 step 471 (group schedule vs oracle)")
    o.close()
    finals = {}
    for groups in ("default", "0"):
        if groups == "0":
            monkeypatch.setenv("AVBD_GROUPS", "0")
        g = H.HipWorld(); g.scene("StressN:40")
        g.set_bodies(b0); g.set_forces(f0); g.set_step_index(s0)
        g.step(50); g.w.synchronize()
        assert int(g.w.stats()["schedule"]) == (4 if groups == "default" else 3)
        finals[groups] = (g.bodies(), g.forces())
        g.close()
        if groups == "0":
            monkeypatch.delenv("AVBD_GROUPS")
    assert finals["default"][0].tobytes() == finals["0"][0].tobytes(), "body state differs between the two schedules after 50 steps"
    assert finals["default"][1].tobytes() == finals["0"][1].tobytes(), "force state differs between the two schedules after 50 steps"
