"""The fast sweep (avbd_set_option(ctx, "sweep", 1), ABI 5; avbd-demo3d_amd/csrc/avbd_fast.hip) against the exact mode and the reference's golden data.

The exact mode replays the reference's Gauss-Seidel sweep (solver.cpp:344-430) bit for bit and is pinned to the reference by the other test files; the fast mode
changes the ORDER of bodies (multi-colour Gauss-Seidel) and of sums (DPP trees), so it is held to the tolerance DESIGN.md section 2b states — the bounds below,
each with the value tools/fast_tolerance.py measured on an MI355X next to it:
 (i)   one teacher-forced step from the reference's snapshots / from the exact mode's own state: max |dpos| in metres, max rotation angle in radians;
 (ii)  invariants of whole runs against the reference's diagnostics (contact / manifold counts in a band, first contact on the same step, Stack standing);
 (iii) bit-identical results from run to run and between the forms of the kernel (stamped records / grid barrier, latency / throughput form, any grid size).
What the fast sweep does not take (the level mode, a body with more than 64 contacts) must run the exact schedules: bit-exact against the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rot_angle(qa, qb):
    d = np.abs(np.sum(qa.astype(np.float64) * qb.astype(np.float64), axis=1))
    return 2.0 * np.arccos(np.clip(d, 0.0, 1.0))


def body_diff(ba, bb):
    dp = np.linalg.norm(ba["pos"].astype(np.float64) - bb["pos"].astype(np.float64), axis=1)
    return float(dp.max()), float(rot_angle(ba["quat"], bb["quat"]).max())


def fast_world(scene, **opts):
    g = H.HipWorld(); g.scene(scene); g.w.set_option("sweep", 1)
    for k, v in opts.items():
        g.w.set_option(k, v)
    return g


def same_state(a, b):
    fa, fb = a.forces(), b.forces()
    return H.same_bits(a.bodies(), b.bodies()) and len(fa) == len(fb) and H.same_bits(fa["lambda"], fb["lambda"]) and H.same_bits(fa["penalty"], fb["penalty"])


# ---- (i) teacher-forced steps -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,scene,max_dpos,max_drot", [
    ("snapshot_Packed4_step25.npz", "Packed:4", 2e-4, 2e-3),          # measured 1.4e-5 m, 0 rad
    ("snapshot_Pyramid_step60.npz", "Pyramid", 5e-4, 2e-3),           # measured 1.2e-4 m, 0 rad
    ("snapshot_TwoBlockDrop_step120.npz", "TwoBlockDrop", 1e-6, 1e-5),  # measured 1.4e-11 m (one colour per body pair: the same sweep)
])
def test_fast_step_from_reference_snapshot(name, scene, max_dpos, max_drot):
    z = np.load(os.path.join(G, name), allow_pickle=False)
    g = fast_world(scene)
    g.set_bodies(z["bodies_before"].view(H.BODY_DT)); g.set_forces(z["forces_before"].view(H.FORCE_DT))
    g.step()
    assert int(g.w.stats()["schedule"]) == 5
    dp, dr = body_diff(z["bodies_after"].view(H.BODY_DT), g.bodies())
    assert dp <= max_dpos and dr <= max_drot, (scene, dp, dr)
    fa, fb = z["forces_after"].view(H.FORCE_DT), g.forces()
    # the same force list (pairs, contact counts, order): the front end is the exact mode's
    assert len(fa) == len(fb) and np.array_equal(fa["bodyA"], fb["bodyA"]) and np.array_equal(fa["bodyB"], fb["bodyB"]) and np.array_equal(fa["numContacts"], fb["numContacts"])
    # warm-started rows stay close: |dlambda| below 10 % of the largest multiplier of the scene
    assert np.abs(fa["lambda"].astype(np.float64) - fb["lambda"].astype(np.float64)).max() <= 0.1 * max(1.0, float(np.abs(fa["lambda"]).max()))


def _fork(scene, at, snapshot=None):
    ex = H.HipWorld(); ex.scene(scene)
    if snapshot is not None:
        z = np.load(os.path.join(G, snapshot), allow_pickle=False)
        b, f = z["bodies"].view(H.BODY_DT), z["forces"].view(H.FORCE_DT)
    else:
        ex.step(at)
        b, f = ex.bodies(), ex.forces()
    fa = fast_world(scene); fa.set_bodies(b); fa.set_forces(f); fa.set_step_index(at)
    e2 = H.HipWorld(); e2.scene(scene); e2.set_bodies(b); e2.set_forces(f); e2.set_step_index(at)
    ex.close()
    return e2, fa


def test_fast_vs_exact_on_the_settled_stress1000_pile():
    """BASELINE configs[1] in its steady window: one step 1e-3 m / 5e-3 rad (measured 2.0e-4 / 1.0e-3), ten steps 5e-2 m / 7e-2 rad (measured 1.1e-2 / 1.4e-2)"""
    e, f = _fork("Stress1000", 400)
    e.step(); f.step()
    dp, dr = body_diff(e.bodies(), f.bodies())
    assert dp <= 1e-3 and dr <= 5e-3, (dp, dr)
    e.step(9); f.step(9)
    dp, dr = body_diff(e.bodies(), f.bodies())
    assert dp <= 5e-2 and dr <= 7e-2, (dp, dr)
    assert int(f.w.stats()["schedule"]) == 5 and f.w.fast_plan_info()["colors"] <= 8


def test_fast_vs_exact_in_the_landing_of_stress1000():
    """the reference's snapshot at step 140, eight steps after first contact: cubes hit at 16 m/s, twenty Gauss-Seidel iterations are far from converged and the
    order of the sweep shows — the violent-phase bound: one step 0.1 m / 0.3 rad (measured 3.2e-2 / 9.7e-2; the same state moves by 0.2 m in 29 steps when the
    reference itself is merely compiled with FMA contraction)"""
    e, f = _fork("Stress1000", 140, "stress1000_snapshot_step140.npz")
    e.step(); f.step()
    dp, dr = body_diff(e.bodies(), f.bodies())
    assert dp <= 0.1 and dr <= 0.3, (dp, dr)


def test_fast_post_stabilize_step():
    """postStabilize: iterations + 1 sweeps, alpha 1 then 0, no dual behind the last one (solver.cpp:340-342, 411)"""
    ex = H.HipWorld(); ex.scene("Pyramid"); ex.set_params(postStabilize=1); ex.step(60)
    b, f = ex.bodies(), ex.forces()
    fa = fast_world("Pyramid"); fa.set_params(postStabilize=1); fa.set_bodies(b); fa.set_forces(f); fa.set_step_index(60)
    ex.step(); fa.step()
    dp, dr = body_diff(ex.bodies(), fa.bodies())
    assert dp <= 1e-3 and dr <= 5e-3, (dp, dr)
    assert int(fa.w.stats()["schedule"]) == 5


# ---- (iii) forms of the kernel: the same bits ------------------------------------------------------------------------------------
@pytest.mark.parametrize("scene,steps,params", [("Stress1000", 170, None), ("Packed:6", 25, dict(iterations=8)), ("Pyramid", 40, None)])
def test_every_form_of_the_fast_sweep_gives_the_same_bits(scene, steps, params):
    """stamped records vs a grid barrier per colour, latency vs throughput forms of k_fast_sweep, the latency form on one XCD (plain-store hand-offs through its L2; the default
    where the tiles with neighbours fit) vs on all of them, few workgroups (several tiles per wave): only WHEN a body is visited changes, never what it reads — through first
    contact and the landing (colours kept, lost and re-picked every step; the first steps have no colour at all: every body is in the isolated row)"""
    ref = None
    for opts in ({}, {"fast_one_xcd": 0}, {"debug_fast_xcd_mismatch": 1}, {"fast_stamps": 0}, {"fast_form": 2}, {"fast_form": 3}, {"fast_form": 1, "fast_blocks": 3}, {"fast_form": 2, "fast_blocks": 2, "fast_stamps": 0}):
        g = fast_world(scene, **opts)
        if params:
            g.set_params(**params)
        g.step(steps)
        assert int(g.w.stats()["schedule"]) == 5
        if ref is None:
            ref = g
        else:
            assert same_state(ref, g), (scene, opts)
            g.close()


def test_run_to_run_bit_identical():
    a = fast_world("Stress1000"); a.step(200)
    b = fast_world("Stress1000"); b.step(200)
    assert same_state(a, b)


@pytest.mark.parametrize("recolor", [1, 3, 1 << 20])
def test_the_colouring_is_proper(recolor):
    """no two bodies that share a force share a colour; every swept body has one — with colours kept from step to step and with a fresh colouring every step"""
    g = fast_world("Stress1000", fast_recolor=recolor)
    for steps in (135, 20, 60):
        g.step(steps)
        col = g.w.fast_colors()
        b, f = g.bodies(), g.forces()
        dyn = b["invMass"] > 0
        assert (col[dyn] >= 0).all() and (col[~dyn] == -2).all()
        a, bb = f["bodyA"], f["bodyB"]
        both = dyn[a] & dyn[bb]
        assert both.sum() > 0 and (col[a[both]] != col[bb[both]]).all()
        info = g.w.fast_plan_info()
        assert info["bad"] == 0 and info["colors"] == col.max() + 1 and info["swept"] == int(dyn.sum())


# ---- (ii) whole runs ---------------------------------------------------------------------------------------------------------------
def test_stress1000_600_steps_invariants_against_the_reference_diagnostics():
    """A chaotic pile: two equally valid orders of the sweep (another tie-breaking seed of the colouring, tools/fast_angspeed.py: eleven of them) end in piles that differ cube by
    cube, so the bands are those of that spread, not of one run.  First contact on the reference's step; from step 300 on contacts within [0.88, 1.12] and manifolds within
    [0.80, 1.08] of the reference's counts (measured over the eleven orders: [0.92, 1.07] and [0.86, 1.03]); the pile comes to rest like the reference's: the fastest body of the
    last 100 steps (one that left the pile at the landing: free fall) within 15 %, the landing's 50 rad/s gone (max angular speed of the last 100 steps <= 15 rad/s; measured
    2.1 .. 6.9, reference 2.5 — the tail of ONE late-toppling cube), and at step 600 the 99th percentile of |angVel| <= 0.5 rad/s with at most 5 bodies above 1 rad/s (measured
    <= 0.15 and <= 1; the exact mode: 0.05 and 0)"""
    diags = np.load(os.path.join(G, "stress1000_diag_600.npy")).view(H.DIAG_DT)
    g = fast_world("Stress1000")
    rec = []
    for s in range(600):
        g.step()
        d = g.diag()
        rec.append((int(d["activeContacts"]), int(d["activeManifolds"]), float(d["maxLinearSpeed"]), float(d["maxAngularSpeed"])))
    rec = np.array(rec)
    assert int(np.nonzero(rec[:, 0] > 0)[0][0]) == int(np.nonzero(diags["activeContacts"] > 0)[0][0]) == 131
    rc = rec[300:, 0] / diags["activeContacts"][300:]; rm = rec[300:, 1] / diags["activeManifolds"][300:]
    assert 0.88 <= rc.min() and rc.max() <= 1.12, (rc.min(), rc.max())
    assert 0.80 <= rm.min() and rm.max() <= 1.08, (rm.min(), rm.max())
    assert abs(rec[500:, 2].max() / diags["maxLinearSpeed"][500:].max() - 1.0) <= 0.15
    assert rec[500:, 3].max() <= 15.0, rec[500:, 3].max()
    b = g.bodies()
    dyn = b["invMass"] > 0
    w = np.linalg.norm(b["angVel"][dyn].astype(np.float64), axis=1)
    assert np.percentile(w, 99) <= 0.5 and int((w > 1.0).sum()) <= 5, (np.percentile(w, 99), int((w > 1.0).sum()))
    assert np.isfinite(b["pos"]).all() and np.isfinite(b["quat"]).all()
    assert int(g.w.stats()["schedule"]) == 5


def test_stack_stands_and_pyramid_settles():
    """Stack after 300 steps: still a stack (|x|, |z| <= 1e-3: measured 0), heights within 1e-2 m of the exact mode's (measured 3.5e-3); Pyramid at rest like the exact mode's"""
    g = fast_world("Stack"); g.step(300)
    e = H.HipWorld(); e.scene("Stack"); e.step(300)
    b, be = g.bodies(), e.bodies()
    dyn = b["invMass"] > 0
    assert np.abs(b["pos"][dyn][:, [0, 2]]).max() <= 1e-3 and np.abs(b["pos"][dyn][:, 1] - be["pos"][dyn][:, 1]).max() <= 1e-2
    assert int(g.diag()["activeContacts"]) == int(e.diag()["activeContacts"]) == 40
    g = fast_world("Pyramid"); g.step(300)
    e = H.HipWorld(); e.scene("Pyramid"); e.step(300)
    b, be = g.bodies(), e.bodies()
    dyn = b["invMass"] > 0
    assert np.abs(b["pos"][dyn][:, 1] - be["pos"][dyn][:, 1]).max() <= 5e-3                     # measured 5.2e-4
    assert np.linalg.norm(b["linVel"][dyn], axis=1).max() <= 0.1                                # at rest (the exact mode jitters at 0.02-0.03 m/s too)
    assert int(e.diag()["activeContacts"]) == 400 and abs(int(g.diag()["activeContacts"]) - 400) <= 8   # (another order of the sweep may settle with one manifold more or less)


# ---- what the fast sweep declines ----------------------------------------------------------------------------------------------------
def test_the_level_mode_ignores_the_sweep_option_and_stays_bit_exact():
    """mode 0 (one launch per Gauss-Seidel level, the first correct path) is not a data-flow schedule: with sweep = 1 it still runs the exact sweep, bit for bit"""
    o = H.CpuWorld("orc"); g = H.HipWorld()
    for w in (o, g):
        H.build_row_api_scene(w, "@jointed")
    g.w.set_option("mode", 0); g.w.set_option("sweep", 1)
    for s in range(3):
        o.step(5); g.step(5)
        H.assert_same_world(o, g, f"@jointed, level mode with sweep = 1, step {5 * s + 5}")
    assert int(g.w.stats()["schedule"]) == 0


def test_fast_sweep_springs_match_the_exact_mode_to_rounding():
    """bodies hung from a static beam by springs (soft: finite stiffness, no lambda; hard: stiffness FLT_MAX, lambda and a dual update) and a spring between a body and a
    static post: one swept endpoint per spring, so the fast sweep runs the exact mode's sequence per body — 30 steps within 2e-4 m / 2e-3 rad, rows within 0.5 %"""
    def build(w):
        w.scene("Empty")
        post = w.add_body((0.5, 4.0, 0.5), 0.0, 0.5, (-3.0, 2.0, 0.0))
        beam = w.add_body((12.0, 0.4, 0.5), 0.0, 0.5, (3.0, 6.0, 0.0))                          # a static beam the bodies hang from
        ids = [w.add_body((0.8, 0.6 + 0.1 * k, 0.7), 1.0 + 0.5 * k, 0.5, (2.0 * k, 3.0, 0.0), quat=(0.0, 0.1 * k, 0.0, float(np.sqrt(1.0 - 0.01 * k * k))), lin=(0.1, 0.0, 0.1 * k)) for k in range(4)]
        w.add_spring(beam, ids[0], (-3.0, -0.2, 0.0), (0.0, 0.3, 0.0), 800.0, 1.5)
        w.add_spring(beam, ids[1], (-0.8, -0.2, 0.1), (0.2, 0.3, 0.0), 3000.0)
        w.add_spring(beam, ids[2], (1.0, -0.2, 0.0), (0.0, 0.35, 0.0), H.FLT_MAX, 1.8)          # a hard row: lambda + penalty ramp
        w.add_spring(post, ids[3], (0.0, 1.5, 0.0), (0.0, 0.3, 0.0), 500.0, 8.5)
        w.add_spring(beam, ids[3], (3.0, -0.2, 0.0), (0.0, 0.3, 0.2), 1500.0)                   # two springs on one body
    e = H.HipWorld(); build(e)
    f = H.HipWorld(); build(f); f.w.set_option("sweep", 1)
    for s in range(6):
        e.step(5); f.step(5)
        assert int(f.w.stats()["schedule"]) == 5, int(f.w.stats()["schedule"])
        dp, dr = body_diff(e.bodies(), f.bodies())
        assert dp <= 2e-4 and dr <= 2e-3, (s, dp, dr)
        fe, ff = e.forces(), f.forces()
        assert np.array_equal(fe["kind"], ff["kind"]) and (fe["kind"] == 2).sum() == 5
        for name in ("lambda", "penalty", "C"):
            x, y = fe[name][:, :1].astype(np.float64), ff[name][:, :1].astype(np.float64)
            assert np.abs(x - y).max() <= 5e-3 * max(1.0, np.abs(x).max()), (s, name, np.abs(x - y).max(), np.abs(x).max())


def test_a_body_with_more_than_64_contacts_is_declined_on_the_device_and_runs_exactly():
    """a dynamic plate on a 5 x 5 grid of cubes carries 100 contacts: the plan vetoes the step (FastPlan.bad bit 0), the task schedule queued behind the sweep runs it"""
    def build(w):
        w.add_body((40, 1, 40), 0.0, 0.6, (0, -0.5, 0))
        for i in range(5):
            for k in range(5):
                w.add_body((1, 1, 1), 1.0, 0.5, (1.5 * i - 3.0, 0.5, 1.5 * k - 3.0))
        w.add_body((8, 0.5, 8), 0.5, 0.5, (0.0, 1.26, 0.0))
    o = H.CpuWorld("orc"); g = H.HipWorld()
    build(o); build(g)
    o.step(30); g.step(30)                                  # exact mode: the plate has settled on its 25 cubes
    H.assert_same_world(o, g, "plate on 25 cubes, step 30 (exact mode)")
    g.w.set_option("sweep", 1)
    for s in range(5):                                      # every one of these steps is declined (100 contacts on the plate) and must be the oracle's step
        o.step(); g.step()
        assert int(g.w.stats()["schedule"]) != 5 and (g.w.fast_plan_info()["bad"] & 1) == 1
        H.assert_same_world(o, g, f"plate on 25 cubes, declined step {31 + s}")
    f = g.forces()
    per_body = np.bincount(np.concatenate([np.repeat(f["bodyA"], f["numContacts"]), np.repeat(f["bodyB"], f["numContacts"])]))
    assert int(per_body[-1]) > 64                           # the plate (created last)


def test_cli_mode_fast():
    exe = os.path.join(H.ROOT, "avbd-demo3d_amd", "avbd_demo3d")
    out = subprocess.run([exe, "--nogfx", "--scene", "Pyramid", "--steps", "60", "--mode", "fast", "--batch"], capture_output=True, check=True, text=True).stdout
    line = json.loads(out.strip().splitlines()[-1])
    assert line["mode"] == "fast" and line["schedule"] == 5
    out = subprocess.run([exe, "--nogfx", "--scene", "Pyramid", "--steps", "60", "--batch"], capture_output=True, check=True, text=True).stdout
    line = json.loads(out.strip().splitlines()[-1])
    assert line["mode"] == "exact" and line["schedule"] in (1, 2, 3, 4)


# ---- BASELINE configs[2] / [3] sizes ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scene,at,band", [("StressN:20", 410, 0.05), ("StressN:40", 470, 0.05)])
def test_fast_sweep_at_full_size(scene, at, band):
    """8 000 / 64 000 cubes, forked from the exact mode's landed pile: the throughput form with many tiles per wave; stamped and barrier form agree bit for bit
    (two-level grid barrier: more than 64 workgroups); after 10 steps the contact and manifold counts stay within 5 % of the exact mode's"""
    e, f = _fork(scene, at)
    b0, f0 = e.bodies(), e.forces()
    f2 = fast_world(scene, fast_stamps=0); f2.set_bodies(b0); f2.set_forces(f0); f2.set_step_index(at)
    e.step(10); f.step(10); f2.step(10)
    assert int(f.w.stats()["schedule"]) == 5 and f.w.fast_plan_info()["bad"] == 0
    assert same_state(f, f2)
    de, df = e.diag(), f.diag()
    assert abs(int(df["activeContacts"]) / int(de["activeContacts"]) - 1.0) <= band and abs(int(df["activeManifolds"]) / int(de["activeManifolds"]) - 1.0) <= band
    b = f.bodies()
    assert np.isfinite(b["pos"]).all() and np.isfinite(b["quat"]).all()


# ---- sharded: the all-gather form on top of the fast sweep (BASELINE configs[3]) -------------------------------------------
def _ranks(scene, n, **opts):
    pkg = H.load_product()
    worlds = [H.HipWorld() for _ in range(n)]
    for w in worlds:
        w.scene(scene); w.w.set_option("sweep", 1)
        for k, v in opts.items():
            w.w.set_option(k, v)
    b = worlds[0].bodies()
    owner = pkg.partition_by_axis(b["pos"], b["invMass"], n, axis=0)
    for r, w in enumerate(worlds):
        w.w.set_ownership(n, r, owner)
    return pkg, worlds


def test_one_rank_per_iteration_launches_equal_the_single_launch_bit_for_bit():
    """force_sharded: one rank that owns every body runs the sharded schedule — one launch of k_fast_sweep per iteration, pack / exchange / unpack between them —
    and must reproduce the one-launch sweep exactly: the stamps of the pose and contact records carry over from launch to launch"""
    pkg = H.load_product()
    a = fast_world("Stress1000"); a.step(180)
    b = fast_world("Stress1000", force_sharded=1)
    b.w.set_ownership(1, 0, np.where(b.bodies()["invMass"] > 0, 0, -1).astype(np.int32))
    pkg.group_step([b.w], 180)
    assert int(b.w.stats()["schedule"]) == 5
    assert same_state(a, b)


@pytest.mark.parametrize("n", [2, 4])
def test_sharded_fast_sweep_stays_in_the_band_of_the_one_gpu_run(n):
    """n x-slabs of Stress1000 on one device (avbd_group_step: device copies where one process per GPU runs the RCCL all-gather), Jacobi across the cuts:
    every rank holds the same poses after every step (they all gather the same rows), the pile forms like the one-GPU fast run's — the union of the ranks'
    force lists has the one-GPU run's contacts and manifolds within 20 % over steps 250..350, while the pile is still landing (two one-GPU runs that differ in the
    tie-breaking seed of their colouring are 8 % apart from step 300 on, tools/fast_angspeed.py; measured here up to 13 %) — and every rank ran the fast sweep"""
    pkg, worlds = _ranks("Stress1000", n)
    one = fast_world("Stress1000")
    pkg.group_step([w.w for w in worlds], 250); one.step(250)
    for s in range(5):
        pkg.group_step([w.w for w in worlds], 20); one.step(20)
        b0 = worlds[0].bodies()
        for w in worlds[1:]:
            assert H.same_bits(b0["pos"], w.bodies()["pos"]) and H.same_bits(b0["quat"], w.bodies()["quat"])
        assert np.isfinite(b0["pos"]).all()
        # a force across a cut is held by both ranks that sweep one of its bodies: the union over the ranks, every pair once, against the one-GPU run's list
        union = {}
        for w in worlds:
            f = w.forces()
            for a, bb, nc in zip(f["bodyA"], f["bodyB"], f["numContacts"]):
                union[(int(a), int(bb))] = max(union.get((int(a), int(bb)), 0), int(nc))
        d1 = one.diag()
        tot_m, tot_c = len(union), sum(union.values())
        assert abs(tot_c / int(d1["activeContacts"]) - 1.0) <= 0.20, (tot_c, int(d1["activeContacts"]))
        assert abs(tot_m / int(d1["activeManifolds"]) - 1.0) <= 0.20, (tot_m, int(d1["activeManifolds"]))
    for w in worlds:
        st = w.w.stats()
        assert int(st["schedule"]) == 5 and int(st["cutMode"]) == 1


def test_rccl_single_rank_fast_sweep_is_the_single_launch_bit_for_bit():
    """the one-process-per-GPU path (RCCL communicator, ncclAllGather of 32 B per body between the per-iteration launches of k_fast_sweep) with one rank:
    the same bits as the one-launch sweep"""
    pkg = H.load_product()
    g = fast_world("Pyramid", force_sharded=1)
    b = g.bodies()
    g.w.set_ownership(1, 0, np.where(b["invMass"] > 0, 0, -1).astype(np.int32))
    g.w.comm_init(1, 0, pkg.comm_unique_id())
    a = fast_world("Pyramid")
    for s in range(3):
        g.step(15); a.step(15)
        assert same_state(a, g), f"after {15 * (s + 1)} steps"
    assert int(g.w.stats()["schedule"]) == 5


# ---- Joints as lane entries of the fast sweep (joint.cpp:68-139; BASELINE configs[4]) --------------------------------------------------------------------
def test_fast_sweep_world_anchored_joints_match_the_exact_mode_to_rounding():
    """bodies pinned to the world (hard, finite angular stiffness, finite linear stiffness; anchors violated by up to 0.5 m; both angular signs): every body is alone in
    its colour class chain — one swept endpoint per joint — so the fast sweep runs the exact mode's sequence per body and per force (computeConstraint, dual,
    computeConstraint) and only the order of the row sums differs: 20 steps stay within 2e-4 m / 2e-3 rad, the Joint rows within 0.5 % of their scale"""
    def build(w):
        w.scene("Empty")
        ids = []
        for k in range(6):
            b = w.add_body((1.0, 0.6 + 0.1 * k, 0.8), 1.0 + 0.3 * k, 0.5, (2.0 * k, 3.0, 0.0), quat=(0.0, 0.1 * k, 0.0, float(np.sqrt(1.0 - 0.01 * k * k))), lin=(0.1 * k, 0.0, 0.2))
            ids.append(b)
        w.add_joint(-1, ids[0], (0.0, 3.5, 0.0), (0.0, 0.3, 0.0))
        w.add_joint(-1, ids[1], (2.3, 3.2, 0.1), (0.2, 0.2, 0.0), ang_stiff=500.0)
        w.add_joint(-1, ids[2], (4.0, 3.4, -0.2), (0.0, 0.35, 0.1), lin_stiff=3000.0)
        w.add_joint(-1, ids[3], (6.1, 3.3, 0.0), (0.1, 0.3, 0.0), lin_stiff=5000.0, ang_stiff=50.0)
        w.add_joint(-1, ids[4], (8.0, 3.6, 0.3), (0.0, 0.3, 0.0))
        w.add_joint(-1, ids[4], (8.4, 3.0, 0.0), (0.4, 0.0, 0.0), ang_stiff=0.0)       # two joints on one body
    for sign in (0, 1):
        e = H.HipWorld(); build(e); e.w.set_joint_angular_sign(sign)
        f = H.HipWorld(); build(f); f.w.set_option("sweep", 1); f.w.set_joint_angular_sign(sign)
        for s in range(4):
            e.step(5); f.step(5)
            assert int(f.w.stats()["schedule"]) == 5, int(f.w.stats()["schedule"])
            dp, dr = body_diff(e.bodies(), f.bodies())
            assert dp <= 2e-4 and dr <= 2e-3, (sign, s, dp, dr)
            fe, ff = e.forces(), f.forces()
            assert np.array_equal(fe["kind"], ff["kind"]) and (fe["kind"] == 1).sum() == 6
            for name in ("lambda", "penalty", "C"):
                x, y = fe[name][:, :6].astype(np.float64), ff[name][:, :6].astype(np.float64)
                assert np.abs(x - y).max() <= 5e-3 * max(1.0, np.abs(x).max()), (sign, s, name, np.abs(x - y).max(), np.abs(x).max())


def test_fast_sweep_with_body_body_joints_one_step_against_the_exact_mode():
    """a body-body weld and a chain through a soft joint, corrected angular sign (with the reference's sign a body-body weld turns by 146 degrees in ONE step,
    SURVEY.md section 7-5, and any two orders of the sweep end metres apart), anchors satisfied at the start: the two endpoints of a joint are visited in colour
    order instead of creation order — one step within 5e-3 m / 2e-2 rad of the exact mode's, ten steps within 5e-2 m / 0.2 rad"""
    def build(w):
        w.scene("Ground")
        a = w.add_body((1, 1, 1), 1.0, 0.5, (0.0, 2.0, 0.0))
        b = w.add_body((1, 0.5, 1), 2.0, 0.4, (1.0, 2.0, 0.0))
        c = w.add_body((0.6, 0.6, 0.6), 1.0, 0.6, (2.0, 2.0, 0.0), lin=(0.2, 0.0, -0.1))
        w.add_joint(a, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))
        w.add_joint(b, c, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0), ang_stiff=300.0)
    e = H.HipWorld(); build(e); e.w.set_joint_angular_sign(1)
    f = H.HipWorld(); build(f); f.w.set_option("sweep", 1); f.w.set_joint_angular_sign(1)
    e.step(); f.step()
    assert int(f.w.stats()["schedule"]) == 5
    dp, dr = body_diff(e.bodies(), f.bodies())
    assert dp <= 5e-3 and dr <= 2e-2, (dp, dr)
    e.step(9); f.step(9)
    dp, dr = body_diff(e.bodies(), f.bodies())
    assert dp <= 5e-2 and dr <= 0.2, (dp, dr)


def test_fast_sweep_corrected_sign_welds_hold():
    """avbd_set_joint_angular_sign(1): welded pairs fall, land and stay rigid in the fast mode like in the exact one (the relative pose of the welded bodies stays
    within 2 cm / 0.05 rad of its initial value over 200 steps)"""
    def build(w):
        w.scene("Ground")
        ids = []
        for k in range(6):
            a = w.add_body((1, 1, 1), 1.0, 0.5, (2.5 * k - 6.0, 2.0 + 0.3 * k, 0.0))
            b = w.add_body((1, 1, 1), 1.0, 0.5, (2.5 * k - 5.0, 2.0 + 0.3 * k, 0.0))
            w.add_joint(a, b, (0.5, 0.0, 0.0), (-0.5, 0.0, 0.0))
            ids.append((a, b))
        return ids
    f = H.HipWorld(); ids = build(f); f.w.set_option("sweep", 1); f.w.set_joint_angular_sign(1)
    f.step(200)
    assert int(f.w.stats()["schedule"]) == 5
    b = f.bodies()
    for a, c in ids:
        d = np.linalg.norm(b["pos"][a].astype(np.float64) - b["pos"][c].astype(np.float64))
        assert abs(d - 1.0) <= 0.02, (a, c, d)
        assert rot_angle(b["quat"][a:a + 1], b["quat"][c:c + 1])[0] <= 0.05
    assert np.isfinite(b["pos"]).all() and b["pos"][:, 1].min() > -1.0


def test_fast_sweep_config5_64000_cubes_with_10k_welds_runs_the_fast_schedule():
    """BASELINE configs[4] on one GPU in the fast mode: 64 000 cubes + 10 000 weld Joints (bug-compatible sign: the welds blow up, SURVEY.md section 7-5, in every
    mode) — the step runs schedule 5, stamped and barrier forms agree bit for bit, the state stays finite"""
    scene = "StressN:40+welds:10000"
    a = fast_world(scene); b = fast_world(scene, fast_stamps=0)
    a.step(12); b.step(12)
    assert int(a.w.stats()["schedule"]) == 5 and int(a.w.stats()["numJoints"]) == 10000
    assert same_state(a, b)
    assert np.isfinite(a.bodies()["pos"]).all()
