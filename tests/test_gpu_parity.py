"""Parity of the HIP path (through the C ABI) against the CPU oracle — runs on the MI355X box.

Every comparison is BIT-EXACT (tolerance 0 ulp on every float of the body state, the force list
with lambda/penalty/contacts/stick flags, and the diagnostics): the simulated scenes are chaotic, so
any looser criterion is meaningless after a few dozen steps.
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu


def fresh_pair(scene):
    o = H.CpuWorld("orc")
    o.scene(scene)
    g = H.HipWorld()
    g.scene(scene)
    return o, g


def test_fp32_arithmetic_is_ieee_on_device():
    """/, sqrt correctly rounded, no FMA contraction, denormals preserved."""
    g = H.HipWorld()
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.normal(size=40000), rng.normal(size=5000) * 1e-38, rng.normal(size=5000) * 1e30, [1e-45, 3e-39, 1.0, 0.0]]).astype(np.float32)
    b = np.concatenate([rng.normal(size=40000), rng.normal(size=5000) * 1e-3, rng.normal(size=5000) * 1e-10, [2.0, 3.0, 3.0, 1.0]]).astype(np.float32)
    b[b == 0] = 1.0
    div, sq, muladd, rsq = g.w.kat_arith(a, b)
    with np.errstate(all="ignore"):
        assert H.same_bits(div, (a / b).astype(np.float32))
        assert H.same_bits(sq, np.sqrt(np.abs(a)).astype(np.float32))
        assert H.same_bits(muladd, ((a * b).astype(np.float32) + a).astype(np.float32))
        assert H.same_bits(rsq, (np.float32(1.0) / np.sqrt(np.abs(b)).astype(np.float32)).astype(np.float32))


def test_collide_kat_vs_oracle():
    g = H.HipWorld()
    rng = np.random.default_rng(5)
    n = 4000
    sA = rng.uniform(0.3, 1.5, (n, 3)).astype(np.float32)
    sB = rng.uniform(0.3, 1.5, (n, 3)).astype(np.float32)
    pA = rng.uniform(-0.2, 0.2, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pB = (pA + d * rng.uniform(0.2, 1.6, (n, 1))).astype(np.float32)
    qA = np.stack([H.rand_quat(rng) if k % 3 else np.array([0, 0, 0, 1], np.float32) for k in range(n)])
    qB = np.stack([H.rand_quat(rng) if k % 5 else np.array([0, 0, 0, 1], np.float32) for k in range(n)])
    out, cnt = g.w.kat_collide(pA, qA, sA, pB, qB, sB)
    hits = 0
    for k in range(n):
        ref = H.collide("orc", pA[k], qA[k], sA[k], pB[k], qB[k], sB[k])
        assert cnt[k] == len(ref), f"pair {k}: {cnt[k]} vs {len(ref)} contacts"
        got = out[k][:cnt[k]].view(H.CONTACT_DT)
        for name in ("feature", "rA", "rB", "normal", "penetration"):
            assert H.same_bits(got[name], ref[name]), f"pair {k} field {name}"
        hits += len(ref) > 0
    assert hits > 500


def test_solve6_kat_vs_oracle():
    g = H.HipWorld()
    rng = np.random.default_rng(7)
    n = 2000
    As, bs = [], []
    for k in range(n):
        J = rng.normal(size=(6, 9)).astype(np.float32)
        M = (J @ J.T + np.eye(6, dtype=np.float32) * np.float32(rng.uniform(0.0, 3.0))).astype(np.float32)
        if k % 50 == 0:
            M[0, 0] = 0.0
        As.append(np.concatenate([M[:3, :3].T.ravel(), M[:3, 3:].T.ravel(), M[3:, :3].T.ravel(), M[3:, 3:].T.ravel()]))
        bs.append(rng.normal(size=6))
    As, bs = np.array(As, np.float32), np.array(bs, np.float32)
    x = g.w.kat_solve6(As, bs)
    for k in range(n):
        assert H.same_bits(x[k], H.solve6("orc", As[k], bs[k])), f"system {k}"


@pytest.mark.parametrize("scene", ["Stack", "Pyramid", "Wall", "TwoBlockDrop", "Stress1000", "StressN:4", "Packed:3", "Rod (WIP)", "Soft Body (WIP)", "Ground", "Empty"])
def test_host_scene_builders_match_oracle(scene):
    o, g = fresh_pair(scene)
    assert H.same_bits(o.bodies(), g.bodies())
    po, pg = o.params(), g.params()
    for k in ("dt", "gravity", "iterations", "alpha", "beta", "gamma", "postStabilize"):
        assert H.same_bits(np.asarray(po[k]), np.asarray(pg[k])), k


@pytest.mark.parametrize("scene,steps,every", [("Stack", 150, 1), ("Pyramid", 80, 4), ("Wall", 80, 4), ("TwoBlockDrop", 220, 2),
                                               ("Soft Body (WIP)", 100, 5), ("Rod (WIP)", 10, 1), ("StressN:3", 260, 4),
                                               ("Packed:3", 80, 2), ("Packed:5", 40, 4)])
def test_scene_full_state_bit_exact(scene, steps, every):
    o, g = fresh_pair(scene)
    for s in range(steps):
        o.step(); g.step()
        if s % every == 0 or s == steps - 1:
            H.assert_same_world(o, g, f"{scene} after step {s + 1}")


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_piles_bit_exact(seed):
    o, g = H.CpuWorld("orc"), H.HipWorld()
    for w in (o, g):
        H.random_pile(w, np.random.default_rng(seed), 30)
        w.set_params(iterations=6 + seed)
    for s in range(100):
        o.step(); g.step()
        if s % 3 == 0:
            H.assert_same_world(o, g, f"pile seed {seed} step {s + 1}")
    assert o.diag()["activeContacts"] > 0


def test_post_stabilize_bit_exact():
    o, g = H.CpuWorld("orc"), H.HipWorld()
    for w in (o, g):
        H.random_pile(w, np.random.default_rng(11), 14)
        w.set_params(postStabilize=1, iterations=5, beta=5e4, gamma=0.98, alpha=0.9, gravity=(0.5, -9.0, 0.25))
    for s in range(60):
        o.step(); g.step()
        if s % 4 == 0:
            H.assert_same_world(o, g, f"postStabilize step {s + 1}")


@pytest.mark.parametrize("unfused", [0, 1], ids=["fused", "large-scene kernels"])
def test_joints_springs_ignore_bit_exact(unfused):
    """unfused: the kernels of scenes beyond 4096 bodies — k_narrowphase, k_scan / k_fill_adj / k_sort_expand (adjacency ranked in registers: two entries per
    joint, one per spring, none for an ignore marker) — instead of the fused front end and topology of small scenes"""
    o, g = H.CpuWorld("orc"), H.HipWorld()
    if unfused:
        g.w.set_option("fused_front", 0); g.w.set_option("fused_topology", 0)
    for w in (o, g):
        w.add_body((30, 1, 30), 0.0, 0.5, (0, -0.5, 0))
        i1 = w.add_body((1, 1, 1), 1.0, 0.5, (0, 3.0, 0))
        i2 = w.add_body((1, 1, 1), 1.0, 0.5, (1.0, 3.0, 0))
        i3 = w.add_body((1, 0.5, 1), 2.0, 0.4, (4.0, 2.0, 0), (0, 0, 0.1736482, 0.9848078))
        i4 = w.add_body((0.5, 0.5, 0.5), 1.0, 0.5, (4.0, 4.0, 0.3))
        i5 = w.add_body((1, 1, 1), 1.0, 0.5, (-4.0, 0.5, 0))
        i6 = w.add_body((1, 1, 1), 1.0, 0.5, (-4.0, 1.45, 0))
        w.add_joint(i1, i2, (0.5, 0, 0), (-0.5, 0, 0))
        w.add_joint(-1, i3, (4.0, 2.5, 0))
        w.add_spring(i3, i4, (0, 0.25, 0), (0, -0.25, 0), 500.0)
        w.add_spring(i1, i4, (0, 0, 0), (0, 0, 0), 200.0, 2.5)
        w.add_ignore(i5, i6)
    for s in range(12 + 48 * unfused):
        o.step(); g.step()
        H.assert_same_world(o, g, f"joints step {s + 1}")


def test_stress1000_checkpoints_bit_exact():
    """configs[1]: Stress1000 (1000 cubes, iterations=20) — full state at the SURVEY §8c checkpoints."""
    o, g = fresh_pair("Stress1000")
    done = 0
    for target in (1, 131, 132, 140, 170, 200, 260, 300):
        o.step(target - done); g.step(target - done); done = target
        H.assert_same_world(o, g, f"Stress1000 after step {target}")
    d = g.diag()
    assert d["activeManifolds"] > 100 and d["dynamicBodies"] == 1000


def test_teacher_forced_single_step_from_snapshot():
    """Upload an oracle snapshot (bodies + force list) into a fresh context and take one step."""
    o = H.CpuWorld("orc"); o.scene("Packed:4"); o.step(25)
    bodies, forces = o.bodies(), o.forces()
    g = H.HipWorld(); g.scene("Packed:4")
    g.set_bodies(bodies); g.set_forces(forces)
    H.assert_same_world(o, g, "snapshot upload round trip", check_rows=True, check_diag=False)
    o.step(); g.step()
    H.assert_same_world(o, g, "one teacher-forced step")
    o.step(5); g.step(5)
    H.assert_same_world(o, g, "five more steps")


@pytest.mark.parametrize("tag,scene", [("pyramid", "Pyramid"), ("jointed", "@jointed"), ("stack", "Stack")])
def test_row_api_on_device_is_bit_exact_with_the_reference_virtuals(tag, scene):
    """Force::computeConstraint / computeDerivatives through the C ABI (solver.h:104-106) against golden vectors produced by
    calling the reference's own virtuals (oracle/ref_shim.cpp: ref_force_compute_constraint / _derivatives) on a snapshot:
    rows C / fmin / fmax / lambda, the in-place mutations of the force (normal, stick, penetration, cone-clamped lambda) and
    the Jacobians of both endpoints for every row, bit for bit."""
    z = np.load(os.path.join(H.ROOT, "tests", "golden", "row_api_vectors.npz"))
    bodies, forces, after, derivs, alpha = z[f"{tag}_bodies"], z[f"{tag}_forces"], z[f"{tag}_after"], z[f"{tag}_derivs"], float(z[f"{tag}_alpha"])
    g = H.HipWorld()
    H.build_row_api_scene(g, scene)
    g.step(int(z[f"{tag}_steps"]))                               # same force kinds / creation order as the snapshot
    g.set_bodies(bodies.view(H.BODY_DT)); g.set_forces(forces.view(H.FORCE_DT))
    nf = len(forces)
    assert nf == len(g.forces()) and nf > 0
    for i in range(nf):
        rows = int(forces["rows"][i])
        assert g.w.force_row_count(i) == rows
        Cc, lo, hi, lam = g.w.force_compute_constraint(i, alpha)
        for name, got in (("C", Cc), ("fmin", lo), ("fmax", hi), ("lambda", lam)):
            assert H.same_bits(got[:rows], after[name][i][:rows]), (tag, i, name, got[:rows], after[name][i][:rows])
        for e, body in enumerate((int(forces["bodyA"][i]), int(forces["bodyB"][i]))):
            for row in range(rows):
                Jl, Ja = g.w.force_compute_derivatives(i, body, row)
                assert H.same_bits(np.concatenate([Jl, Ja]), derivs[i, e, row]), (tag, i, e, row, Jl, Ja, derivs[i, e, row])
    got = g.forces()
    for name in ("C", "fmin", "fmax", "lambda", "penalty"):
        for i in range(nf):
            rows = int(forces["rows"][i])
            assert H.same_bits(got[name][i][:rows], after[name][i][:rows]), (tag, i, name)
    for i in range(nf):
        for ci in range(int(forces["numContacts"][i])):
            for name in ("normal", "penetration", "stick", "rA", "rB", "feature"):
                assert H.same_bits(np.asarray(got["contacts"][i][ci][name]), np.asarray(after["contacts"][i][ci][name])), (tag, i, ci, name)


def test_cli_headless_trace_md5():
    """`--nogfx --scene Stack --steps 300` reproduces the reference trace byte for byte."""
    exe = os.path.join(H.ROOT, "avbd-demo3d_amd", "avbd_demo3d")
    assert os.path.exists(exe), "CLI not built"
    for scene, md5 in (("Stack", "d5acb696d92d001afcd726553fddc04e"), ("TwoBlockDrop", "3f2d6a60b7a426efc49bf6735d0e740a"),
                       ("Pyramid", "db3fed3c95794b39711042a0e5bdcabd"), ("Wall", "c38136d2ed722e131450c0700881f8ab")):
        out = subprocess.run([exe, "--nogfx", "--scene", scene, "--steps", "300"], capture_output=True, check=True).stdout
        assert hashlib.md5(out).hexdigest() == md5, scene


def test_cli_headless_trace_md5_every_reference_scene():
    """All nine scenes of the reference's table (scenes.h:186-212), 300 steps each, through the headless CLI: the md5 of the
    whole stdout equals the reference build's (tests/golden/golden_meta.json, generated from oracle/_ref).  Stress1000 is
    301 201 lines; the two WIP scenes exercise joints, springs and ignore-collision markers."""
    import json
    meta = json.load(open(os.path.join(H.ROOT, "tests", "golden", "golden_meta.json")))["trace_md5_300_steps"]
    exe = os.path.join(H.ROOT, "avbd-demo3d_amd", "avbd_demo3d")
    assert len(meta) >= 9
    for scene, md5 in meta.items():
        out = subprocess.run([exe, "--nogfx", "--scene", scene, "--steps", "300"], capture_output=True, check=True).stdout
        assert hashlib.md5(out).hexdigest() == md5, scene


def test_cli_flag_aliases_and_unknown_scene():
    """main.cpp:195-219: `--headless`, `-s`, `-n` are aliases; an unknown scene name silently selects scene 0 ("Empty");
    `--steps` goes through atoi (garbage -> 0 steps)."""
    exe = os.path.join(H.ROOT, "avbd-demo3d_amd", "avbd_demo3d")
    run = lambda *a: subprocess.run([exe, *a], capture_output=True, check=True).stdout
    assert run("--headless", "-s", "Stack", "-n", "7") == run("--nogfx", "--scene", "Stack", "--steps", "7")
    assert run("--nogfx", "--scene", "NoSuchScene", "--steps", "3") == run("--nogfx", "--scene", "Empty", "--steps", "3")
    assert run("--nogfx", "--scene", "Ground", "--steps", "abc").splitlines() == [b"Running in headless mode: scene 'Ground', steps=0"]


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the driver's fields plus the roofline and cpu_baseline objects."""
    import json
    import sys
    out = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-reference", "--no-strong-n1"],
                         capture_output=True, text=True, check=True, cwd=H.ROOT).stdout
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 6 and b["warmup"] == 2 and b["higher_is_better"] is True and b["vs_baseline"] is None
    assert b["unit"] == "steps/s" and b["dtype"] == "f32" and b["data"] == "synthetic" and b["scaling"] is None   # one GPU: no scaling claim
    assert "workload" in b["config"] and "Stress1000" in b["config"]["workload"]
    assert abs(b["value"] - 1000.0 / b["ms_per_step"]) / b["value"] < 1e-3
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    # PMC counters cannot be read in-process: a traffic figure is either absent or labelled with the committed profile it comes from
    assert (r["traffic"] is None) == (r["traffic_source"] is None)
    assert r["traffic"] is None or "profiles/" in r["traffic_source"]
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["parity"] == "bit-exact" and c["value"] > 0


def test_bench_sharded_path_reports_strong_scaling_of_one_scene():
    """The N > 1 code path (torch.distributed + RCCL + sharded schedule) with one rank on a small scene: `value` is the
    steps/s of the ONE scene all ranks work on (never multiplied by the rank count) and the line says "strong"."""
    import json
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--force-dist", "--scene", "Packed:6", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, check=True, cwd=H.ROOT, env=env).stdout
    b = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert b["scaling"] == "strong" and b["n_gpus"] == 1 and b["steps"] == 5
    assert abs(b["value"] - 1000.0 / b["ms_per_step"]) / b["value"] < 1e-3
    assert "Packed:6" in b["config"]["workload"] and "all-gather" in b["config"]["parallelism"]
