#!/usr/bin/env python3
"""bench.py — solver steps/s of the AVBD hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one Solver::step() (solver.cpp:255-514) on the device.  Workload at N=1 is
BASELINE.json configs[1]: Stress1000 (1000 unit cubes, iterations=20, beta=3e4, gamma=0.995),
measured in its contact-rich steady window: the scene is first simulated `--presim` steps untimed
(default 400, the window SURVEY.md §8d names), then W warm-up steps, then exactly K timed steps with
all state resident in HBM, bracketed by barrier + device synchronize; value = steps / max-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel (k_iterate: the persistent launch that runs the whole iteration loop — resident
                schedule, one wavefront per body; k_primal with AVBD_MODE=levels): algorithmic bytes per launch
                (SURVEY.md §8d model, recomputed from the live scene every step) / mean launch duration from HIP
                events recorded around every launch on the solver's own stream, against 8 TB/s HBM peak.
                The kernel is bound by a dependency chain (203 sequential body solves per step), not by bytes:
                DESIGN.md §5 gives the per-hop latency budget that actually bounds it.
  cpu_baseline  the CPU oracle (port of the reference, bit-identical to it) timed on this box's host, one
                thread, started from the SAME state as the timed GPU window; when it covers the whole window the
                final states are compared bit for bit ("parity": "bit-exact").
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_pkg():
    import importlib.util
    name = "avbd_demo3d_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(ROOT, "avbd-demo3d_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def algorithmic_bytes(stats, iterations):
    """B_iter = 100*Nb + 124*Ninc + 160*Nc ; B_step = iters*B_iter + 264*Nb + 704*Nm."""
    nb, ninc, nc, nm = int(stats["numDynamic"]), int(stats["numIncidences"]), int(stats["numContacts"]), int(stats["numManifolds"])
    b_iter = 100 * nb + 124 * ninc + 160 * nc
    return {"B_iter": b_iter, "B_step": iterations * b_iter + 264 * nb + 704 * nm,
            "B_primal_iter": 100 * nb + 124 * ninc, "B_dual_iter": 160 * nc}


def dependency_chain_hops(forces, bodies, total_iterations):
    """Length (in body solves) of the longest dependency chain of the exact Gauss-Seidel sweep over `total_iterations`
    iterations on the current contact graph: task (b, it) needs (o, it) of every later-created swept neighbour o and
    (e, it-1) of every earlier-created one (whose task also runs the dual of the shared contact) and (b, it-1).
    This chain, not bytes, is what bounds k_iterate."""
    dyn = bodies["invMass"] > 0
    n = len(bodies)
    later = [[] for _ in range(n)]
    earlier = [[] for _ in range(n)]
    for a, b, nc in zip(forces["bodyA"], forces["bodyB"], forces["numContacts"]):
        a, b = int(a), int(b)
        if nc <= 0 or a < 0 or b < 0 or not (dyn[a] and dyn[b]):
            continue
        lo, hi = (a, b) if a < b else (b, a)
        later[lo].append(hi)
        earlier[hi].append(lo)
    prev = np.zeros(n, np.int64)
    longest = 0
    for _ in range(total_iterations):
        cur = np.zeros(n, np.int64)
        for b in range(n - 1, -1, -1):                       # the sweep visits later-created bodies first
            if not dyn[b]:
                continue
            d = prev[b]
            for o in later[b]:
                d = max(d, cur[o])
            for e in earlier[b]:
                d = max(d, prev[e])
            cur[b] = d + 1
        prev = cur
        longest = int(cur.max()) if n else 0
    return longest


PMC_SUMMARIES = {"Stress1000": "r6_stress1000_cli_summary.json", "StressN:20": "r6_8k_cli_summary.json", "StressN:40": "r6_64k_cli_summary.json",
                 "StressN:40+welds:10000": "r6_64k_welds_cli_summary.json",
                 # the fast sweep (option "sweep" = 1, CLI --mode fast): profiles/run_all_r6.sh
                 "fast:Stress1000": "r6_fast_1000_cli_summary.json", "fast:StressN:20": "r6_fast_8k_cli_summary.json", "fast:StressN:40": "r6_fast_64k_cli_summary.json"}


def source_id():
    """sha256 over the sources the library is built from (tools/source_id.py): what a committed profile must have been collected on"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import source_id as S
        return S.source_id()
    except Exception:
        return None


def pmc_traffic(scene, fused):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside the process, so this is NOT a
    measurement of this run: it is the figure of the committed rocprofv3 --pmc passes of the same command on the same build
    (profiles/collect.sh; (2*FETCH_SIZE + WRITE_SIZE) * 1024, MI355X_MICROARCH.md §HBM), labelled as such by `traffic_source`;
    null for any workload or mode without a committed profile."""
    name = PMC_SUMMARIES.get(scene)
    if name is None or not fused:
        return None, None
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    try:
        summary = json.load(open(path))
        d = summary["counters_per_dispatch_of_dominant_kernel"]     # profiles/collect_cli.sh + summarise_cli.py: KB per dispatch
        same = summary.get("source_id") is not None and summary.get("source_id") == source_id()
        return round((2.0 * d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024.0, 1), \
            "committed profile profiles/" + name + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the CLI on the same window; FETCH_SIZE x2 on gfx950; not this run; " + \
            ("collected on these very sources: source_id matches)" if same else "collected on EARLIER sources: source_id differs)")
    except Exception:
        return None, None


STRONG_SCENE, STRONG_PRESIM = "StressN:40", 460     # BASELINE.json configs[3]: 64 000 cubes, steady window after the pile has landed


def spawn_ranks(args):
    """`--gpus N` without torchrun's environment: start the N ranks as CHILD processes (python -m torch.distributed.run) before this
    process has touched the GPU, relay their output, exit with their code.  Nothing that has initialised HIP is ever re-exec'ed."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("AVBD_BENCH_SPAWN_ECHO"):                      # tests: show the launch instead of doing it
        print(json.dumps({"spawn": cmd}))
        sys.exit(0)
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def cpu_baseline(pkg, world, scene, snapshot_bodies, snapshot_forces, step_index, cpu_steps):
    """Time the CPU oracle (kind 'port') from the given snapshot on one host thread."""
    import harness as H
    o = H.CpuWorld("orc")
    o.scene(scene)
    o.set_bodies(snapshot_bodies.view(H.BODY_DT))
    o.set_forces(snapshot_forces.view(H.FORCE_DT))
    o.set_step_index(step_index)
    t0 = time.perf_counter()
    o.step(cpu_steps)
    dt = time.perf_counter() - t0
    return o, dt


def reference_baseline(scene, start_step, steps, end_bodies):
    """Time the REAL reference (oracle/_ref/libavbd_ref.so = the unmodified sources of /root/reference compiled by
    oracle/Makefile) on the same window: it cannot load a snapshot, so it simulates the scene from step 0 untimed, then
    `steps` timed steps; its state after them is compared bit for bit with the GPU's when the windows coincide."""
    import harness as H
    if not os.path.exists(H.lib_path("ref")):
        return None
    r = H.CpuWorld("ref")
    r.scene(scene)
    r.step(start_step)
    t0 = time.perf_counter()
    r.step(steps)
    dt = time.perf_counter() - t0
    same = None
    if end_bodies is not None:
        same = H.same_bits(r.bodies(), end_bodies.view(H.BODY_DT))
    return steps / dt, dt / steps * 1e3, same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default=None, help="override the workload (e.g. StressN:20, Packed:10)")
    ap.add_argument("--presim", type=int, default=None, help="untimed steps before warm-up (default 400 for falling scenes)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--force-dist", action="store_true", help="take the multi-GPU code path (torch.distributed + RCCL + sharded schedule) even with one rank")
    ap.add_argument("--no-reference", action="store_true", help="cpu_baseline from the port only (skip the run of oracle/_ref)")
    ap.add_argument("--no-strong-n1", action="store_true", help="N=1: skip the extra 64 000-cube line (the N=1 point of the strong-scaling experiment)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus and not (args.gpus == 1 and world_size == 1):
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node {args.gpus} ...)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    sharded = world_size > 1 or args.force_dist
    if sharded:
        import torch
        import torch.distributed as dist
        # the persistent all-gather form runs its collective beside a launch that fills the device but four workgroup slots: RCCL reads the
        # channel cap once per process, when the first communicator is created — that is the one below
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "4")
        # AVBD_BENCH_ONE_DEVICE=1: every rank on device 0 (how the multi-process path is exercised on a one-GPU box: RCCL refuses two ranks on
        # one device, so rendezvous goes over gloo and only the exact-cut form — which needs no collective — can run)
        one_device = os.environ.get("AVBD_BENCH_ONE_DEVICE") == "1"
        if one_device:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    tdev = "cpu" if (sharded and os.environ.get("AVBD_BENCH_ONE_DEVICE") == "1") else "cuda"

    pkg = load_pkg()
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()
    if args.force_dist and world_size == 1:
        os.environ["AVBD_FORCE_SHARDED"] = "1"                   # one rank, but the sharded schedule (exchange every iteration)
    # N = 1: BASELINE configs[1] (Stress1000).  N > 1: north_star's strong-scaling experiment, the SAME 64 000-cube scene sharded over
    # the N GPUs by slabs along x (configs[3]); value = steps/s of that one scene, never multiplied by N
    scene = args.scene or ("Stress1000" if not sharded else STRONG_SCENE)
    presim = args.presim if args.presim is not None else (0 if scene.startswith("Packed") else (STRONG_PRESIM if scene == STRONG_SCENE else 400))

    EXACT_FORM = "exact cuts (one process per GPU, record windows over hipIpc; bit-identical to one GPU)"

    def vote(ok):
        """every rank takes part; True only when every rank said ok"""
        if dist is None:
            return bool(ok)
        import torch
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    def make_world():
        w = pkg.World(local_rank)
        w.scene(scene)
        return w

    def make_sharded_world(form):
        """One world per rank in the given form, or None on EVERY rank.  The parts that can fail on one rank alone (context, scene,
        ownership, window export / attach, communicator) run under try and are followed by a vote; the collectives in between are
        entered by every rank or by none (a rank that raised used to jump to the all_reduce while the others sat in
        all_gather_object or the barrier)."""
        import torch
        w, err, mine = None, None, None
        try:
            w = make_world()
            b0 = w.bodies()
            owner = pkg.partition_by_axis(b0["position"], b0["invMass"], world_size, axis=0)
            w.set_ownership(world_size, rank, owner)
            if form == EXACT_FORM:
                mine = w.exact_window_export()
        except Exception as e:
            err = repr(e)
        if vote(err is None):
            if form == EXACT_FORM:
                # every rank exports its record window, all ranks open all windows: nothing is exchanged during a step, every rank polls
                # its neighbours' stamped records where they are written
                handles, devs = [None] * world_size, [None] * world_size
                dist.all_gather_object(handles, mine)
                dist.all_gather_object(devs, (os.uname().nodename, torch.cuda.current_device()))
                try:
                    w.exact_window_attach(handles, ranks_on_this_device=sum(1 for d in devs if d == devs[rank]))
                except Exception as e:
                    err = repr(e)
            else:
                # (rank 0's uid under try, broadcast unconditionally — None on failure —, then a vote: no rank is left alone in a collective)
                uid = [None]
                if rank == 0:
                    try:
                        uid = [pkg.comm_unique_id()]
                    except Exception as e:
                        err = repr(e)
                dist.broadcast_object_list(uid, src=0)
                if vote(uid[0] is not None and err is None):
                    try:
                        w.comm_init(world_size, rank, uid[0])
                    except Exception as e:
                        err = repr(e)
                elif err is None:
                    err = "rank 0 could not make the communicator's unique id"
            if vote(err is None):
                dist.barrier()                                       # the ranks start stepping together (a wave gives up after 4 s without progress)
                return w
        if err is not None:
            print(f"bench.py rank {rank}: the form '{form}' could not be set up ({err})", file=sys.stderr)
        if w is not None:
            try:
                w.close()
            except Exception:
                pass
        return None

    want_cpu = (rank == 0 and not sharded and not args.no_cpu)
    prof_steps = min(args.steps, 50)

    def measure(w):
        """untimed steps, the timed region (exactly K steps, barrier + synchronize on both sides, max over ranks), then the
        profiled steps (HIP events around every launch on the solver's stream).  With several ranks every barrier is a vote
        (an all-reduce every rank enters whether its own leg failed or not): None on EVERY rank when a leg failed anywhere."""
        m, err = {}, None
        try:
            if presim:
                w.step(presim)
            w.step(args.warmup)
            w.synchronize()
            # snapshot for the CPU leg / parity (rank 0, N=1 only)
            m = {"snap_b": None, "snap_f": None, "snap_step": w.step_index()}
            if want_cpu:
                m["snap_b"], m["snap_f"] = w.bodies(), w.forces()
        except Exception as e:
            err = e
        if not vote(err is None):                                    # barrier
            return fail(err, "untimed steps")
        try:
            w.synchronize()
            t0 = time.perf_counter()
            w.step(args.steps)
            w.synchronize()
            m["elapsed"] = time.perf_counter() - t0
        except Exception as e:
            err = e
        if not vote(err is None):                                    # barrier
            return fail(err, "timed steps")
        try:
            w.synchronize()
            m["end_b"] = w.bodies() if want_cpu else None
            m["stats"] = w.stats()
            m["p"] = None
            w.profile_enable(True)
            w.profile_reset()
            w.step(prof_steps)
            w.synchronize()
            m["p"] = w.profile()
            w.profile_enable(False)
        except Exception as e:
            err = e
        if not vote(err is None):
            # the timed legs are in: a form whose PROFILED steps failed somewhere keeps its figure and loses its roofline object (sharded runs; one GPU raises)
            fail(err, "profiled steps")
            m["p"] = None
        return m

    def fail(err, leg):
        if err is not None:
            if dist is None:
                raise err
            print(f"bench.py rank {rank}: the run failed in its {leg} ({err})", file=sys.stderr)
        return None

    w = make_world()                                                 # (sharded: every form gets a world of its own below)
    iterations = int(w.params()["iterations"])
    shard_form = None
    shard_other = None
    fast_sharded = None
    if sharded:                                                      # (--force-dist with one rank takes the same path)
        # Both forms of the sharded step are measured — one persistent launch per step with the collective on a second stream BESIDE
        # it, and one launch per iteration with the collective between the launches — and the line reports the faster
        # one (config.shard_form; the other one's figure is in config.shard_other_form).  The persistent form could not be validated
        # on a multi-GPU box: a rank that fails it anywhere (a wait that times out, an RCCL error) makes every rank drop it.
        import torch
        forms = ["one launch per iteration"] if os.environ.get("AVBD_SHARD_PERSISTENT") == "0" else \
                ["persistent", "persistent, exchange stream gated by kernels instead of stream wait / write values", "one launch per iteration"]
        if world_size > 1 and os.environ.get("AVBD_BENCH_EXACT", "1") != "0":
            forms = [EXACT_FORM] + forms
        if os.environ.get("AVBD_BENCH_ONE_DEVICE") == "1":
            forms = [EXACT_FORM]
        results = {}
        for k, form in enumerate(forms):
            # (the library takes the persistent form with a real communicator only on request — it could not be validated on a
            # multi-GPU box —, so every form is asked for explicitly, in a context of its own)
            os.environ["AVBD_SHARD_PERSISTENT"] = "1" if form.startswith("persistent") else "0"
            if form.startswith("persistent, exchange stream gated"):
                os.environ["AVBD_SHARD_GATE"] = "kernel"
            else:
                os.environ.pop("AVBD_SHARD_GATE", None)
            try:
                if w is not None:
                    w.close()
            except Exception:
                pass
            w = make_sharded_world(form)                             # None on every rank or a world on every rank
            if w is None:
                continue
            mk = measure(w)                                          # None on every rank when any rank failed
            if mk is not None:
                te = torch.tensor([mk["elapsed"]], dtype=torch.float64, device=tdev)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                mk["elapsed"] = float(te.item())
                results[form] = mk
        if not results:
            sys.exit("bench.py: the sharded run failed in every form")
        # the same sharded scene in the FAST sweep (option "sweep" = 1): one launch of k_fast_sweep per iteration, the all-gather between the
        # launches — north_star's own multi-GPU form; reported beside the exact forms (config.fast_sharded), never as `value`
        fast_sharded = None
        if os.environ.get("AVBD_BENCH_FAST_SHARDED", "1") != "0" and os.environ.get("AVBD_BENCH_ONE_DEVICE") != "1":
            os.environ["AVBD_SHARD_PERSISTENT"] = "0"
            os.environ.pop("AVBD_SHARD_GATE", None)
            try:
                if w is not None:
                    w.close()
            except Exception:
                pass
            w = make_sharded_world("one launch per iteration")
            if w is not None:
                ok = True
                try:
                    w.set_option("sweep", 1)
                except Exception:
                    ok = False
                if vote(ok):
                    mk = measure(w)
                    if mk is not None:
                        te = torch.tensor([mk["elapsed"]], dtype=torch.float64, device=tdev)
                        dist.all_reduce(te, op=dist.ReduceOp.MAX)
                        fast_sharded = {"form": "fast sweep (multi-colour Gauss-Seidel inside every slab), one launch per iteration, RCCL all-gather of 32 B/body between the launches; Jacobi across the cuts",
                                        "ms_per_step": round(float(te.item()) / args.steps * 1e3, 5), "value": round(args.steps / float(te.item()), 3), "unit": "steps/s",
                                        "schedule": int(mk["stats"]["schedule"]), "tolerance": "DESIGN.md section 2b (fast mode) + Jacobi across the cuts (section 7)"}
        shard_form = min(results, key=lambda f: results[f]["elapsed"])
        m = results[shard_form]
        shard_other = [{"form": o, "ms_per_step": round(results[o]["elapsed"] / args.steps * 1e3, 5)} if o in results else {"form": o, "failed": True}
                       for o in forms if o != shard_form]
    else:
        m = measure(w)
        if m is None:
            sys.exit("bench.py: the run failed")
        if dist is not None:
            import torch
            te = torch.tensor([m["elapsed"]], dtype=torch.float64, device=tdev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            m["elapsed"] = float(te.item())
    snap_b, snap_f, snap_step, elapsed, end_b, stats, p = m["snap_b"], m["snap_f"], m["snap_step"], m["elapsed"], m["end_b"], m["stats"], m["p"]
    ab = algorithmic_bytes(stats, iterations)

    # ---- roofline of the dominant kernel ----
    roof = None
    if p is not None and p["launches_primal"] > 0 and p["ms_primal"] > 0:
        per_launch_s = p["ms_primal"] * 1e-3 / p["launches_primal"]
        bytes_per_launch = p["algorithmic_bytes_primal"] / p["launches_primal"]
        achieved = bytes_per_launch / per_launch_s / 1e9
        fused = p["launches_primal"] <= 1.5 * max(1, p["steps"]) * max(1, world_size if world_size > 1 else 1) or os.environ.get("AVBD_MODE") != "levels"
        traffic, traffic_source = pmc_traffic(scene, fused and not sharded)
        sched = int(stats["schedule"])
        roof = {"bound": "hbm", "limited_by": "dependency latency (a chain of body solves, see latency_floor), not bytes", "kernel": ("k_iterate_groups" if sched == 4 else ("k_iterate_single" if sched == 2 and os.environ.get("AVBD_SINGLE_KERNEL", "1") != "0" else "k_iterate")) if fused else "k_primal", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_source,
                "bytes_per_launch": round(bytes_per_launch, 1), "us_per_launch": round(per_launch_s * 1e6, 3),
                "launches_per_step": round(p["launches_primal"] / max(1, p["steps"]), 2),
                "step_algorithmic_GBps": round(ab["B_step"] * args.steps / elapsed / 1e9, 3),
                "kernel_ms_per_step": {k[3:]: round(float(p[k]) / max(1, p["steps"]), 4) for k in
                                       ("ms_broadphase", "ms_narrowphase", "ms_topology", "ms_predict", "ms_primal", "ms_dual", "ms_velocity")}}

    if roof is not None and rank == 0 and not sharded and int(stats["numBodies"]) <= 20000:
        try:
            hops = dependency_chain_hops(w.forces(), w.bodies(), iterations)
            roof["dependency_chain"] = {"sequential_body_solves_per_step": hops,
                                        "us_per_solve_and_handoff": round(roof["us_per_launch"] / max(1, hops), 3)}
            # what the launch cannot go below while results must equal the reference's: every hop of the chain hands a pose from one
            # wavefront to another — 0.8 us through memory (sc1 store -> fabric -> sc1 load), ~0.2 us through LDS when producer and
            # consumer share a workgroup (tools/debug_critical.py)
            roof["latency_floor"] = {"hops": hops, "handoff_us": 0.8, "handoff_us_same_workgroup": 0.2,
                                     "frac": round(hops * 0.8 / max(1e-9, roof["us_per_launch"]), 4),
                                     "frac_all_hops_through_lds": round(hops * 0.2 / max(1e-9, roof["us_per_launch"]), 4)}
        except Exception as e:                                # informative only
            roof["dependency_chain"] = {"error": repr(e)}

    # ---- CPU baseline on the same window ----------------------------------------------------
    cpu = None
    if want_cpu:
        try:
            ms_guess = 50.0 * max(1, int(stats["numDynamic"])) / 1000.0  # ~45 ms/step at Stress1000 steady state
            cpu_steps = args.cpu_steps or max(1, min(args.steps, int(20000.0 / ms_guess)))
            o, dt = cpu_baseline(pkg, w, scene, snap_b, snap_f, snap_step, cpu_steps)
            parity = "not-compared"
            if cpu_steps == args.steps:
                import harness as H
                parity = "bit-exact" if H.same_bits(o.bodies(), end_b.view(H.BODY_DT)) else "MISMATCH"
            cpu = {"value": round(cpu_steps / dt, 3), "unit": "steps/s", "cores": 1, "kind": "port",
                   "sample": f"{cpu_steps} steps of {scene} from the GPU run's own state at step {snap_step} (oracle/liboracle.so, "
                             f"-O2 -ffp-contract=off, 1 thread of {os.cpu_count()} host cores)",
                   "ms_per_step": round(dt / cpu_steps * 1e3, 3), "parity": parity}
            # the real reference on the same window, when its build travelled with the snapshot (oracle/_ref)
            if not args.no_reference and args.scene is None and snap_step <= 1000:
                ref = reference_baseline(scene, snap_step, args.steps if args.steps <= 200 else 200, end_b if args.steps <= 200 else None)
                if ref is not None:
                    port = cpu
                    cpu = {"value": round(ref[0], 3), "unit": "steps/s", "cores": 1, "kind": "reference",
                           "sample": f"{min(args.steps, 200)} steps of {scene}, steps {snap_step}..{snap_step + min(args.steps, 200)} of the reference's own run "
                                     f"from step 0 (oracle/_ref/libavbd_ref.so: the unmodified reference sources, g++ -O2 -ffp-contract=off, "
                                     f"1 thread of {os.cpu_count()} host cores)",
                           "ms_per_step": round(ref[1], 3),
                           "parity": "not-compared" if ref[2] is None else ("bit-exact" if ref[2] else "MISMATCH"),
                           "port": {"value": port["value"], "ms_per_step": port["ms_per_step"], "parity": port["parity"]}}
        except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
            cpu = {"value": None, "unit": "steps/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}

    # SURVEY.md §8d also asks for the whole 600-step run from scratch (free fall + landing + steady state), untimed parts none
    whole = None
    if rank == 0 and not sharded and args.scene is None:
        w2 = pkg.World(local_rank)
        w2.scene(scene)
        w2.synchronize()
        t0 = time.perf_counter()
        w2.step(600)
        w2.synchronize()
        tw = time.perf_counter() - t0
        whole = {"steps": 600, "steps_per_s": round(600 / tw, 3), "ms_per_step": round(tw / 600 * 1e3, 5)}
        w2.close()

    # Nested N = 1 lines of the larger configurations, each with a roofline object of its own (the launches where "% of HBM peak" means
    # something): BASELINE configs[3]'s scene on this one GPU = the N = 1 point of the strong-scaling experiment, and configs[2]
    # (StressN:20, 8 000 cubes: "HBM roofline measurement"), driver-timed since round 4.
    def nested_line(nscene, npresim, k3):
        try:
            w3 = pkg.World(local_rank)
            w3.scene(nscene)
            w3.step(npresim)
            w3.synchronize()
            t0 = time.perf_counter()
            w3.step(k3)
            w3.synchronize()
            t3 = time.perf_counter() - t0
            st3 = w3.stats()
            out = {"workload": f"{nscene} (iterations={int(w3.params()['iterations'])}) steady window: steps {npresim}..{npresim + k3}",
                   "value": round(k3 / t3, 3), "unit": "steps/s", "ms_per_step": round(t3 / k3 * 1e3, 4), "n_gpus": 1,
                   "bodies": int(st3["numBodies"]), "contacts": int(st3["numContacts"]), "joints": int(st3["numJoints"]), "schedule": int(st3["schedule"])}
            # HIP events around the launches of 20 more steps
            w3.profile_enable(True); w3.profile_reset(); w3.step(20); w3.synchronize(); p3 = w3.profile(); w3.profile_enable(False)
            if p3["launches_primal"] > 0 and p3["ms_primal"] > 0:
                s3 = p3["ms_primal"] * 1e-3 / p3["launches_primal"]
                b3 = p3["algorithmic_bytes_primal"] / p3["launches_primal"]
                t3r, t3s = pmc_traffic(nscene, True)
                out["roofline"] = {"bound": "hbm", "kernel": "k_iterate_groups" if int(st3["schedule"]) == 4 else "k_iterate", "achieved": round(b3 / s3 / 1e9, 3), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(b3 / s3 / 1e9 / 8000.0, 6), "traffic": t3r, "traffic_source": t3s,
                                   "bytes_per_launch": round(b3, 1), "us_per_launch": round(s3 * 1e6, 3),
                                   "kernel_ms_per_step": {k[3:]: round(float(p3[k]) / max(1, p3["steps"]), 4) for k in
                                                          ("ms_broadphase", "ms_narrowphase", "ms_topology", "ms_predict", "ms_primal", "ms_dual", "ms_velocity")}}
            w3.close()
            return out
        except Exception as e:
            return {"error": repr(e)}

    def fast_line(nscene, npresim, k3):
        """The same scene in the fast sweep (option "sweep" = 1: multi-colour Gauss-Seidel, tree-reduced sums), driver-timed like the
        exact lines, with a roofline object of its own (same byte model) and the tolerance it is held to, MEASURED here: one step
        of the fast mode against one step of the exact mode from the same state at the end of the timed window."""
        try:
            import harness as H
            w3 = pkg.World(local_rank)
            w3.scene(nscene)
            w3.set_option("sweep", 1)
            w3.step(npresim)
            w3.synchronize()
            t0 = time.perf_counter()
            w3.step(k3)
            w3.synchronize()
            t3 = time.perf_counter() - t0
            st3 = w3.stats()
            plan = w3.fast_plan_info()
            out = {"workload": f"{nscene} (iterations={int(w3.params()['iterations'])}) steady window: steps {npresim}..{npresim + k3}, fast sweep from step 0",
                   "mode": "fast (option sweep = 1 / --mode fast: multi-colour Gauss-Seidel, DPP tree sums; within the stated tolerance of the exact mode, not bit-identical)",
                   "value": round(k3 / t3, 3), "unit": "steps/s", "ms_per_step": round(t3 / k3 * 1e3, 4), "n_gpus": 1,
                   "bodies": int(st3["numBodies"]), "contacts": int(st3["numContacts"]), "joints": int(st3["numJoints"]), "schedule": int(st3["schedule"]), "colours": plan["colors"], "tiles": plan["tiles"]}
            w3.profile_enable(True); w3.profile_reset(); w3.step(20); w3.synchronize(); p3 = w3.profile(); w3.profile_enable(False)
            if p3["launches_primal"] > 0 and p3["ms_primal"] > 0:
                s3 = p3["ms_primal"] * 1e-3 / p3["launches_primal"]
                b3 = p3["algorithmic_bytes_primal"] / p3["launches_primal"]
                t3r, t3s = pmc_traffic("fast:" + nscene, True)
                chain = plan["colors"] * int(w3.params()["iterations"])
                out["roofline"] = {"bound": "hbm", "limited_by": "dependency latency: colours x iterations dependent body solves per step", "kernel": "k_fast_sweep",
                                   "achieved": round(b3 / s3 / 1e9, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(b3 / s3 / 1e9 / 8000.0, 6),
                                   "traffic": t3r, "traffic_source": t3s, "bytes_per_launch": round(b3, 1), "us_per_launch": round(s3 * 1e6, 3),
                                   "dependency_chain": {"sequential_body_solves_per_step": chain, "us_per_solve_and_handoff": round(s3 * 1e6 / max(1, chain), 3)},
                                   "kernel_ms_per_step": {k[3:]: round(float(p3[k]) / max(1, p3["steps"]), 4) for k in
                                                          ("ms_broadphase", "ms_narrowphase", "ms_topology", "ms_predict", "ms_primal", "ms_dual", "ms_velocity")}}
            # the tolerance, measured: fork the state, one step exact, one step fast
            b, f, si = w3.bodies(), w3.forces(), w3.step_index()
            we = pkg.World(local_rank); we.scene(nscene); we.set_bodies(b); we.set_forces(f); we.set_step_index(si)
            wf = pkg.World(local_rank); wf.scene(nscene); wf.set_option("sweep", 1); wf.set_bodies(b); wf.set_forces(f); wf.set_step_index(si)
            we.step(1); wf.step(1)
            be, bf = we.bodies(), wf.bodies()
            dp = float(np.linalg.norm(be["position"].astype(np.float64) - bf["position"].astype(np.float64), axis=1).max())
            dq = np.abs(np.sum(be["orientation"].astype(np.float64) * bf["orientation"].astype(np.float64), axis=1))
            dr = float((2.0 * np.arccos(np.clip(dq, 0.0, 1.0))).max())
            big = int(st3["numBodies"]) > 2000
            if int(st3["numJoints"]) > 0:
                # BASELINE configs[4]: body-body welds with the reference's angular Jacobian sign turn by 146 degrees in ONE step in every mode;
                # any two orders of the sweep end metres apart — a throughput workload, held to schedule / determinism / finiteness (tests/test_gpu_fast.py), not to a distance
                out["tolerance"] = {"criterion": "none for this scene: the reference's own weld Joints blow up (SURVEY.md section 7-5: bug-compatible angular sign), one step of ANY two sweep orders differs by metres; "
                                                 "Joint rows are held to the exact mode on stable joints (tests/test_gpu_fast.py: world-anchored joints to rounding, corrected-sign welds within 5e-3 m per step)",
                                    "bound": None, "measured": {"dpos_m": dp, "drot_rad": dr}, "within_bound": None}
                we.close(); wf.close(); w3.close()
                return out
            out["tolerance"] = {"criterion": "one step of the fast sweep against one step of the exact mode (= the reference, bit for bit) from the same state: max |dpos| over the bodies in metres, max rotation angle in radians (DESIGN.md section 2b)",
                                "bound": {"dpos_m": 0.1 if big else 1e-3, "drot_rad": 0.3 if big else 5e-3,
                                          "why": "a pile still in motion (violent-phase bound)" if big else "a settled pile (steady-state bound)"},
                                "measured": {"dpos_m": dp, "drot_rad": dr}, "within_bound": bool(dp <= (0.1 if big else 1e-3) and dr <= (0.3 if big else 5e-3))}
            we.close(); wf.close(); w3.close()
            return out
        except Exception as e:
            return {"error": repr(e)}

    strong_n1 = roofline_8k = joints_n1 = None
    fast_1000 = fast_8k = fast_64k = fast_64k_welds = None
    if rank == 0 and not sharded and args.scene is None and not args.no_strong_n1:
        fast_1000 = fast_line("Stress1000", 420, 200)
        fast_8k = fast_line("StressN:20", 410, 50)
        fast_64k = fast_line(STRONG_SCENE, STRONG_PRESIM + 10, 50)
        fast_64k_welds = fast_line(STRONG_SCENE + "+welds:10000", STRONG_PRESIM + 10, 30)      # BASELINE configs[4] in the fast sweep (Joint halves as lane entries)
        strong_n1 = nested_line(STRONG_SCENE, STRONG_PRESIM + 10, 50)
        roofline_8k = nested_line("StressN:20", 410, 50)
        # BASELINE configs[4]'s scene on this one GPU: 64 000 cubes + 10 000 weld Joints between x-neighbours (host/scenes.h "+welds:<k>"); the Joint rows run in
        # the group schedule since round 5 (a lane holds a contact or a joint half) — one Joint used to send the whole sweep to the resident schedule's parked form
        joints_n1 = nested_line(STRONG_SCENE + "+welds:10000", STRONG_PRESIM + 10, 30)

    if rank == 0:
        total_steps = args.steps                                 # steps of the one scene every rank works on
        line = {
            "metric": "solver steps/sec + ms/step, Stress1000 @ iters=20, 1/2/4/8 MI355X",
            "value": round(total_steps / elapsed, 3), "unit": "steps/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
            "scaling": "strong" if sharded else None, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{scene} (iterations={iterations}) steady window: steps {snap_step}..{snap_step + args.steps}",
                       "mode": "exact (bit-identical to the reference CPU solver)" if world_size == 1 else
                               ("sharded with exact cuts: every rank sweeps its slab with the single-GPU protocol and polls its neighbours' stamped records in their windows (bit-identical to one GPU)"
                                if shard_form == EXACT_FORM else
                                "sharded: exact sweep inside every GPU's slab, poses of the other slabs from the last all-gather (Jacobi across the cut)"),
                       "parallelism": "1 GPU" if not sharded else
                                      (f"{world_size} x-slabs of one scene, no exchange during a step (peer windows over xGMI); value = steps/s of that scene" if shard_form == EXACT_FORM else
                                       f"{world_size} x-slabs of one scene, RCCL all-gather of 32 B/body after every iteration; value = steps/s of that scene"),
                       "shard_form": shard_form, "shard_other_form": shard_other, "fast_sharded": fast_sharded,
                       "strong_scaling_n1": strong_n1, "roofline_8k": roofline_8k, "joints_n1": joints_n1,
                       "fast_1000": fast_1000, "fast_8k": fast_8k, "fast_64k": fast_64k, "fast_64k_welds": fast_64k_welds,
                       "bodies": int(stats["numBodies"]), "manifolds": int(stats["numManifolds"]), "contacts": int(stats["numContacts"]),
                       "incidences": int(stats["numIncidences"]), "levels": int(stats["numLevels"]), "B_step_bytes": ab["B_step"],
                       "whole_run_from_scratch": whole},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
