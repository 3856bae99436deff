"""avbd-demo3d_amd — MI355X (gfx950) implementation of the AVBD constraint-solve hot path.

This Python module is plumbing only: a ctypes binding of the C ABI in include/avbd_hip.h
(libavbd_hip.so, built in-tree by `make -C avbd-demo3d_amd/csrc`), used by tests/ and bench.py.
The product is the HIP library plus the C++17 host mirror in avbd-demo3d_amd/host/.

There is NO CPU fallback: importing works without a GPU (so symbols can be checked), but creating
a `World` raises unless a gfx950 device and the compiled library are present.
"""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libavbd_hip.so")
if os.environ.get("AVBD_LIB"):                       # measurement builds (tools/): libavbd_hip_stats.so, an older build for A/B runs
    LIB_PATH = os.path.join(_HERE, os.environ["AVBD_LIB"]) if not os.path.isabs(os.environ["AVBD_LIB"]) else os.environ["AVBD_LIB"]
INCLUDE = os.path.join(os.path.dirname(_HERE), "include", "avbd_hip.h")

FLT_MAX = float(np.finfo(np.float32).max)

BODY_DT = np.dtype([
    ("position", "<f4", 3), ("orientation", "<f4", 4), ("linearVelocity", "<f4", 3), ("angularVelocity", "<f4", 3),
    ("prevLinearVelocity", "<f4", 3), ("prevAngularVelocity", "<f4", 3), ("initialPosition", "<f4", 3),
    ("initialOrientation", "<f4", 4), ("inertialPosition", "<f4", 3), ("inertialOrientation", "<f4", 4),
    ("size", "<f4", 3), ("mass", "<f4"), ("invMass", "<f4"), ("friction", "<f4"), ("radius", "<f4"),
])
CONTACT_DT = np.dtype([
    ("feature", "<i4"), ("rA", "<f4", 3), ("rB", "<f4", 3), ("normal", "<f4", 3),
    ("penetration", "<f4"), ("C0_n", "<f4"), ("C0_t", "<f4", 3), ("stick", "<i4"),
])
FORCE_DT = np.dtype([
    ("kind", "<i4"), ("bodyA", "<i4"), ("bodyB", "<i4"), ("rows", "<i4"),
    ("C", "<f4", 12), ("fmin", "<f4", 12), ("fmax", "<f4", 12), ("lambda", "<f4", 12),
    ("penalty", "<f4", 12), ("motor", "<f4", 12), ("stiffness", "<f4", 12),
    ("numContacts", "<i4"), ("combinedFriction", "<f4"), ("contacts", CONTACT_DT, 4),
    ("rA", "<f4", 3), ("rB", "<f4", 3), ("q0", "<f4", 4), ("restLength", "<f4"),
])
DIAG_DT = np.dtype([
    ("maxPenetration", "<f4"), ("maxConstraintViolation", "<f4"), ("maxLinearSpeed", "<f4"),
    ("maxAngularSpeed", "<f4"), ("maxNormalImpulse", "<f4"),
    ("activeContacts", "<i4"), ("activeManifolds", "<i4"), ("dynamicBodies", "<i4"),
])
PARAMS_DT = np.dtype([
    ("dt", "<f4"), ("gravity", "<f4", 3), ("iterations", "<i4"), ("alpha", "<f4"), ("beta", "<f4"),
    ("gamma", "<f4"), ("postStabilize", "<i4"),
])
STATS_DT = np.dtype([(n, "<i4") for n in (
    "numBodies", "numDynamic", "numManifolds", "numContacts", "numIncidences", "numCandidates", "numLevels",
    "numJoints", "numSprings", "numIgnores", "warnings", "schedule", "cutMode")])
PROFILE_DT = np.dtype([
    ("ms_broadphase", "<f8"), ("ms_narrowphase", "<f8"), ("ms_topology", "<f8"), ("ms_predict", "<f8"),
    ("ms_primal", "<f8"), ("ms_dual", "<f8"), ("ms_velocity", "<f8"),
    ("launches_primal", "<i8"), ("launches_dual", "<i8"), ("launches_total", "<i8"), ("steps", "<i8"),
    ("algorithmic_bytes_primal", "<f8"), ("algorithmic_bytes_dual", "<f8"),
])

EXPORTS = [
    "avbd_abi_version", "avbd_create", "avbd_destroy", "avbd_last_error", "avbd_default_params", "avbd_set_params",
    "avbd_get_params", "avbd_clear", "avbd_add_bodies", "avbd_add_joint", "avbd_add_spring", "avbd_add_ignore", "avbd_add_force",
    "avbd_step", "avbd_synchronize", "avbd_step_index", "avbd_get_diagnostics", "avbd_get_step_stats",
    "avbd_num_bodies", "avbd_download_bodies", "avbd_download_state13", "avbd_upload_state13", "avbd_upload_bodies",
    "avbd_num_forces", "avbd_download_forces", "avbd_upload_forces", "avbd_force_row_count",
    "avbd_force_compute_constraint", "avbd_force_compute_derivatives", "avbd_kat_collide", "avbd_kat_solve6",
    "avbd_kat_arith", "avbd_profile_enable", "avbd_profile_reset", "avbd_profile_get", "avbd_stream",
    "avbd_build_scene", "avbd_partition_by_axis", "avbd_set_ownership", "avbd_comm_unique_id", "avbd_comm_init",
    "avbd_group_step", "avbd_add_host_force", "avbd_set_host_force_callback", "avbd_debug_group_task_times", "avbd_download_force", "avbd_exchange_layout", "avbd_debug_iterate_stats", "avbd_debug_task_times", "avbd_remove_body", "avbd_remove_force", "avbd_save_snapshot", "avbd_load_snapshot", "avbd_set_step_index", "avbd_debug_topology_times", "avbd_debug_front_times", "avbd_debug_front_blocks", "avbd_pick", "avbd_get_warnings", "avbd_set_shard_mode", "avbd_group_rebalance", "avbd_rebalance_export", "avbd_rebalance_import", "avbd_set_option", "avbd_get_option", "avbd_option_count", "avbd_option_key", "avbd_exact_window_export", "avbd_exact_window_attach", "avbd_set_exchange_floats", "avbd_exchange_mismatches", "avbd_device_count", "avbd_set_joint_angular_sign", "avbd_fast_plan_info", "avbd_fast_colors",
]


class AvbdError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libavbd_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise AvbdError("hipcc build of libavbd_hip.so failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


_lib = None
_live_worlds = weakref.WeakSet()      # every World that has not been closed (close_all_worlds)


def lib():
    """Load libavbd_hip.so; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AvbdError(f"{LIB_PATH} is missing: run `make -C avbd-demo3d_amd/csrc` (no CPU fallback exists)")
    L = C.CDLL(LIB_PATH)
    L.avbd_last_error.restype = C.c_char_p
    L.avbd_last_error.argtypes = [C.c_void_p]
    L.avbd_stream.restype = C.c_void_p
    L.avbd_stream.argtypes = [C.c_void_p]
    L.avbd_option_key.restype = C.c_char_p
    L.avbd_option_key.argtypes = [C.c_int]
    for name in EXPORTS:
        if name in ("avbd_last_error", "avbd_stream", "avbd_option_key"):
            continue
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def option_keys():
    """the keys avbd_set_option accepts"""
    L = lib()
    return [L.avbd_option_key(i).decode() for i in range(L.avbd_option_count())]


def partition_by_axis(position, inv_mass, nranks, axis=0):
    """Equal-count slabs of the dynamic bodies along one axis (pure host code, no GPU needed)."""
    position = np.ascontiguousarray(position, dtype=np.float32).reshape(-1, 3)
    inv_mass = np.ascontiguousarray(inv_mass, dtype=np.float32)
    owner = np.zeros(len(inv_mass), np.int32)
    rc = lib().avbd_partition_by_axis(len(inv_mass), _fp(position), _fp(inv_mass), int(nranks), int(axis), owner.ctypes.data_as(C.POINTER(C.c_int)))
    if rc:
        raise AvbdError(f"avbd_partition_by_axis failed: {rc}")
    return owner


def exchange_layout(owner, nranks):
    """Layout of the per-iteration pose exchange (pure host code of the library, no GPU needed): (rankStart, perm, row, maxCount)."""
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    n = len(owner)
    rank_start = np.zeros(nranks + 1, np.int32); perm = np.zeros(max(n, 1), np.int32); row = np.zeros(max(n, 1), np.int32)
    most = C.c_int(0)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    rc = lib().avbd_exchange_layout(n, ip(owner), int(nranks), ip(rank_start), ip(perm), ip(row), C.byref(most))
    if rc:
        raise AvbdError(f"avbd_exchange_layout failed: {rc}")
    return rank_start, perm[:int(rank_start[-1])], row[:n], most.value


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().avbd_comm_unique_id(buf)
    if rc:
        raise AvbdError("RCCL is not available (avbd_comm_unique_id failed)")
    return buf.raw


def close_all_worlds():
    """Destroy every context this process still holds.  A context owns two HIP streams, and how many contexts are alive on a device when another
    one is made decides whether its streams get hardware queues of their own (avbd_group_step's persistent all-gather form asks for that): a test
    session closes what a test left behind — a failed test's locals stay alive in its traceback — instead of letting it change later tests."""
    n = 0
    for w in list(_live_worlds):
        try:
            if getattr(w, "h", None):
                w.close(); n += 1
        except Exception:
            pass
    return n


def group_step(worlds, nsteps=1):
    arr = (C.c_void_p * len(worlds))(*[w.h for w in worlds])
    rc = lib().avbd_group_step(arr, len(worlds), int(nsteps))
    if rc:
        raise AvbdError(f"avbd_group_step failed: {rc}: " + "; ".join(lib().avbd_last_error(w.h).decode() for w in worlds))


def group_rebalance(worlds, owner):
    """new ownership for the ranks of one group_step call, after any number of steps (the manifolds move with the bodies)"""
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    arr = (C.c_void_p * len(worlds))(*[w.h for w in worlds])
    rc = lib().avbd_group_rebalance(arr, len(worlds), owner.ctypes.data_as(C.POINTER(C.c_int)), len(owner))
    if rc:
        raise AvbdError(f"avbd_group_rebalance failed: {rc}: " + "; ".join(lib().avbd_last_error(w.h).decode() for w in worlds))


class World:
    """One solver context on one MI355X: the `Solver` object of solver.h:146-181 behind the C ABI."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.avbd_create(int(device), C.byref(h))
        if rc != 0:
            raise AvbdError(f"avbd_create(device={device}) failed with code {rc} ({self.L.avbd_last_error(None).decode()}): no usable gfx950 "
                            "device / HIP runtime (this framework has no CPU fallback)")
        self.h = h
        _live_worlds.add(self)

    # -- lifetime
    def close(self):
        if getattr(self, "h", None):
            self.L.avbd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise AvbdError(f"avbd error {rc}: {self.L.avbd_last_error(self.h).decode()}")
        return rc

    # -- params
    def params(self):
        p = np.zeros(1, PARAMS_DT)
        self._chk(self.L.avbd_get_params(self.h, p.ctypes.data_as(C.c_void_p)))
        return p[0]

    def set_params(self, **kw):
        p = np.zeros(1, PARAMS_DT)
        self._chk(self.L.avbd_get_params(self.h, p.ctypes.data_as(C.c_void_p)))
        for k, v in kw.items():
            p[0][k] = v
        self._chk(self.L.avbd_set_params(self.h, p.ctypes.data_as(C.c_void_p)))

    def clear(self):
        self._chk(self.L.avbd_clear(self.h))

    def scene(self, name):
        """Build one of the reference scenes (scenes.h) or StressN:<n> / Packed:<n> through the host mirror."""
        rc = self.L.avbd_build_scene(self.h, name.encode())
        if rc < 0:
            raise AvbdError(f"unknown scene {name!r}")
        return rc

    # -- construction
    def add_bodies(self, size, density, friction, position, orientation=None, linear_velocity=None, angular_velocity=None):
        size = _f32(size, (-1, 3))
        n = size.shape[0]
        density = _f32(np.broadcast_to(np.asarray(density, np.float32), (n,)))
        friction = _f32(np.broadcast_to(np.asarray(friction, np.float32), (n,)))
        position = _f32(position, (n, 3))
        orientation = _f32(orientation, (n, 4))
        lv = _f32(linear_velocity, (n, 3))
        av = _f32(angular_velocity, (n, 3))
        first = C.c_int(-1)
        self._chk(self.L.avbd_add_bodies(self.h, n, _fp(size), _fp(density), _fp(friction), _fp(position), _fp(orientation),
                                         _fp(lv), _fp(av), C.byref(first)))
        return first.value

    def add_body(self, size, density, friction, pos, quat=(0, 0, 0, 1), lin=(0, 0, 0), ang=(0, 0, 0)):
        return self.add_bodies([size], [density], [friction], [pos], [quat], [lin], [ang])

    def add_joint(self, a, b, rA, rB=(0, 0, 0), lin_stiff=FLT_MAX, ang_stiff=FLT_MAX):
        self._chk(self.L.avbd_add_joint(self.h, int(a), int(b), _fp(_f32(rA)), _fp(_f32(rB)), C.c_float(lin_stiff), C.c_float(ang_stiff)))

    def add_spring(self, a, b, rA, rB, stiffness, rest=-1.0):
        self._chk(self.L.avbd_add_spring(self.h, int(a), int(b), _fp(_f32(rA)), _fp(_f32(rB)), C.c_float(stiffness), C.c_float(rest)))

    def add_ignore(self, a, b):
        self._chk(self.L.avbd_add_ignore(self.h, int(a), int(b)))

    # -- stepping
    def step(self, n=1):
        self._chk(self.L.avbd_step(self.h, int(n)))

    def synchronize(self):
        self._chk(self.L.avbd_synchronize(self.h))

    def remove_body(self, index):
        self._chk(self.L.avbd_remove_body(self.h, int(index)))

    def remove_force(self, list_index):
        self._chk(self.L.avbd_remove_force(self.h, int(list_index)))

    WARNING_LABELS = ("linear velocity", "angular velocity", "predicted position", "predicted orientation", "position", "orientation")

    def warnings(self, capacity=128):
        """Non-finite resets of the last step in the reference's print order -> (count, [(body, kind, (x, y, z)), ...])."""
        dt = np.dtype([("body", "<i4"), ("kind", "<i4"), ("value", "<f4", 3)])
        buf = np.zeros(capacity, dt)
        n = self.L.avbd_get_warnings(self.h, buf.ctypes.data_as(C.c_void_p), capacity)
        if n < 0:
            self._chk(n)
        return n, [(int(r["body"]), int(r["kind"]), tuple(float(x) for x in r["value"])) for r in buf[:min(n, capacity)]]

    def pick(self, origins, directions):
        """Solver::pick (solver.cpp:145-228) for a batch of rays → (body indices, -1 = miss; local hit points)."""
        o = _f32(origins, (-1, 3)); d = _f32(directions, (-1, 3))
        if o.shape != d.shape:
            raise ValueError("origins and directions must have the same shape")
        n = o.shape[0]
        bodies = np.full(n, -1, np.int32); local = np.zeros((n, 3), np.float32)
        self._chk(self.L.avbd_pick(self.h, n, _fp(o), _fp(d), bodies.ctypes.data_as(C.POINTER(C.c_int)), _fp(local)))
        return bodies, local

    def save_snapshot(self, path):
        self._chk(self.L.avbd_save_snapshot(self.h, str(path).encode()))

    def load_snapshot(self, path):
        self._chk(self.L.avbd_load_snapshot(self.h, str(path).encode()))

    def step_index(self):
        return self.L.avbd_step_index(self.h)

    def diag(self):
        d = np.zeros(1, DIAG_DT)
        self._chk(self.L.avbd_get_diagnostics(self.h, d.ctypes.data_as(C.c_void_p)))
        return d[0]

    def stats(self):
        s = np.zeros(1, STATS_DT)
        self._chk(self.L.avbd_get_step_stats(self.h, s.ctypes.data_as(C.c_void_p)))
        return s[0]

    # -- state
    def num_bodies(self):
        return self.L.avbd_num_bodies(self.h)

    def bodies(self):
        n = self.num_bodies()
        out = np.zeros(n, BODY_DT)
        self._chk(self.L.avbd_download_bodies(self.h, out.ctypes.data_as(C.c_void_p), n))
        return out

    def state13(self):
        n = self.num_bodies()
        out = np.zeros((n, 13), np.float32)
        self._chk(self.L.avbd_download_state13(self.h, _fp(out), n))
        return out

    def set_state13(self, s):
        s = _f32(s, (-1, 13))
        self._chk(self.L.avbd_upload_state13(self.h, _fp(s), s.shape[0]))

    def set_joint_angular_sign(self, corrected):
        self._chk(self.L.avbd_set_joint_angular_sign(self.h, int(corrected)))

    def set_shard_mode(self, mode):
        self._chk(self.L.avbd_set_shard_mode(self.h, int(mode)))

    def fast_plan_info(self):
        """the plan of the last fast step: colours, tiles, veto bits, colouring rounds, swept bodies"""
        out = (C.c_int * 8)()
        self._chk(lib().avbd_fast_plan_info(self.h, out))
        return dict(colors=out[0], tiles=out[1], bad=out[2], rounds=out[3], swept=out[4])

    def fast_colors(self):
        """colour of every body in the last fast step's plan (-2: not swept)"""
        out = np.zeros(max(1, self.num_bodies()), np.int32)
        n = lib().avbd_fast_colors(self.h, out.ctypes.data_as(C.POINTER(C.c_int)), len(out))
        if n < 0:
            self._chk(n)
        return out[:n]

    def set_option(self, key, value):
        self._chk(self.L.avbd_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int(0)
        self._chk(self.L.avbd_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    def exact_window_export(self):
        """this rank's record window for exact cuts between processes: the 64 opaque bytes of its hipIpcMemHandle_t"""
        h = (C.c_ubyte * 64)()
        n = C.c_ulonglong(0)
        self._chk(self.L.avbd_exact_window_export(self.h, h, C.byref(n)))
        return bytes(h)

    def rebalance_export(self):
        """this rank's force list (newest first) and creation keys, for a re-balance between ranks that are processes"""
        n = self._chk(self.L.avbd_rebalance_export(self.h, None, None, 0))
        forces, keys = np.zeros(max(n, 1), FORCE_DT), np.zeros(max(n, 1), np.uint64)
        got = self._chk(self.L.avbd_rebalance_export(self.h, forces.ctypes.data_as(C.c_void_p), keys.ctypes.data_as(C.c_void_p), max(n, 1)))
        return forces[:got], keys[:got]

    def rebalance_import(self, owner, exports):
        """exports: [(forces, keys)] of ALL ranks, index = rank (every rank passes the same list and the same owner[])"""
        owner = np.ascontiguousarray(owner, dtype=np.int32)
        forces = np.ascontiguousarray(np.concatenate([np.asarray(f, dtype=FORCE_DT) for f, _ in exports]), dtype=FORCE_DT)
        keys = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=np.uint64) for _, k in exports]), dtype=np.uint64)
        frm = np.ascontiguousarray(np.concatenate([np.full(len(k), r, np.int32) for r, (_, k) in enumerate(exports)]), dtype=np.int32)
        self._chk(self.L.avbd_rebalance_import(self.h, owner.ctypes.data_as(C.POINTER(C.c_int)), len(owner), forces.ctypes.data_as(C.c_void_p),
                                               keys.ctypes.data_as(C.c_void_p), frm.ctypes.data_as(C.c_void_p), len(keys)))

    def exact_window_attach(self, handles, ranks_on_this_device=1):
        blob = b"".join(handles)
        assert len(blob) == 64 * len(handles)
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        self._chk(self.L.avbd_exact_window_attach(self.h, len(handles), buf, int(ranks_on_this_device)))

    def set_exchange_floats(self, floats_per_body):
        self._chk(self.L.avbd_set_exchange_floats(self.h, int(floats_per_body)))

    def exchange_mismatches(self):
        """(32-bit words of gathered velocities that differed from the re-derived ones, velocity rows received)"""
        w, rows = C.c_uint(0), C.c_uint(0)
        self._chk(self.L.avbd_exchange_mismatches(self.h, C.byref(w), C.byref(rows)))
        return int(w.value), int(rows.value)

    def set_step_index(self, s):
        self._chk(self.L.avbd_set_step_index(self.h, int(s)))

    def set_bodies(self, arr):
        arr = np.ascontiguousarray(arr, dtype=BODY_DT)
        self._chk(self.L.avbd_upload_bodies(self.h, arr.ctypes.data_as(C.c_void_p), len(arr)))

    def forces(self):
        n = self._chk(self.L.avbd_num_forces(self.h))
        out = np.zeros(n, FORCE_DT)
        if n:
            self._chk(self.L.avbd_download_forces(self.h, out.ctypes.data_as(C.c_void_p), n))
        return out

    def set_forces(self, arr):
        arr = np.ascontiguousarray(arr, dtype=FORCE_DT)
        self._chk(self.L.avbd_upload_forces(self.h, arr.ctypes.data_as(C.c_void_p), len(arr)))

    # -- row API (solver.h:102-108) evaluated on the device
    def force_row_count(self, i):
        r = C.c_int(0)
        self._chk(self.L.avbd_force_row_count(self.h, int(i), C.byref(r)))
        return r.value

    def force_compute_constraint(self, i, alpha):
        Cc, lo, hi, lam = (np.zeros(12, np.float32) for _ in range(4))
        self._chk(self.L.avbd_force_compute_constraint(self.h, int(i), C.c_float(alpha), _fp(Cc), _fp(lo), _fp(hi), _fp(lam)))
        return Cc, lo, hi, lam

    def force_compute_derivatives(self, i, body, row):
        Jl, Ja = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._chk(self.L.avbd_force_compute_derivatives(self.h, int(i), int(body), int(row), _fp(Jl), _fp(Ja)))
        return Jl, Ja

    # -- KATs
    def kat_collide(self, pA, qA, sA, pB, qB, sB):
        pA, qA, sA, pB, qB, sB = _f32(pA, (-1, 3)), _f32(qA, (-1, 4)), _f32(sA, (-1, 3)), _f32(pB, (-1, 3)), _f32(qB, (-1, 4)), _f32(sB, (-1, 3))
        n = pA.shape[0]
        out = np.zeros((n, 4), CONTACT_DT)
        cnt = np.zeros(n, np.int32)
        self._chk(self.L.avbd_kat_collide(self.h, n, _fp(pA), _fp(qA), _fp(sA), _fp(pB), _fp(qB), _fp(sB), out.ctypes.data_as(C.c_void_p),
                                          cnt.ctypes.data_as(C.POINTER(C.c_int))))
        return out, cnt

    def kat_solve6(self, A, b):
        A, b = _f32(A, (-1, 36)), _f32(b, (-1, 6))
        x = np.zeros_like(b)
        self._chk(self.L.avbd_kat_solve6(self.h, A.shape[0], _fp(A), _fp(b), _fp(x)))
        return x

    def kat_arith(self, a, b):
        a, b = _f32(a), _f32(b)
        outs = [np.zeros_like(a) for _ in range(4)]
        self._chk(self.L.avbd_kat_arith(self.h, a.size, _fp(a), _fp(b), *[_fp(o) for o in outs]))
        return outs

    # -- sharding
    def set_ownership(self, nranks, rank, owner):
        owner = np.ascontiguousarray(owner, dtype=np.int32)
        self._chk(self.L.avbd_set_ownership(self.h, int(nranks), int(rank), owner.ctypes.data_as(C.POINTER(C.c_int)), len(owner)))

    def comm_init(self, nranks, rank, uid):
        self._chk(self.L.avbd_comm_init(self.h, int(nranks), int(rank), uid))

    # -- measurement
    def profile_enable(self, on=True):
        self._chk(self.L.avbd_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self.L.avbd_profile_reset(self.h))

    def profile(self):
        p = np.zeros(1, PROFILE_DT)
        self._chk(self.L.avbd_profile_get(self.h, p.ctypes.data_as(C.c_void_p)))
        return p[0]

    def stream(self):
        return self.L.avbd_stream(self.h)
