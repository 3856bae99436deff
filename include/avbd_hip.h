/*
 * avbd_hip.h — C ABI of libavbd_hip.so: the MI355X (gfx950) implementation of the AVBD
 * constraint-solve hot path of alxspiker/avbd-demo3d.
 *
 * The reference has no FFI: its boundary is the C++ object API of source/solver.h.  Each entry
 * point below names the reference interface it replaces (file:line under /root/reference/source);
 * the C++17 host mirror in avbd-demo3d_amd/host/ (Solver / Rigid / Force / Joint / Spring /
 * IgnoreCollision, scenes, headless CLI) is a thin veneer over exactly these calls, and
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only; every call returns 0 on success or a negative
 * AVBD_E_* code and never throws; `avbd_last_error` gives the message.  A context owns one device,
 * one HIP stream and all device memory; calls on one context are not thread-safe.  Host buffers
 * are caller-owned and only touched during the call.  Bodies are addressed by creation index
 * (0 = first body added) — the reference's intrusive list order is the reverse (rigid.cpp:20-21).
 * Quaternions are (x, y, z, w) like maths.h:55.  All floating point is IEEE binary32 and results
 * are bit-identical to the reference CPU solver.
 */
#ifndef AVBD_HIP_H
#define AVBD_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define AVBD_ABI_VERSION 5   /* 5: the fast sweep (option "sweep", AvbdStepStats.schedule 5, avbd_fast_plan_info); 4: AvbdStepStats.cutMode, avbd_set_shard_mode 2, avbd_set_option / avbd_get_option; 3: avbd_set_exchange_floats, avbd_exchange_mismatches, avbd_set_shard_mode, avbd_device_count, avbd_set_joint_angular_sign; 2: AvbdStepStats.schedule, snapshots, removal, set_step_index (all additions; 1 and 2 are subsets) */

enum {
    AVBD_OK = 0,
    AVBD_E_INVALID = -1,   /* bad argument */
    AVBD_E_HIP = -2,       /* HIP runtime error (message in avbd_last_error) */
    AVBD_E_CAPACITY = -3,  /* a device-side pool overflowed (candidate pairs / manifolds) */
    AVBD_E_NODEVICE = -4,  /* no usable gfx950 device */
    AVBD_E_COMM = -5       /* RCCL error */
};

typedef struct AvbdContext AvbdContext;

/* Solver public parameters — solver.h:147-151, defaults solver.cpp:240-248. */
typedef struct AvbdParams {
    float dt;
    float gravity[3];
    int   iterations;
    float alpha, beta, gamma;
    int   postStabilize;
} AvbdParams;

/* Solver::Diagnostics — solver.h:155-164 (same field order). */
typedef struct AvbdDiagnostics {
    float maxPenetration;
    float maxConstraintViolation;
    float maxLinearSpeed;
    float maxAngularSpeed;
    float maxNormalImpulse;
    int   activeContacts;
    int   activeManifolds;
    int   dynamicBodies;
} AvbdDiagnostics;

/* Full per-body record — the data members of struct Rigid, solver.h:56-70 (40 floats). */
typedef struct AvbdBody {
    float position[3];
    float orientation[4];
    float linearVelocity[3];
    float angularVelocity[3];
    float prevLinearVelocity[3];
    float prevAngularVelocity[3];
    float initialPosition[3];
    float initialOrientation[4];
    float inertialPosition[3];
    float inertialOrientation[4];
    float size[3];
    float mass, invMass, friction, radius;
} AvbdBody;

/* Manifold::Contact — solver.h:120-130. */
typedef struct AvbdContact {
    int   feature;
    float rA[3], rB[3], normal[3];
    float penetration, C0_n, C0_t[3];
    int   stick;
} AvbdContact;

enum { AVBD_FORCE_MANIFOLD = 0, AVBD_FORCE_JOINT = 1, AVBD_FORCE_SPRING = 2, AVBD_FORCE_IGNORE = 3,
       AVBD_FORCE_HOST = 4 /* user-defined subclass: rows evaluated by the caller, see avbd_add_host_force */ };

/* One Force with its row arrays — struct Force solver.h:85-97 plus the Manifold / Joint / Spring
 * payloads (solver.h:112-143, joint.h:16-24, spring.h:15-25).  bodyA = -1 means the world. */
typedef struct AvbdForce {
    int   kind;
    int   bodyA, bodyB;
    int   rows;
    float C[12], fmin[12], fmax[12], lambda[12], penalty[12], motor[12], stiffness[12];
    int   numContacts;
    float combinedFriction;
    AvbdContact contacts[4];
    float rA[3], rB[3], q0[4], restLength;
} AvbdForce;

/* Live scene statistics of the last completed step; feeds the algorithmic-byte model of
 * SURVEY.md §8d (B_iter = 100*Nb + 124*Ninc + 160*Nc) and the level schedule. */
typedef struct AvbdStepStats {
    int numBodies;          /* all bodies, static included */
    int numDynamic;         /* Nb */
    int numManifolds;       /* Nm */
    int numContacts;        /* Nc */
    int numIncidences;      /* Ninc = sum over contacts of dynamic endpoints */
    int numCandidates;      /* sphere-overlap pairs tested by SAT this step that were not persistent */
    int numLevels;          /* Gauss-Seidel dependency levels of the primal sweep (AVBD_MODE=levels or AVBD_ORDER=levels; the
                               data-flow schedules sweep in descending creation index and report 1) */
    int numJoints, numSprings, numIgnores;
    int warnings;           /* non-finite sanitizer hits this step (solver.cpp:51-66) */
    int schedule;           /* how the iteration loop of that step ran: 0 one launch per level, 1 data-flow tasks (k_iterate),
                               2 resident (one wavefront per body for the whole loop, avbd_resident.h),
                               3 resident with several bodies per wavefront (more swept bodies than resident waves),
                               4 group schedule (several bodies in flight per wavefront, avbd_groups.h),
                               5 fast sweep (option "sweep" = 1: multi-colour Gauss-Seidel, avbd_fast.hip) */
    int cutMode;            /* how the ranks of a sharded step were coupled across their cuts (ABI 4): 0 not sharded, 1 all-gather, one launch
                               per iteration, 2 all-gather beside ONE persistent launch (both: Jacobi across the cut), 3 exact cuts
                               (bit-identical to one GPU).  What RAN, whatever avbd_set_shard_mode asked for */
} AvbdStepStats;

/* Accumulated device time per kernel family since the last avbd_profile_reset, measured with HIP
 * events on the context's stream (only while profiling is enabled: it serialises the stream). */
typedef struct AvbdProfile {
    double ms_broadphase, ms_narrowphase, ms_topology, ms_predict, ms_primal, ms_dual, ms_velocity;
    long long launches_primal, launches_dual, launches_total;
    long long steps;
    double algorithmic_bytes_primal, algorithmic_bytes_dual; /* summed per launch, SURVEY §8d model */
} AvbdProfile;

/* Threading / concurrency: one context = one device + one stream; calls on a context are not thread-safe.  The iteration
 * kernel needs all its workgroups resident on the device, so two contexts that share a device must not be stepped
 * CONCURRENTLY from different host threads (stepping them one after the other, or through avbd_group_step, is fine):
 * two half-resident launches would wait for each other until the bounded spin (4 s) gives up and the step returns AVBD_E_HIP ("dependency wait timed out"). */

/* ---- lifetime: `new Solver` / `delete solver` (solver.cpp:127-143) ---------------------- */
int  avbd_create(int device, AvbdContext** out);
int  avbd_destroy(AvbdContext* ctx);
const char* avbd_last_error(const AvbdContext* ctx);
int  avbd_abi_version(void);

/* ---- Solver fields and Solver::defaultParams / clear (solver.cpp:230-253) ---------------- */
int  avbd_default_params(AvbdParams* out);
int  avbd_set_params(AvbdContext* ctx, const AvbdParams* p);
int  avbd_get_params(const AvbdContext* ctx, AvbdParams* out);
int  avbd_clear(AvbdContext* ctx);

/* ---- `new Rigid(solver, size, density, friction, pos, orient, linVel, angVel)` (rigid.cpp:12-41).
 * Adds n bodies in order; arrays are packed (size: 3n, density: n, friction: n, position: 3n,
 * orientation: 4n or NULL for identity, linearVelocity / angularVelocity: 3n or NULL for zero).
 * Returns the creation index of the first added body in *first_index (may be NULL). */
int  avbd_add_bodies(AvbdContext* ctx, int n, const float* size, const float* density, const float* friction,
                     const float* position, const float* orientation, const float* linearVelocity,
                     const float* angularVelocity, int* first_index);

/* ---- `new Joint(...)` both constructors (joint.cpp:11-61): bodyA = -1 selects the world-anchor
 * form where rA is the world-space anchor and rB is ignored. ---- */
int  avbd_add_joint(AvbdContext* ctx, int bodyA, int bodyB, const float rA[3], const float rB[3],
                    float linearStiffness, float angularStiffness);
/* ---- `new Spring(...)` (spring.cpp:10-31); rest < 0 means "use the current distance". ---- */
int  avbd_add_spring(AvbdContext* ctx, int bodyA, int bodyB, const float rA[3], const float rB[3],
                     float stiffness, float rest);
/* ---- `new IgnoreCollision(...)` (ignorecollision.h:15-16). ---- */
int  avbd_add_ignore(AvbdContext* ctx, int bodyA, int bodyB);

/* ---- generic form of the three constructors above: append one host-created Joint / Spring /
 * IgnoreCollision from an explicit record (kind, bodyA/bodyB, rA, rB, q0, restLength and the row
 * arrays lambda / penalty / stiffness / motor / fmin / fmax as the reference constructors leave them:
 * force.cpp:32-40, joint.cpp:22-38, spring.cpp:16-30).  Used by the C++ host mirror, which evaluates
 * the constructor arithmetic itself at construction time like the reference does. ---- */
int  avbd_add_force(AvbdContext* ctx, const AvbdForce* record);

/* ---- scene builders of scenes.h:23-212 by exact name ("Stack", "Stress1000", ...), plus the build's
 * "StressN:<n>" / "Packed:<n>" / "StressGrid:<x>,<y>,<z>" / "PackedGrid:...", each optionally followed by "+welds:<k>" (BASELINE configs[4]:
 * weld Joints between the x-adjacent cubes 2j+1 | 2j+2 of the first k pairs); clears the context first like every reference scene does.  Returns
 * the scene index (>= 0) or AVBD_E_INVALID for an unknown name. ---- */
int  avbd_build_scene(AvbdContext* ctx, const char* name);

/* ---- Solver::step() (solver.cpp:255-514), nsteps times back to back on the device.
 * Several contexts on ONE device (two Solvers in one program): their avbd_step / avbd_group_step calls take turns on the device —
 * a call is ordered behind the call of another context that used the device last (one event per device, no host wait); the
 * iteration loop of a step needs the whole device.  One context per device never waits. ---- */
int  avbd_step(AvbdContext* ctx, int nsteps);
int  avbd_synchronize(AvbdContext* ctx);
int  avbd_step_index(const AvbdContext* ctx);                       /* Solver::stepIndex */
int  avbd_get_diagnostics(AvbdContext* ctx, AvbdDiagnostics* out);  /* Solver::lastDiagnostics */
int  avbd_get_step_stats(AvbdContext* ctx, AvbdStepStats* out);

/* ---- state access (what main.cpp:228-234 reads from Rigid every step) ---- */
int  avbd_num_bodies(const AvbdContext* ctx);
int  avbd_download_bodies(AvbdContext* ctx, AvbdBody* out, int capacity);
/* 13-float rows: position[3], orientation[4], linearVelocity[3], angularVelocity[3] */
int  avbd_download_state13(AvbdContext* ctx, float* out, int capacity_bodies);
int  avbd_upload_state13(AvbdContext* ctx, const float* in, int n_bodies);
/* full dynamic state incl. prev velocities (teacher forcing / snapshots) */
int  avbd_upload_bodies(AvbdContext* ctx, const AvbdBody* in, int n_bodies);

/* ---- user-defined Force subclasses (solver.h:85-109): rows evaluated by the caller, between launches -----------------------------
 * A Force whose getRowCount / initialize / computeConstraint / computeDerivatives are the caller's own code cannot run on the
 * device.  It is registered with avbd_add_host_force; while such a force exists a step runs one launch per Gauss-Seidel level
 * with the host in the loop (exact, slow): before a level that visits one of its endpoints, and in every dual pass, the library
 * calls `fn` with the current poses of the two endpoints and the phase, and the callee fills / reads `rows`:
 *   AVBD_HOST_INITIALIZE  step begin (solver.cpp:273-296): call initialize(); return rows (rows = 0: the force sits this step out) and the row
 *                         state — or rows = -1 when initialize() returned false: the force is DELETED as solver.cpp:274-279 does, the
 *                         ordinals of the host forces created after it move down by one (the callee drops its entry in the same
 *                         call) and no other phase is called for it; otherwise the library applies the warm-start decay of solver.cpp:281-293 and hands lambda / penalty back with
 *   AVBD_HOST_STORE       copy rows->lambda / rows->penalty into the object
 *   AVBD_HOST_PRIMAL      visit of `body` (solver.cpp:371-399): call computeConstraint(alpha), then computeDerivatives(Jl, Ja, body, row)
 *                         for every row into rows->JlA / JaA (body is endpoint A) or JlB / JaB; return C, fmin, fmax, lambda, penalty,
 *                         stiffness, motor as the call left them
 *   AVBD_HOST_DUAL        dual pass (solver.cpp:411-430): call computeConstraint(alpha) and computeDerivatives for BOTH endpoints; the library
 *                         does the lambda / penalty update (rowPenaltyGain, solver.cpp:94-125) and hands the result back with AVBD_HOST_STORE
 * A non-zero return value aborts the step with AVBD_E_INVALID. */
enum { AVBD_HOST_INITIALIZE = 0, AVBD_HOST_STORE = 1, AVBD_HOST_PRIMAL = 2, AVBD_HOST_DUAL = 3 };
typedef struct AvbdHostRows {
    int   rows;
    float C[12], fmin[12], fmax[12], lambda[12], penalty[12], stiffness[12], motor[12];
    float JlA[12][3], JaA[12][3], JlB[12][3], JaB[12][3];
} AvbdHostRows;
typedef int (*AvbdHostForceFn)(void* user, int host_force, int phase, float alpha, int body,
                               const float poseA[7] /* position, orientation; NULL for the world */, const float poseB[7], AvbdHostRows* rows);
int  avbd_add_host_force(AvbdContext* ctx, int bodyA /* -1 = world */, int bodyB, int* host_force /* out: ordinal passed to the callback */);
int  avbd_set_host_force_callback(AvbdContext* ctx, AvbdHostForceFn fn, void* user);

/* ---- force list in the reference's solver-list order: newest first (force.cpp:16-18) ---- */
int  avbd_num_forces(AvbdContext* ctx);
int  avbd_download_forces(AvbdContext* ctx, AvbdForce* out, int capacity);
/* one record of that list (what a host mirror of Force reads back after a row-API call: solver.h:85-109 members) */
int  avbd_download_force(AvbdContext* ctx, int list_index, AvbdForce* out);
/* replace every force by `n` records given newest first (snapshot restore / teacher forcing) */
int  avbd_upload_forces(AvbdContext* ctx, const AvbdForce* in, int n);

/* ---- snapshots (the reference has no checkpoint/resume) ------------------------------------
 * One little-endian file: header (magic, record sizes, counts, Solver::stepIndex, parameters) + the AvbdBody and AvbdForce
 * arrays of avbd_download_bodies / avbd_download_forces.  avbd_load_snapshot replaces the whole content of the context;
 * stepping on continues bit for bit like the context that saved.  CLI: --save-state <file>, --load-state <file>. */
int  avbd_save_snapshot(AvbdContext* ctx, const char* path);
int  avbd_load_snapshot(AvbdContext* ctx, const char* path);
int  avbd_set_step_index(AvbdContext* ctx, int stepIndex);

/* ---- removal: `delete body` (rigid.cpp:43-49) / `delete force` (force.cpp:31-69) ------------------------------
 * The device state is compact and ordered by creation, so a removal rebuilds it without the element (cost: one download and
 * one upload of the whole state — an editing operation, not a per-step one).  Removing a body removes the forces attached to
 * it as well (the reference leaves them dangling); later-created bodies move down one index.  `list_index` counts the force
 * list newest first, as avbd_download_forces returns it. */
int  avbd_remove_body(AvbdContext* ctx, int body_index);
int  avbd_remove_force(AvbdContext* ctx, int list_index);

/* ---- row API on the device for one force of the downloaded list (index into newest-first order):
 * Force::computeConstraint / computeDerivatives (solver.h:104-106).  compute_constraint applies the
 * reference's in-place mutations to the device-resident force and returns its rows. ---- */
int  avbd_force_row_count(AvbdContext* ctx, int force_index, int* rows);
int  avbd_force_compute_constraint(AvbdContext* ctx, int force_index, float alpha, float* C, float* fmin, float* fmax, float* lambda);
int  avbd_force_compute_derivatives(AvbdContext* ctx, int force_index, int body, int row, float J_linear[3], float J_angular[3]);

/* ---- the warnings of the last step (solver.cpp:51-66: "[Physics] Warning: body %d produced non-finite %s ...; resetting").
 * The kernels reset the quantity exactly like the reference and log what it would have printed; this call returns the
 * number of resets of the last step (>= 0) and fills up to `capacity` records (at most the first 128 are kept) in the order
 * the reference's sequential loops print them.  `kind`: 0 linear velocity, 1 angular velocity, 2 predicted position,
 * 3 predicted orientation, 4 position, 5 orientation; `value` is the offending vector (xyz of a quaternion).  Inputs that
 * are NaN keep their sign; a NaN GENERATED by an operation is positive on gfx950 and negative on x86, so printf("%.3f")
 * may show `nan` where the reference shows `-nan` — the only character of the text that can differ. ---- */
typedef struct AvbdWarning { int body; int kind; float value[3]; } AvbdWarning;
int  avbd_get_warnings(AvbdContext* ctx, AvbdWarning* out, int capacity);

/* ---- Solver::pick (solver.cpp:145-228): for each of `nrays` rays (origin xyz, direction xyz — any non-zero length),
 * the creation index of the nearest DYNAMIC body whose box the ray enters (ties: the newer body, as the reference's
 * newest-first list walk resolves them) and the hit point in that body's local frame; -1 (and a zero point) for a miss
 * or a direction shorter than 1e-3.  Evaluated on the device against the current body poses, bit-identical with the
 * reference's slab test. ---- */
int  avbd_pick(AvbdContext* ctx, int nrays, const float* origins /* 3*nrays */, const float* directions /* 3*nrays */,
               int* bodies /* nrays */, float* locals /* 3*nrays */);

/* ---- known-answer entry points evaluated on the device (one lane) ---- */
int  avbd_kat_collide(AvbdContext* ctx, int npairs, const float* posA, const float* quatA, const float* sizeA,
                      const float* posB, const float* quatB, const float* sizeB, AvbdContact* out /* 4*npairs */, int* counts);
int  avbd_kat_solve6(AvbdContext* ctx, int n, const float* A /* 36n: ll,la,al,aa column-major */, const float* b /* 6n */, float* x /* 6n */);
int  avbd_kat_arith(AvbdContext* ctx, int n, const float* a, const float* b, float* out_div, float* out_sqrt, float* out_mul, float* out_add);

/* ---- sharding over the GPUs of one node (BASELINE configs[3..4]): one context per GPU ----
 * Dynamic bodies are owned statically by one rank each; every rank keeps ALL body poses, builds the manifolds that
 * touch one of its bodies, sweeps its own bodies exactly (later-created first) against the last gathered poses of
 * the others, and after EVERY solver iteration all ranks all-gather {position, orientation} (RCCL over xGMI, 32 B
 * per body).  Across the cut this is a Jacobi coupling, so sharded results are close to, not bit-identical with,
 * the single-GPU sweep.  There is no such mode in the reference.
 * avbd_set_ownership is accepted while the context has not stepped on its present content: on a freshly built scene, or right after
 * avbd_load_snapshot / the upload calls (every rank loads the same snapshot, cuts the slabs from the restored positions, steps on);
 * once steps have run, ownership moves with avbd_group_rebalance / avbd_rebalance_import. */
int  avbd_partition_by_axis(int n, const float* position /* 3n */, const float* invMass /* n */, int nranks, int axis, int* owner /* n out */);
int  avbd_set_ownership(AvbdContext* ctx, int nranks, int rank, const int* owner, int n);
/* Layout of the per-iteration exchange for an ownership vector; host-only, needs no device (the contract of the
 * exchange is 13 floats per body — position, orientation, velocities; this library exchanges the 8 floats that change inside
 * the iteration loop, {position + radius, orientation}, and re-derives the velocities from the gathered poses exactly as
 * solver.cpp:434-469 does, so both forms carry the same information).  rankStart[nranks + 1] prefix counts, perm[n] swept
 * bodies grouped by rank, row[n] the row of a body in every all-gather buffer (equal blocks of max(1, maxCount) rows per rank;
 * -1 for replicated bodies), maxCount the largest block. */
int  avbd_exchange_layout(int n, const int* owner, int nranks, int* rankStart, int* perm, int* row, int* maxCount);
int  avbd_comm_unique_id(char id[128]);                                   /* ncclGetUniqueId on one rank, broadcast by the caller */
int  avbd_comm_init(AvbdContext* ctx, int nranks, int rank, const char id[128]);  /* ncclCommInitRank; avbd_step then exchanges per iteration */
/* the same sharded schedule for `n` contexts living in ONE process (exchange by device-to-device copies) */
int  avbd_group_step(AvbdContext** ctxs, int n, int nsteps);
/* How the ranks of avbd_group_step couple across a cut.  0 (default; AVBD_SHARD_EXACT=1 changes it): the all-gather schedule above
 * (north_star's form: Jacobi across the cut).  1: EXACT cuts — every rank sweeps its slab with the single-GPU protocol and polls
 * the stamped pose / visit / dual records of the bodies and contacts across a cut exactly as it polls local ones (one address
 * space for the records of all ranks: one allocation for contexts of one process, peer-to-peer windows between GPUs); no exchange
 * during the iteration loop, one copy of the final poses per step, results bit-identical to the single-GPU sweep
 * (tests/test_gpu_exact_cuts.py).  Exact cuts carry contacts and — since round 5 — Joints (a Joint half has one record slot in the space all
 * ranks share; no springs or user-defined forces), at most 8 bodies per sub-group, and the
 * ranks of one avbd_group_step call must share a device.  A step that cannot run them FAILS with AVBD_E_INVALID and the reason in
 * avbd_last_error — a caller that asked for single-GPU bits never gets a Jacobi coupling under that name.
 * 2: exact cuts when the step can run them, the all-gather form otherwise; AvbdStepStats.cutMode tells which one ran. */
/* Ownership re-balanced AFTER the first step, for the ranks of one avbd_group_step call (slabs re-balanced every K steps):
 * the manifolds move with the bodies (contacts, lambda, penalty, list order), every rank continues from the state the single-GPU run would hold.
 * owner[] as for avbd_set_ownership (avbd_partition_by_axis on the current positions gives equal slabs again).  Synchronous, host-side. */
int  avbd_group_rebalance(AvbdContext** ctxs, int n, const int* owner, int nbodies);
/* The same for ranks that are PROCESSES (exact cuts through windows, or the RCCL all-gather form), in two halves with the caller's own gather in between
 * (bench.py / the Python mirror: torch.distributed.all_gather_object; any transport does):
 *   avbd_rebalance_export  this rank's force list, newest first, with its creation keys.  capacity 0: returns the count only.  Returns the count.
 *   avbd_rebalance_import  ALL ranks' exports in any order (fromRank[i] = the rank that exported item i) and the new ownership; keeps the copy of the
 *                          rank that swept each contact's owner, rebuilds this rank under the new ownership.  Every rank must call it with the same
 *                          items and the same owner[]; attached exact-cut windows stay attached (the pools do not grow in a re-balance). */
int  avbd_rebalance_export(AvbdContext* ctx, AvbdForce* forces, unsigned long long* keys, int capacity);
int  avbd_rebalance_import(AvbdContext* ctx, const int* owner, int nbodies, const AvbdForce* forces, const unsigned long long* keys, const int* fromRank, int count);
int  avbd_set_shard_mode(AvbdContext* ctx, int mode);
/* Exact cuts between ranks that are PROCESSES — one per GPU, or several on one device (how they are validated: N processes on one
 * MI355X reproduce the single-GPU sweep bit for bit, tests/test_gpu_exact_processes.py).  Every rank holds the whole scene and has
 * set the same ownership (avbd_set_ownership).  avbd_exact_window_export allocates this rank's record window — its pose records
 * and the visit / dual records of its contacts, cuts included — and returns the window's hipIpcMemHandle_t as 64 opaque bytes;
 * the caller exchanges the handles of all ranks by whatever means it has (bench.py: torch.distributed) and hands the table,
 * rank-major, to avbd_exact_window_attach, which opens the peers' windows (hipIpcOpenMemHandle with lazy peer access) and puts
 * the context into shard mode 1.  From then on avbd_step(ctx, n) on every rank runs exact cuts: a rank writes into its own
 * window and reads a partner's records from the window of the rank that sweeps the partner, with system-scope accesses; nothing
 * is exchanged during the loop, and a one-word handshake per step (a rank's "records of step s complete") lets the ranks copy
 * each other's final poses.  All ranks must call avbd_step with the same step counts.  `ranksOnThisDevice`: how many of the
 * ranks' loop kernels share this context's device (they are resident together, so each takes that share of the wavefronts).
 * The reference has no counterpart (solver.cpp is one thread); the loop being cut is solver.cpp:340-431. */
int  avbd_exact_window_export(AvbdContext* ctx, unsigned char handle[64], unsigned long long* bytes /* may be NULL */);
int  avbd_exact_window_attach(AvbdContext* ctx, int nranks, const unsigned char* handles /* nranks x 64 */, int ranksOnThisDevice);
/* What the ranks exchange.  8 (default): {position + radius, orientation} every iteration, velocities re-derived by
 * every rank from the gathered poses.  13: additionally ONE gather of {linear, angular velocity} per step, after the velocity
 * stage (pos3 + quat4 + linVel3 + angVel3 = north_star's 13 floats per body); the arriving rows overwrite the re-derived ones and
 * must equal them bit for bit — avbd_exchange_mismatches returns the number of 32-bit words that ever differed on this rank
 * (tests/test_gpu_sharded.py keeps it at 0).  Exact cuts (avbd_set_shard_mode 1) exchange nothing during a step and ignore it. */
int  avbd_set_exchange_floats(AvbdContext* ctx, int floatsPerBody);
/* Schedule options (ABI 4).  Everything that chooses HOW the sweep of solver.cpp:340-431 is run — which schedule, which form of a
 * kernel, how many workgroups — is one (key, value) pair; none of them changes a result bit (tests/test_gpu_options.py runs every
 * key against the oracle).  avbd_option_count / avbd_option_key enumerate the keys; DESIGN.md "Options" documents them.  The
 * environment variables AVBD_<KEY IN CAPITALS> are a debug override read when a context is created.  There is no counterpart in the
 * reference (one thread, one schedule). */
int  avbd_set_option(AvbdContext* ctx, const char* key, int value);
int  avbd_get_option(AvbdContext* ctx, const char* key, int* value);
int  avbd_option_count(void);
const char* avbd_option_key(int index);
int  avbd_exchange_mismatches(AvbdContext* ctx, unsigned* words, unsigned* rowsCompared /* may be NULL: velocity rows received so far */);
int  avbd_device_count(void);                                            /* visible gfx950 devices (0 without one) */
/* Joint (joint.cpp:108-139): 0 (default) keeps the reference's angular Jacobian sign — bug-compatible: body-body welds blow up,
 * SURVEY.md §7-5 —, 1 uses J_angular = dC/dtheta (-e_k for bodyA, +e_k for bodyB), with which welds hold.  Applies to the
 * joints that exist and to those added later.  There is no such switch in the reference. */
int  avbd_set_joint_angular_sign(AvbdContext* ctx, int corrected);

/* ---- the fast sweep (ABI 5): avbd_set_option(ctx, "sweep", 1) — CLI `--mode fast` -------------------------------------------------
 * The iteration loop of solver.cpp:340-431 as a multi-colour Gauss-Seidel: the swept bodies are coloured on the device every step, the
 * bodies of one colour are solved together (one lane per contact, the 27 entries of a body's 6x6 system summed over its lanes with
 * DPP butterflies), a force's dual update (solver.cpp:411-430) runs at its first visit of the next iteration.  Row arithmetic is the
 * exact mode's (same expression trees, no contraction); the ORDER of bodies and of sums is not, so results agree with the exact mode
 * within the tolerance DESIGN.md section 2b states, not bit for bit; they are bit-identical from run to run.  Contacts, Joints and Springs are
 * lane entries of the sweep; scenes with user forces, exact cuts and bodies with more than 64 contacts take the exact schedules (AvbdStepStats.schedule says
 * which one ran: 5 = fast sweep).  avbd_fast_plan_info: the plan of the last fast step — out[0] colours, [1] tiles of 64 lanes,
 * [2] veto bits (0: the sweep ran), [3] rounds the colouring took, [4] swept bodies.  Synchronises.  No counterpart in the reference. */
int  avbd_fast_plan_info(AvbdContext* ctx, int out[8]);
/* the colour of every body in the last fast step's plan (-2: not swept); tests check that no two bodies that share a force share a colour */
int  avbd_fast_colors(AvbdContext* ctx, int* out, int capacity_bodies);

/* ---- measurement ---- */
int  avbd_profile_enable(AvbdContext* ctx, int enable);
int  avbd_profile_reset(AvbdContext* ctx);
int  avbd_profile_get(AvbdContext* ctx, AvbdProfile* out);
/* the raw hipStream_t of the context (as void*), so a caller can record its own events on it */
void* avbd_stream(AvbdContext* ctx);
/* debug builds (-DAVBD_DF_STATS): per-wave time breakdown of the last k_iterate launch, 16 counters per wave */
int  avbd_debug_group_task_times(AvbdContext* ctx, unsigned* out /* iterations * bodies * 4 */, int iterations);
int  avbd_debug_iterate_stats(AvbdContext* ctx, unsigned long long* out, int waves);
/* Per-task time stamps of the resident schedule (builds with -DAVBD_DF_STATS only; zeros otherwise): out[it][1024][8] u32. */
int  avbd_debug_task_times(AvbdContext* ctx, unsigned* out, int iterations);
/* phase boundaries of the last k_topology_small launch (s_memrealtime ticks, -DAVBD_DF_STATS builds only) */
int  avbd_debug_topology_times(AvbdContext* ctx, unsigned long long* out7);
/* s_memrealtime at start / after loads / after Manifold::initialize / at the end of the first pair of k_front_small (stats builds) */
int  avbd_debug_front_times(AvbdContext* ctx, unsigned long long* out4);
int  avbd_debug_front_blocks(AvbdContext* ctx, unsigned long long* out /* 16 * blocks */, int blocks);

#ifdef __cplusplus
}
#endif
#endif /* AVBD_HIP_H */
