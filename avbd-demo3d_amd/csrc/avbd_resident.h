/*
 * avbd_resident.h — "resident" schedule of the iteration loop (solver.cpp:340-431) for scenes of contacts and joints (no
 * springs, at most 64 incidence entries per body; a lane holds one contact or one half of a joint).  Three forms of the same
 * code (iterate_resident<MULTI, JOINTS>):
 *   - up to 1024 swept bodies: every body gets ONE wavefront for the whole launch (all iterations), state in registers;
 *   - more bodies: every wave takes every W-th body of the sweep order and parks a body's state between its tasks
 *     (two workgroups per CU, 2048 waves, from 1024 swept bodies; in order, or — from 8 bodies per wave — whichever body
 *     has its inputs published, see X.outOfOrder);
 *   - sharded runs: one launch per iteration; state is parked between the launches the same way.
 * The description below is written for the first form.
 *
 * Same dependencies, same arithmetic and same summation order as k_iterate's task schedule (avbd_iterate.h); what
 * changes is where state lives and how it moves between wavefronts:
 *
 *   - the body's pose, its incidence entries (<= 64 Manifold contacts, one per lane) and the state of the contacts it
 *     "owns" (it is the last visitor of the contact in the sweep, so it also runs the dual update) stay in REGISTERS
 *     from the first iteration to the last; the tables in HBM are read once before and written once after the loop;
 *   - what another wavefront needs travels as self-validating 16-byte records: a record carries the iteration stamp
 *     of its producer in its last word and is written / read with ONE buffer_store/load_dwordx4 ... sc1 by one lane.
 *     A consumer polls the records themselves, so the poll that sees the stamp has the payload in its registers:
 *     no counter, no s_waitcnt drain before publishing, no second round trip for the payload.
 *         pose   : 3 records / body   {p.xyz,s} {q.xyz,s} {q.w,-,-,s}       producer: the body's wave after its solve
 *         visit  : 1 record / contact {lambda1',lambda2',stick',s}          producer: first visitor (later-created body)
 *         dual   : 2 records / contact {lambda.xyz,s} {penalty.xyz,2s+stick} producer: owner, after its dual update
 *     (computeConstraint, manifold.cpp:183-243, mutates lambda[1..2] and stick on EVERY call, so the first visitor's
 *     mutations are an input of the second visit; the stored normal and its tangent memo are mutated too, but only
 *     as a function of themselves: each side replays the other side's re-normalisations locally — see replay_normal.)
 *   - stamps are unique per step (serial * 64 + iteration + 1; the launches of a sharded step share the serial), so the
 *     exchange buffers are never cleared between steps (only when the serial wraps, every 2^24 steps).
 *
 * The single-copy atomicity of one lane's aligned 16-byte access is a property of gfx950's memory pipeline (one
 * request, one 64-byte line), not of the HIP memory model; this file is gfx950-only like the rest of csrc/.
 * A producer can never overwrite a record before its consumer has read it: the next write of every record depends,
 * through the Gauss-Seidel dependencies themselves, on the consumer's own progress.
 */
#pragma once
#include "avbd_primal.h"

#pragma clang fp contract(off)

namespace avbd {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

/* -DAVBD_ASM_MARKS: comments in the generated assembly that tools/kernel_resources.py splits the instruction mix at */
#ifdef AVBD_ASM_MARKS
#define AVBD_MARK(name) asm volatile("; AVBD_MARK " #name ::: "memory")
#else
#define AVBD_MARK(name) do { } while (0)
#endif

struct ResidentExchange {
    float4* pose;    /* [3 * bodyCap] */
    float4* visit;   /* [4 * manifoldCap] */
    float4* dual;    /* [8 * manifoldCap] */
    unsigned poseBytes, visitBytes, dualBytes;
    unsigned base;   /* stamp of "nothing published yet" for this launch */
    int enabled;     /* host-side prerequisites hold (single GPU, no joints/springs, few enough iterations) */
    int maxWaves;    /* wavefronts the launch keeps resident */
    int quietPermille; /* share of the previous iteration's wait a wave spends asleep before it starts polling */
    float4* privN;     /* [3 * incidence capacity] MULTI: a first visitor's own copy of (normal, u1, u2) between its tasks */
    float4* privPose;  /* [2 * bodyCap]            MULTI: the pose between the body's tasks */
    int spread;        /* one body per wave: slot = wave-in-workgroup * workgroups + workgroup instead of the linear wave index */
    int prStride;      /* floats per product line: kResStride, or kResStrideLite when two workgroups share a CU */
    int joints;        /* the scene has Joint forces: instantiate the joint-aware code */
    int jointRecBase;  /* record index of joint half 0 of GForce 0 in `dual` (= 4 * manifold capacity) */
    GForce* gf;        /* group schedule with joints (k_iterate_groups<*, *, true>): the Joint records (the other kernels get them as an argument) */
    int multi;         /* more swept bodies than resident waves: every wave takes every W-th body */
    int forceMulti;    /* launches that cover part of the iteration loop (sharded runs) park state even with one body per wave */
    int planesAlways;  /* sharded runs: other kernels of the step read the pose planes after every iteration */
    float4* send;      /* sharded runs: this rank's rows of the all-gather, filled by the wave that publishes the pose */
    const int* sendSlot; /* [bodyCap] row of a body this rank solves */
    int outOfOrder;    /* MULTI: a wave runs whichever of its bodies has its inputs ready (oldest iteration first) instead of sweeping them in order */
    int patience;      /* out of order, nothing ready: ticks (10 ns) a wave waits on its oldest task, prepared, before it looks at its other bodies again (0: never blocks) */
    int* probe;        /* [incidence capacity] out of order: who an entry waits for (partner | moving << 24 | owned << 25 | hasA << 26), kept from a body's first examination */
    int specRounds;    /* ... only in classes with at most this many bodies per sub-group */
    int specMode;      /* group schedule, several bodies per sub-group: when a sub-group with nothing ready takes its oldest task in hand (0: when its wave has nothing else to do in the pass, 1: always, 2: never) */
    float4 *parkC, *parkS, *parkB;   /* group schedule, several bodies per sub-group: packed records of a parked body (k_iterate_groups::load_rest) */
    const int* rankOf; /* [bodyCap] class-major rank of a swept body in this step's plan (k_group_plan): where its arrivals are counted */
    int poolMode;      /* several bodies per sub-group with arrival counters: the 16 sub-groups of a class-0 workgroup share their bodies (0: every sub-group keeps its own) */
    int poolMinRounds; /* ... from this many bodies per sub-group on */
    int depKeep;       /* arrival counters also in classes that take a task in hand (which they then pick by the fewest missing arrivals) */
    unsigned* depCnt;  /* [3 * bodyCap] group schedule, several bodies per sub-group on one GPU: arrivals at a body — every task that ends adds one to each moving
                        * partner whose NEXT task reads what it has just published (k_group_plan zeroes the words of the swept bodies every step); nullptr: off */
    const int* place;  /* one body per wave: body of wave w (-1: none) instead of the sweep order's slot — chains of bodies that hand their pose to each
                        * other share a workgroup and pass it through LDS (k_place_chains); nullptr: slots of the sweep order */
    int2* crit;        /* [bodyCap] one body per wave, out: {later-created partner whose pose arrived last in the final iterations or -1, ticks (10 ns)
                        * from the launch's start to the body's last publish}: what k_place_chains builds next step's chains from */
};

/* Co-located hand-off (one body per wave): the four waves of a workgroup publish their pose in LDS as well; a lane whose partner's
 * wave sits in the same workgroup polls a flag word there instead of the record in memory (50 ns instead of 0.8 us).
 * Protocol: the producer writes the three records, waits for the LDS writes to complete (s_waitcnt lgkmcnt(0)) and then writes
 * the flag; the LDS executes a CU's instructions in order, so a consumer that has seen the flag reads the records behind it. */
#define AVBD_LDS __attribute__((address_space(3)))   /* explicit LDS pointers: a volatile access through a generic pointer becomes a flat, system-scope one */
struct CoLocal {
    const volatile int AVBD_LDS* bodyOf;    /* [kIterWaves] body of each wave of this workgroup, -1: none */
    volatile v4u AVBD_LDS* rec;             /* [kIterWaves][3] pose records, as in memory */
    volatile unsigned AVBD_LDS* flag;       /* [kIterWaves] stamp of the records' last complete publish */
    int self;                               /* this wave */
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xrsrc(void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }
/* aux 16 = sc1: agent scope, served from / written through to the level all XCDs share */
__device__ __forceinline__ v4u xld(__amdgpu_buffer_rsrc_t r, int byteOff) { return __builtin_amdgcn_raw_buffer_load_b128(r, byteOff, 0, 16); }
__device__ __forceinline__ void xst(__amdgpu_buffer_rsrc_t r, int byteOff, v4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, byteOff, 0, 16); }
__device__ __forceinline__ bool stamp_reached(unsigned have, unsigned need) { return (int)(have - need) >= 0; }
__device__ __forceinline__ unsigned fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float bitsf(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ float bcast0(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)); }

#ifdef AVBD_DF_STATS
/* Debug race checker for the stamped records (stats builds only): "a record is never overwritten before its
 * reader has consumed it".  Every task notes `consumed[body] = it + 1` once its inputs are in registers; a task that is about
 * to overwrite its pose record with iteration `it` checks the readers of the value it destroys (iteration it - 1): an
 * earlier-created partner read it in ITS task it - 1, a later-created partner reads it in its task `it`.  The Gauss-Seidel
 * dependencies imply both; a violation sets bit 5 of Counters.overflow. */
__device__ __forceinline__ void race_note_consumed(int* consumed, int body, int it)
{
    __hip_atomic_store(consumed + body, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void race_check_publish(const int* consumed, Counters* cnt, int body, int other, bool hasMovingPartner, int it)
{
    if (!hasMovingPartner || it < 1) return;
    const int seen = __hip_atomic_load(consumed + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int need = other > body ? it + 1 : it;
    if (seen < need) atomicOr(&cnt->overflow, 32);
}
#endif

/* What one computeConstraint call does to the stored normal and its tangent memo (manifold.cpp:185-186 through
 * contact_basis_memo): used to replay the calls the OTHER visitor / the dual made since this wave last touched the
 * contact.  Once the normal is a fixed point of the normalisation every later call leaves it alone. */
template <bool BITS = false>
__device__ __forceinline__ void replay_normal(ContactRows& c, bool& fixedN)
{
    if (fixedN) return;
    f3 n, t1, t2; bool hit;
    contact_basis_memo<BITS>(c.normal, c.u1, c.u2, n, t1, t2, hit);
    c.normal = n; c.u1 = t1; c.u2 = t2;
    fixedN = hit;
}


/* ---- ordered accumulation from pre-multiplied products -------------------------------------------------------------
 * Per-wave LDS of the resident schedule (floats):
 *   J[row][6]            Jl.xyz Ja.xyz of every row, written by the contact lanes BEFORE the wait
 *   S[8][kResRows]       slot arrays f | penalty | gyro.x | gyro.y | gyro.z | penalty-usable flag | zeros | Manifold-row flag, written AFTER the wait
 *   Pr[27][kResStride]   one line per accumulator entry: Ja*Jb of every row (rhs entries: J alone), written BEFORE the wait
 * so that after the wait an accumulator lane adds (product * penalty [+ gyro]) or (J * f) row by row — the reference's
 * order and association, (Ja*Jb)*penalty (solver.cpp:387-391) — from three 16-byte LDS reads per four rows. */
constexpr int kResRows = 192, kResStride = 160;      /* a product line holds rows 36..191 (the first 36 live in registers) */
constexpr int kResJFloats = kResRows * 6, kResSFloats = 8 * kResRows;   /* slot 6: zeros (what off-diagonal entries add), slot 7: Manifold row? */
constexpr int kResProductBytesPerWave = 27 * kResStride * 4;
/* two workgroups per CU (k_iterate<2>): short product lines, rows 36..103 (4 x 7 344 B + 45 KB of row staging = 75 KB per
 * workgroup); a body with more than kResLiteMaxEntries entries multiplies its last rows after the wait (accumulate_products) */
constexpr int kResStrideLite = 68, kResLiteMaxEntries = 33;
constexpr int kResProductBytesPerWaveLite = 27 * kResStrideLite * 4;
static_assert(kResJFloats + kResSFloats <= 66 * kEntryRec, "the J and S areas reuse the row staging area of k_iterate");

/* rows 0..kResRegRows-1 keep their product in a register of the accumulator lane (no LDS read after the wait);
 * later rows (a body with more than 12 contacts) go through the lane's product line in LDS */
constexpr int kResRegRows = 36;

__device__ __forceinline__ void store_products(const AccLane& a, const float* J, float (&pr)[kResRegRows], float* prLine, int nRows, int lineRows)
{
#pragma unroll
    for (int r = 0; r < kResRegRows; ++r) pr[r] = 0.0f;
    const bool rhs = a.mode == 2;                                   /* idle lanes (mode 3) multiply J[.][0] by itself: harmless, unread */
    const float* ja = J + a.opA;
    const float* jb = J + (rhs ? a.opA : a.opB);
#pragma unroll
    for (int e = 0; e < kResRegRows / 3; ++e) {
        if (e * 3 < nRows) {                                        /* wave-uniform */
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = e * 3 + k;
                const float va = ja[r * 6], vb = jb[r * 6];
                pr[r] = rhs ? va : va * vb;
            }
        }
    }
    const int lineEnd = nRows < kResRegRows + lineRows ? nRows : kResRegRows + lineRows;   /* rows past the line: multiplied after the wait */
    for (int r = kResRegRows; r < lineEnd; ++r) {
        const float va = ja[r * 6], vb = jb[r * 6];
        prLine[r - kResRegRows] = rhs ? va : va * vb;
    }
}

__device__ __forceinline__ float comp4(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

template <bool ALLON, bool JOINTS, bool SHORT>
__device__ __forceinline__ void accumulate_products(AccLane& a, const float (&pr)[kResRegRows], const float* prLine, const float* S, int nRows, const float* J, int lineRows)
{
    /* every lane runs the same instruction stream (idle lanes 27..63 compute a value nobody reads): no divergent branches,
     * only wave-uniform ones on the row count.  ALLON: every row is a Manifold row with a usable penalty (the normal case);
     * otherwise two per-row flags gate the add (penalty usable, solver.cpp:387) and the geometric-stiffness term (Manifold
     * rows only, solver.cpp:393). */
    const bool isAA = a.mode == 1, diag = a.gy >= 0, rhs = a.mode == 2;
    const float4* s4 = reinterpret_cast<const float4*>(S + (rhs ? 0 : kResRows));
    /* what an entry adds behind the row term: its gyroscopic component (diagonal aa entries), +0.0f (off-diagonal aa entries, like
     * the reference's diagonal() matrix), or — contacts-only kernels — -0.0f for every other entry: x + (-0.0f) == x for EVERY x,
     * so those lanes need no select around the addition the aa lanes make (slot 7 holds -0.0f when the kernel has no joints) */
    const float4* g4 = reinterpret_cast<const float4*>(S + kResRows * (diag ? a.gy - 6 : ((!JOINTS && !isAA) ? 7 : 6)));
    const float4* o4 = reinterpret_cast<const float4*>(S + kResRows * 5);
    const float4* m4 = reinterpret_cast<const float4*>(S + kResRows * 7);
    constexpr int NQ = kResRegRows / 4;
    float4 sv[NQ], gv[NQ], ov[NQ], mv[NQ];
    /* every operand of the first 36 rows is requested before the dependent chain of additions starts */
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        /* the first blocks are fetched unconditionally (their slots always exist), the later ones only when the body has
         * that many rows; a block that is not fetched is not read below (same wave-uniform condition) */
        if (q < 3 || q * 4 < nRows) {                               /* wave-uniform */
            sv[q] = s4[q];
            gv[q] = g4[q];
            if (!ALLON) { ov[q] = o4[q]; if (JOINTS) mv[q] = m4[q]; }
        }
    }
#pragma unroll
    for (int e = 0; e < kResRegRows / 3; ++e) {
        if (e * 3 < nRows) {                                        /* wave-uniform */
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = e * 3 + k;
                const float nv = a.acc + pr[r] * comp4(sv[r >> 2], r & 3);
                const float nvg = nv + comp4(gv[r >> 2], r & 3);     /* off-diagonal aa entries add 0.0f like the reference's diagonal() */
                if (ALLON) a.acc = JOINTS ? (isAA ? nvg : nv) : nvg;
                else {
                    const float nv2 = (isAA && (!JOINTS || comp4(mv[r >> 2], r & 3) != 0.0f)) ? nvg : nv;
                    a.acc = (rhs || comp4(ov[r >> 2], r & 3) != 0.0f) ? nv2 : a.acc;
                }
            }
        }
    }
    /* the rare tail: rows past the register window, four at a time from the LDS product line; rows past the end of a short
     * line (two workgroups per CU, a body with more than 34 entries) are multiplied here, from J — same product, same bits */
    const float4* p4 = reinterpret_cast<const float4*>(prLine);
    const float* ja = J + a.opA;
    const float* jb = J + (rhs ? a.opA : a.opB);
    /* (round 5: EIGHT rows per trip — every trip starts with an LDS round trip in front of its dependent additions, ~130 cycles, and four bodies of the critical
     * cycle of the steady Stress1000 pile have 20 to 36 rows past the register window; the second block of a trip is fetched only when the body has it) */
    auto tail_products = [&](int r4) __attribute__((always_inline)) -> float4 {
        if (!SHORT || r4 - kResRegRows + 4 <= lineRows) return p4[(r4 - kResRegRows) >> 2];   /* wave-uniform; SHORT = false: the line holds every row */
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r4 + k < nRows ? r4 + k : r4;
            const float va = ja[r * 6], vb = jb[r * 6];
            t[k] = rhs ? va : va * vb;
        }
        return make_float4(t[0], t[1], t[2], t[3]);
    };
    auto tail_add = [&](int r4, const float4& pv, const float4& sw, const float4& gw, const float4& ow, const float4& mw) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (r4 + k < nRows) {                                  /* wave-uniform */
                const float nv = a.acc + comp4(pv, k) * comp4(sw, k);
                const float nvg = nv + comp4(gw, k);
                if (ALLON) a.acc = JOINTS ? (isAA ? nvg : nv) : nvg;
                else {
                    const float nv2 = (isAA && comp4(mw, k) != 0.0f) ? nvg : nv;
                    a.acc = (rhs || comp4(ow, k) != 0.0f) ? nv2 : a.acc;
                }
            }
        }
    };
    for (int r4 = kResRegRows; r4 < nRows; r4 += 8) {
        const bool two = r4 + 4 < nRows;                               /* wave-uniform */
        const float4 pv0 = tail_products(r4), sw0 = s4[r4 >> 2], gw0 = g4[r4 >> 2];
        float4 ow0 = make_float4(1.f, 1.f, 1.f, 1.f), mw0 = ow0, ow1 = ow0, mw1 = ow0, pv1 = ow0, sw1 = ow0, gw1 = ow0;
        if (!ALLON) { ow0 = o4[r4 >> 2]; if (JOINTS) mw0 = m4[r4 >> 2]; }
        if (two) {
            pv1 = tail_products(r4 + 4); sw1 = s4[(r4 + 4) >> 2]; gw1 = g4[(r4 + 4) >> 2];
            if (!ALLON) { ow1 = o4[(r4 + 4) >> 2]; if (JOINTS) mw1 = m4[(r4 + 4) >> 2]; }
        }
        tail_add(r4, pv0, sw0, gw0, ow0, mw0);
        if (two) tail_add(r4 + 4, pv1, sw1, gw1, ow1, mw1);
    }
}

/* what the owner of a joint half leaves in the GForce record: between its tasks (MULTI) and at the end of the step */
__device__ __forceinline__ void store_joint_half(GForce& f, int part, const ContactRows& c, bool withRows)
{
    for (int k = 0; k < 3; ++k) {
        f.lambda[part * 3 + k] = c.lambda[k]; f.penalty[part * 3 + k] = c.penalty[k];
        if (withRows) { f.C[part * 3 + k] = c.C[k]; f.fmin[part * 3 + k] = -FLT_MAX; f.fmax[part * 3 + k] = FLT_MAX; }
    }
}

/* One wavefront = one body for all iterations (MULTI = false: the body's state stays in registers from the first iteration to
 * the last), or one wavefront = every W-th body of the level-sorted order (MULTI = true, scenes with more swept bodies than
 * resident waves: the state of a body is parked between its tasks — owned contacts in the Manifold tables, which only their
 * owner writes; the first visitor's copy of normal + tangent memo and the pose in arrays private to the wave).  Tasks are
 * taken in the order t = it * slots + slot, a linear extension of the dependency order, so the smallest unfinished task is
 * always at the head of a resident wave's list.  `rows` / `accs` are the wave's LDS staging areas of k_iterate. */
template <bool MULTI, bool JOINTS, bool COLO = false, bool TWOSTAGE = false>      /* TWOSTAGE: accs has 96 floats, [64..88] for solve_pose_pre */
__device__ __forceinline__ void iterate_resident(const BodyTable& B, const ManifoldTable& M, GForce* gf, const int* __restrict__ order, int nd, int w, int W,
                                                 const int* __restrict__ incStart, const int* __restrict__ inc, const SolverParams& P, int* errWord,
                                                 Counters* cntW, const ResidentExchange& X, int sleepSel, int lane, float* rows, float* accs, float* products,
                                                 int itBegin, int itEnd, CoLocal CL = CoLocal{nullptr, nullptr, nullptr, 0}, int placedBody = -1)
{
    static_assert(!COLO || !MULTI, "the co-located hand-off is for one body per wave");
    static_assert(!TWOSTAGE || (!MULTI && !JOINTS), "the two-stage solve is for one body per wave, contacts only");
    const __amdgpu_buffer_rsrc_t rP = xrsrc(X.pose, X.poseBytes), rV = xrsrc(X.visit, X.visitBytes), rD = xrsrc(X.dual, X.dualBytes);
    const int totalIt = P.iterations + (P.postStabilize ? 1 : 0);
    const float4* massPlane = B.plane(BP_MASS);

    /* ---- the state of the task in hand ---- */
    int body = 0, eBeg = 0, nEnt = 0;
    bool has = false;
    int idx = 0, other = 0; bool isA = false, owned = false, moving = false, fixedN = false;
    bool isJoint = false, hasA = true;                               /* a lane holds one Manifold contact or one half (3 rows) of a Joint */
    int co = -1;                                                     /* COLO: wave of this workgroup that holds the partner, -1: none */
    q4 jq0 = identq(); f3 jStiff = zero3(), jMotor = zero3(); int jPart = 0; bool jFix = false;
    float oinv = 0.0f, fric = 0.0f;
    ContactRows c;
    c.rA = c.rB = c.normal = c.u1 = c.u2 = zero3(); c.C0n = c.C0t1 = c.C0t2 = 0.0f; c.penetration = 0.0f; c.stick = false;
    for (int k = 0; k < 3; ++k) { c.lambda[k] = 0.0f; c.penalty[k] = 0.0f; c.C[k] = 0.0f; c.fmin[k] = 0.0f; c.fmax[k] = 0.0f; }
    f3 opos = zero3(); q4 orot = identq(); f3 wOther = zero3(), pOther = zero3();
    BodyLocal L;
    L.invDt2 = 1.0f / (P.dt * P.dt);
    float radius = 0.0f;
    m33 Ib, Jb;
    int offP = 0, offV = 0, offD = 0;

    /* everything a task needs that does not come from another wave of this launch: the tables the earlier kernels of this step
     * wrote (first iteration), or what this same wave parked after the body's previous task (MULTI, later iterations) */
    /* who the entries of a body are and whom they wait for: enough to tell whether a task can start */
    auto load_header = [&](int body_, int eBeg_, int nEnt_, bool parked) {
        body = body_; eBeg = eBeg_; nEnt = nEnt_;
        has = lane < nEnt;
        isJoint = false; moving = false; owned = false; other = 0; idx = 0; isA = false;
        int pw = 0;
        if (has) {
            const int ent = inc[eBeg + lane];
            if (MULTI && parked) pw = X.probe[eBeg + lane];
            idx = inc_index(ent); isA = inc_isA(ent) != 0;
            isJoint = JOINTS && inc_kind(ent) == FK_JOINT;         /* JOINTS = false: contacts only, the joint code below is compiled out */
        }
        if (MULTI && parked) {                                     /* wave-uniform: the body has been examined before */
            other = pw & 0xffffff; moving = (pw >> 24) & 1; owned = (pw >> 25) & 1;
            if (JOINTS) { hasA = (pw >> 26) & 1; jPart = idx & 1; }
        } else if (JOINTS && has && isJoint) {
            const GForce& f = gf[idx >> 1];
            jPart = idx & 1;
            hasA = f.bodyA >= 0;
            other = isA ? f.bodyB : f.bodyA;
            moving = other >= 0 && B.solved[other] != 0;
            owned = !moving || other > body;
            if (other < 0) other = 0;                                  /* world anchor: a fixed partner, never polled */
        } else if (has) {
            const int2 pr = M.pair[idx >> 2];
            other = isA ? pr.y : pr.x;
            moving = B.solved[other] != 0;
            owned = !moving || other > body;
        }
        if (MULTI && !parked && X.outOfOrder && has) X.probe[eBeg + lane] = other | ((moving ? 1 : 0) << 24) | ((owned ? 1 : 0) << 25) | ((hasA ? 1 : 0) << 26);
        offP = other * 48; offV = idx * 16; offD = ((JOINTS && isJoint) ? X.jointRecBase + idx : idx) * 32;
        if (COLO) {
            co = -1;
            if (has && moving)
                for (int k = 0; k < kIterWaves; ++k) if (CL.bodyOf[k] == other) co = k;
        }
    };
    /* everything else a task needs that does not come from another wave of this launch: the tables the earlier kernels of this
     * step wrote (first iteration), or what this same wave parked after the body's previous task (MULTI, later iterations) */
    auto load_rest = [&](int it) {
        fixedN = false;
        if (JOINTS && has && isJoint) {
            /* one half of a Joint (joint.cpp:68-139): rows 3*part .. 3*part+2; lambda / penalty live in the GForce record, which
             * only the owner writes during the launch (and where it parks them between its tasks) */
            const GForce& f = gf[idx >> 1];
            c.rA = mk3(f.rA[0], f.rA[1], f.rA[2]); c.rB = mk3(f.rB[0], f.rB[1], f.rB[2]);
            jq0 = mkq(f.q0[0], f.q0[1], f.q0[2], f.q0[3]); jFix = (f.pad & 1) != 0;
            for (int k = 0; k < 3; ++k) { c.lambda[k] = f.lambda[jPart * 3 + k]; c.penalty[k] = f.penalty[jPart * 3 + k]; c.C[k] = f.C[jPart * 3 + k]; }
            jStiff = mk3(f.stiffness[jPart * 3], f.stiffness[jPart * 3 + 1], f.stiffness[jPart * 3 + 2]);
            jMotor = mk3(f.motor[jPart * 3], f.motor[jPart * 3 + 1], f.motor[jPart * 3 + 2]);
            opos = zero3(); orot = identq();
            if (isA || hasA) {
                const float4 op = B.plane(BP_POSR)[other], oq = B.plane(BP_QUAT)[other];
                opos = mk3(op.x, op.y, op.z); orot = mkq(oq.x, oq.y, oq.z, oq.w);
            }
        } else if (has) {
            const int m = idx >> 2;
            oinv = massPlane[other].y;
            fric = M.fric[m];
            load_contact<false>(M, idx, c);                           /* owned contacts: also what this wave's last dual left there */
            if (MULTI && it > 0 && !owned) {                          /* this visitor's own copy of the normal and its memo */
                const float4 pn = X.privN[(size_t)(eBeg + lane) * 3], p1 = X.privN[(size_t)(eBeg + lane) * 3 + 1], p2 = X.privN[(size_t)(eBeg + lane) * 3 + 2];
                c.normal = mk3(pn.x, pn.y, pn.z); c.u1 = mk3(p1.x, p1.y, p1.z); c.u2 = mk3(p2.x, p2.y, p2.z);
            }
            const float4 op = B.plane(BP_POSR)[other], oq = B.plane(BP_QUAT)[other];   /* predicted pose (solver.cpp:299-337); constant for unswept partners */
            opos = mk3(op.x, op.y, op.z); orot = mkq(oq.x, oq.y, oq.z, oq.w);
            wOther = qrotate(orot, isA ? c.rB : c.rA);
            pOther = opos + wOther;
        }
        {
            float4 pp = B.plane(BP_POSR)[body], pq = B.plane(BP_QUAT)[body];
            radius = pp.w;
            if (MULTI && it > 0) { pp = X.privPose[(size_t)body * 2]; pq = X.privPose[(size_t)body * 2 + 1]; }
            const float4 snp = B.plane(BP_NPOS)[body], snq = B.plane(BP_NQUAT)[body], smp = massPlane[body];
            L.pos = mk3(pp.x, pp.y, pp.z); L.rot = mkq(pq.x, pq.y, pq.z, pq.w);
            L.npos = mk3(snp.x, snp.y, snp.z); L.nrot = mkq(snq.x, snq.y, snq.z, snq.w);
            L.mass = smp.x; L.invMass = smp.y;
        }
        const float4 i0 = B.plane(BP_I0)[body], i1 = B.plane(BP_I1)[body], i2 = B.plane(BP_I2)[body];
        const float4 j0 = B.plane(BP_J0)[body], j1 = B.plane(BP_J1)[body], j2 = B.plane(BP_J2)[body];
        Ib = mkm(mk3(i0.x, i0.y, i0.z), mk3(i1.x, i1.y, i1.z), mk3(i2.x, i2.y, i2.z));
        Jb = mkm(mk3(j0.x, j0.y, j0.z), mk3(j1.x, j1.y, j1.z), mk3(j2.x, j2.y, j2.z));
    };
    auto load_task = [&](int slot, int it) {
        const int b = __builtin_amdgcn_readfirstlane((COLO && placedBody >= 0) ? placedBody : order[slot]);    /* wave-uniform: scalar loads and scalar branches below */
        const int e0 = __builtin_amdgcn_readfirstlane(incStart[b]);
        load_header(b, e0, __builtin_amdgcn_readfirstlane(incStart[b + 1]) - e0, false);
        load_rest(it);
    };
    /* MULTI: park the body's state until its next task; the last task leaves the step's final state (rows included) */
    auto save_task = [&](bool last) {
        if (JOINTS && has && isJoint) {
            if (owned) store_joint_half(gf[idx >> 1], jPart, c, last);
        } else if (has) {
            if (owned) store_contact_dual<false>(M, idx, c, last);
            else {
                X.privN[(size_t)(eBeg + lane) * 3] = make_float4(c.normal.x, c.normal.y, c.normal.z, 0.f);
                X.privN[(size_t)(eBeg + lane) * 3 + 1] = make_float4(c.u1.x, c.u1.y, c.u1.z, 0.f);
                X.privN[(size_t)(eBeg + lane) * 3 + 2] = make_float4(c.u2.x, c.u2.y, c.u2.z, 0.f);
            }
        }
        if (lane == 0) {
            if (last || X.planesAlways) { B.plane(BP_POSR)[body] = pack3(L.pos, radius); B.plane(BP_QUAT)[body] = packq(L.rot); }
            if (X.send) { const int k = X.sendSlot[body]; X.send[2 * k] = pack3(L.pos, radius); X.send[2 * k + 1] = packq(L.rot); }
            if (!last) { X.privPose[(size_t)body * 2] = pack3(L.pos, radius); X.privPose[(size_t)body * 2 + 1] = packq(L.rot); }
        }
    };

    for (int r = lane; r < kResRows; r += 64) { rows[kResJFloats + 6 * kResRows + r] = 0.0f; if (!JOINTS) rows[kResJFloats + 7 * kResRows + r] = -0.0f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned prevWait = 0;                                           /* ticks (10 ns) this wave waited in the previous iteration */
    int critOther = -1;                                              /* COLO: the later-created partner whose pose arrived last in the most recent task that waited for one */
#ifdef AVBD_DF_STATS
    unsigned long long stat[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = t0;
#define RS_STAMP(i) do { unsigned long long _n = __builtin_amdgcn_s_memrealtime(); stat[i] += _n - tprev; tprev = _n; } while (0)
#define RS_STAMP_DRAINED(i) do { __builtin_amdgcn_s_waitcnt(0); RS_STAMP(i); } while (0)   /* after the loads / stores issued so far have completed */
#else
#define RS_STAMP(i) do { } while (0)
#define RS_STAMP_DRAINED(i) do { } while (0)
#endif

    /* [itBegin, itEnd): the whole loop on one GPU; one iteration per launch when poses are exchanged between GPUs after every
     * iteration (MULTI only: the parked state and the stamped records carry over from launch to launch) */
    const int itFirst = MULTI ? itBegin : 0, itStop = (MULTI && itEnd < totalIt) ? itEnd : totalIt;

    /* One task = prepare_task (everything that depends only on this wave's own state: may run long before the inputs arrive)
     * + finish_task (wait for the partners' records, second half of the visit, accumulate, solve, publish, dual).
     * finish_task returns 1 when the task is done, 0 when the launch is being abandoned, 2 when `patience` (ticks of 10 ns,
     * 0 = unlimited) ran out before the inputs arrived: nothing has been written by then, the task is simply taken up again
     * later.  What the first half leaves for the second: */
    struct TaskPrep {
        AccLane a;
        f3 wOwn;
        bool needP, needV, needD;
        unsigned sP, sV, sD;
        ContactPre pre; f3 G[3];
        float pr[kResRegRows];
    };
    float* const J = rows;
    float* const S = rows + kResJFloats;
    float* const prLine = products + (lane < 27 ? lane : 0) * X.prStride;
    int tmoOff = -1; unsigned tmoStamp = 0;                         /* the record a timed-out task was still waiting for */
#ifdef AVBD_DF_STATS
    unsigned trecDummy[8];                                          /* per-task time stamps are kept for one body per wave only */
#define RS_TREC(it) (MULTI ? trecDummy : reinterpret_cast<unsigned*>(errWord + 16 + 1024 * 32) + ((size_t)(it) * 1024 + (blockIdx.x * kIterWaves + (threadIdx.x >> 6))) * 8)
#endif

    auto prepare_task = [&](int it, TaskPrep& T) __attribute__((always_inline)) {
        AccLane& a = T.a; f3& wOwn = T.wOwn; bool& needP = T.needP; bool& needV = T.needV; bool& needD = T.needD;
        unsigned& sP = T.sP; unsigned& sV = T.sV; unsigned& sD = T.sD; ContactPre& pre = T.pre; f3 (&G)[3] = T.G; float (&pr)[kResRegRows] = T.pr;
#ifdef AVBD_DF_STATS
        unsigned* trec = RS_TREC(it);
        if (lane == 0) { trec[0] = (unsigned)__builtin_amdgcn_s_memrealtime(); trec[5] = (unsigned)body; }
#endif
        AVBD_MARK(prepare);
        const float alpha = P.postStabilize ? (it < P.iterations ? 1.0f : 0.0f) : P.alpha;
        const unsigned stampNow = X.base + (unsigned)it + 1u;        /* what this task publishes */

        /* ---- before the wait: everything that depends only on this wave's own state ---- */
        world_inertia_parallel(lane, L.rot, Ib, Jb, accs + 32, L.Iw, L.invIw);   /* rigid.cpp:51-59 */
        a = init_acc_lane(lane, L);
        wOwn = zero3();
        needP = false; needV = false; needD = false;
        sP = 0; sV = 0; sD = 0;
        if (JOINTS && has && isJoint) {
            if (moving) {
                needP = true;
                if (owned) sP = stampNow;
                else if (it > 0) { needD = true; sD = X.base + (unsigned)it; sP = sD; }
                else needP = false;
            }
            /* Jacobians of this body's three rows (joint.cpp:108-139): functions of its own orientation only */
            JointRows jr0; jr0.rA = c.rA; jr0.rB = c.rB; jr0.q0 = jq0; jr0.fixAngular = jFix;
            for (int k = 0; k < 3; ++k) {
                f3 Jl, Ja;
                joint_compute_derivatives(jr0, isA, false, L.rot, jPart * 3 + k, Jl, Ja);
                G[k] = zero3();
                float* jr = J + (lane * 3 + k) * 6;
                jr[0] = Jl.x; jr[1] = Jl.y; jr[2] = Jl.z; jr[3] = Ja.x; jr[4] = Ja.y; jr[5] = Ja.z;
            }
        } else if (has) {
            wOwn = qrotate(L.rot, isA ? c.rA : c.rB);                 /* manifold.cpp:27 and :265 use the same product */
            if (moving) {
                if (owned) { replay_normal(c, fixedN); needV = true; sV = stampNow; needP = true; sP = stampNow; }
                else {
                    if (it > 0) { replay_normal(c, fixedN); replay_normal(c, fixedN); needD = true; sD = X.base + (unsigned)it; needP = true; sP = sD; }
                }
            }
            /* first half of this visit's computeConstraint + computeDerivatives (manifold.cpp:247-271): own state only */
            ContactBasis db;
            contact_constraint_pre(c, alpha, isA ? L.invMass : oinv, isA ? oinv : L.invMass, pre, db);
            f3 Jl[3], Ja[3];
            contact_compute_derivatives_w(db, isA, wOwn, Jl, Ja);
            for (int k = 0; k < 3; ++k) {
                G[k] = abs3(cross3(Ja[k], L.invIw * Ja[k]));          /* solver.cpp:393-397, to be scaled by |f| */
                float* jr = J + (lane * 3 + k) * 6;
                jr[0] = Jl[k].x; jr[1] = Jl[k].y; jr[2] = Jl[k].z; jr[3] = Ja[k].x; jr[4] = Ja[k].y; jr[5] = Ja[k].z;
            }
        }
        wave_sync();
        store_products(a, J, pr, prLine, nEnt * 3, MULTI ? X.prStride : kResStride);
        RS_STAMP(7);
#ifdef AVBD_DF_STATS
        if (lane == 0) trec[1] = (unsigned)tprev;
#endif
    };

    auto finish_task = [&](int it, TaskPrep& T, unsigned patience) __attribute__((always_inline)) -> int {
        AccLane& a = T.a; f3& wOwn = T.wOwn; bool& needP = T.needP; bool& needV = T.needV; bool& needD = T.needD;
        unsigned& sP = T.sP; unsigned& sV = T.sV; unsigned& sD = T.sD; ContactPre& pre = T.pre; f3 (&G)[3] = T.G; float (&pr)[kResRegRows] = T.pr;
#ifdef AVBD_DF_STATS
        unsigned* trec = RS_TREC(it);
#endif
        const float alpha = P.postStabilize ? (it < P.iterations ? 1.0f : 0.0f) : P.alpha;
        const bool doDual = it < P.iterations;
        const unsigned stampNow = X.base + (unsigned)it + 1u;
        bool preDone = false;                                        /* TWOSTAGE: stage one of the solve was made during the wait */

        /* ---- the wait IS the load: poll the records until every stamp is current ---- */
        AVBD_MARK(wait);
        {
            /* A wave that waited w last iteration will wait about w again (the dependency pattern repeats every sweep): it
             * stays off the memory system for the first part of that time.  Polls of idle waves were measured to lengthen
             * everybody's hand-off latency. */
            const unsigned long long tWait0 = __builtin_amdgcn_s_memrealtime();
            if (!MULTI && (needP | needV | needD)) {
                const unsigned quiet = (prevWait * (unsigned)X.quietPermille) / 1000u;
                while ((unsigned)(__builtin_amdgcn_s_memrealtime() - tWait0) < quiet) __builtin_amdgcn_s_sleep(8);
            }
            preDone = false;
            const bool viaLds = COLO && needP && co >= 0;             /* the partner's wave shares this workgroup: its pose comes through LDS (second loop below) */
            bool okP = !needP || viaLds, okV = !needV, okD = !needD;
            v4u p0 = {0, 0, 0, 0}, p1 = p0, p2 = p0, vv = p0, d0 = p0, d1 = p0;
            unsigned spins = 0;
            unsigned long long lastMissing = 0;                       /* lanes that still missed a same-iteration pose in the round before the last */
            /* TWOSTAGE: once the dual records (the penalties) are in and a pose or visit record is still out, the sums and the part of the
             * solve that depend on nothing else are made while waiting (solve_pose_pre); the poll then goes on */
            for (int pass = 0; pass < (TWOSTAGE ? 2 : 1); ++pass) {
            bool allIn = false;
            while (true) {
                const bool lp = !okP, lv = !okV, ld = !okD;
                if (lp) { p0 = xld(rP, offP); p1 = xld(rP, offP + 16); p2 = xld(rP, offP + 32); }
                if (lv) vv = xld(rV, offV);
                if (ld) { d0 = xld(rD, offD); d1 = xld(rD, offD + 16); }
                asm volatile("" ::: "memory");                        /* the loads are re-issued on every round */
                if (lp) okP = stamp_reached(p0.w, sP) & stamp_reached(p1.w, sP) & stamp_reached(p2.w, sP);
                if (lv) okV = stamp_reached(vv.w, sV);
                if (ld) okD = stamp_reached(d0.w, sD) & stamp_reached(d1.w >> 1, sD);
                if (__all(okP & okV & okD)) { allIn = true; break; }
                if (TWOSTAGE && pass == 0 && __all(okD)) break;
                if (COLO) lastMissing = __ballot(!okP && owned);
#ifdef AVBD_DF_STATS
                stat[14] += 1;
#endif
                if (sleepSel == 0) __builtin_amdgcn_s_sleep(1); else if (sleepSel == 1) __builtin_amdgcn_s_sleep(2);
                else if (sleepSel == 2) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(8);
                if ((++spins & 1023u) == 0) {
                    bool dead = (__builtin_amdgcn_s_memrealtime() - tWait0 > kSpinTimeoutTicks) ||   /* measured from the start of THIS wait */
                                __hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                    if (__any(dead)) {
                        if constexpr (MULTI) {   /* what the first wave to give up was waiting for (errWord[1..15]; read by the host for its message); not in the
                                                  * one-body-per-wave kernels: the extra live values cost the headline launch 1.6 % */
                            const bool mine = __builtin_amdgcn_s_memrealtime() - tWait0 > kSpinTimeoutTicks;
                            const unsigned long long mP = __ballot(!okP), mV = __ballot(!okV), mD = __ballot(!okD);
                            const unsigned long long mAny = mP | mV | mD;
                            const int stuck = mAny ? __ffsll((long long)mAny) - 1 : 0;
                            const int so = __builtin_amdgcn_readlane(other, stuck);
                            if (lane == 0 && mine && __hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && atomicCAS(errWord + 1, 0, w + 1) == 0) {
                                errWord[2] = body; errWord[3] = so; errWord[4] = it; errWord[5] = (int)mP; errWord[6] = (int)(mP >> 32); errWord[7] = (int)mV; errWord[8] = (int)(mV >> 32);
                                errWord[9] = (int)mD; errWord[10] = (int)(mD >> 32); errWord[11] = (int)blockIdx.x; errWord[12] = (int)gridDim.x; errWord[13] = MULTI ? 1 : 0; errWord[14] = (int)spins; errWord[15] = 50;
                            }
                        }
                        if (lane == 0) __hip_atomic_store(errWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return 0;
                    }
                }
                if (MULTI && patience && (unsigned)(__builtin_amdgcn_s_memrealtime() - tWait0) > patience) {
                    /* give the wave back to its other bodies; remember one record this task was still missing */
                    const int f = __ffsll((long long)__ballot(!(okP & okV & okD))) - 1;
                    tmoOff = __builtin_amdgcn_readlane(offP + 32, f);
                    tmoStamp = (unsigned)__builtin_amdgcn_readlane((int)sP, f);
                    return 2;
                }
            }
            if (TWOSTAGE && pass == 0) {
                bool early = !allIn;
                if (COLO && allIn && __any(viaLds)) {                   /* everything from memory is in: is a pose that comes through LDS still out? */
                    unsigned lf = 0;
                    if (viaLds) lf = CL.flag[co];
                    asm volatile("" ::: "memory");
                    early = !__all(!viaLds || stamp_reached(lf, sP));
                }
                if (early) {
                    AVBD_MARK(stage_one);
                    if (needD) { c.penalty[0] = bitsf(d1.x); c.penalty[1] = bitsf(d1.y); c.penalty[2] = bitsf(d1.z); }
                    bool rowsOnA = true;
                    if (has) {
                        for (int k = 0; k < 3; ++k) {
                            const bool penOk = (c.penalty[k] > 0.0f) & (fabsf(c.penalty[k]) <= FLT_MAX);   /* solver.cpp:387 */
                            const int r = lane * 3 + k;
                            S[kResRows + r] = c.penalty[k]; S[5 * kResRows + r] = penOk ? 1.0f : 0.0f;
                            rowsOnA &= penOk;
                        }
                    }
                    const bool allOnA = __all(rowsOnA);
                    wave_sync();
                    AccLane a1 = a;                                      /* (entries 15..26 read slots of the previous task here: nobody reads their sums) */
                    if (allOnA) accumulate_products<true, JOINTS, MULTI>(a1, pr, prLine, S, nEnt * 3, J, X.prStride); else accumulate_products<false, JOINTS, MULTI>(a1, pr, prLine, S, nEnt * 3, J, X.prStride);
                    if (lane < 15) accs[lane] = a1.acc;
                    wave_sync();
                    solve_pose_pre(accs, accs + 64, lane);
                    preDone = true;
                }
            }
            if (allIn) break;
            }
            if (COLO && __any(viaLds)) {
                /* the poses that come through LDS: a short loop on the flag words alone (nothing is in flight any more, a round is
                 * one LDS access), normally entered long before the flag rises — the placement puts the hand-off a task waits for
                 * LAST into its own workgroup */
                /* (the records are read in the same round as the flag, BEHIND it: the LDS executes a wave's accesses in order and the producer writes the flag
                 * behind the records, so the round that sees the flag holds the pose — no second round trip between the flag and the solve) */
                bool okL = !viaLds;
                unsigned spinsL = 0;
                const volatile v4u AVBD_LDS* rl = CL.rec + (co >= 0 ? co : 0) * 3;
                while (true) {
                    unsigned lf = 0;
                    if (!okL) { lf = CL.flag[co]; p0 = rl[0]; p1 = rl[1]; p2 = rl[2]; }
                    asm volatile("" ::: "memory");
                    if (!okL) okL = stamp_reached(lf, sP);
                    if (__all(okL)) break;
                    lastMissing = __ballot(!okL);
                    if ((++spinsL & 8191u) == 0) {
                        bool dead = (__builtin_amdgcn_s_memrealtime() - tWait0 > kSpinTimeoutTicks) ||
                                    __hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (__any(dead)) { if (lane == 0) __hip_atomic_store(errWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 0; }
                    }
                }
                spins += spinsL;
            }
            AVBD_MARK(post_visit);
            RS_STAMP(0);
#ifdef AVBD_DF_STATS
            if (lane == 0) { trec[2] = (unsigned)tprev; trec[6] = spins; }
            if (spins >= 2) {                                        /* a real wait: how long ago was the last awaited pose published? */
                unsigned now = (unsigned)tprev;
                unsigned age = needP ? now - p2.y : 0xffffffffu;
                for (int o = 32; o > 0; o >>= 1) { unsigned other_ = __shfl_xor(age, o); age = other_ < age ? other_ : age; }
                if (age != 0xffffffffu && age < 100000u) { stat[12] += age; stat[13] += 1; }
            }
#endif
            prevWait = (unsigned)(__builtin_amdgcn_s_memrealtime() - tWait0);
#ifdef AVBD_DF_STATS
            if (lane == 0) race_note_consumed(errWord - 2 * B.cap, body, it);
#endif
            if (COLO) {
                asm volatile("" ::: "memory");
                /* whose pose did this task wait for last?  (a later-created partner: the hand-off a shared workgroup would shorten) */
                if (lastMissing) critOther = __builtin_amdgcn_readlane(other, __ffsll((long long)lastMissing) - 1);
            }
            if (needP) {
                opos = mk3(bitsf(p0.x), bitsf(p0.y), bitsf(p0.z)); orot = mkq(bitsf(p1.x), bitsf(p1.y), bitsf(p1.z), bitsf(p2.x));
                if (!(JOINTS && isJoint)) {
                    wOther = qrotate(orot, isA ? c.rB : c.rA);
                    pOther = opos + wOther;
                }
            }
            if (needV) { c.lambda[1] = bitsf(vv.x); c.lambda[2] = bitsf(vv.y); c.stick = vv.z != 0; }
            if (needD) {
                c.lambda[0] = bitsf(d0.x); c.lambda[1] = bitsf(d0.y); c.lambda[2] = bitsf(d0.z);
                c.penalty[0] = bitsf(d1.x); c.penalty[1] = bitsf(d1.y); c.penalty[2] = bitsf(d1.z);
                c.stick = (d1.w & 1u) != 0;                           /* (unused by joints) */
            }
        }

        /* ---- primal visit (solver.cpp:344-409) ---- */
        bool rowsOn = true;
        if (JOINTS && has && isJoint) {
            /* Joint::computeConstraint (joint.cpp:68-106) + the row forces of solver.cpp:372-385 for this half */
            JointRows jr1; jr1.rA = c.rA; jr1.rB = c.rB; jr1.q0 = jq0;
            if (isA) joint_compute_constraint(jr1, hasA, L.pos, L.rot, opos, orot);
            else     joint_compute_constraint(jr1, hasA, opos, orot, L.pos, L.rot);
            for (int k = 0; k < 3; ++k) {
                const float stiff = comp3(jStiff, k);
                c.C[k] = jr1.C[jPart * 3 + k];
                const float lw = (stiff == FLT_MAX) ? c.lambda[k] : 0.0f;
                const float desired = c.penalty[k] * c.C[k] + lw + comp3(jMotor, k);
                const float f = rclamp(desired, -FLT_MAX, FLT_MAX);
                const bool penOk = (c.penalty[k] > 0.0f) & (fabsf(c.penalty[k]) <= FLT_MAX);
                const int r = lane * 3 + k;
                S[r] = f; S[kResRows + r] = c.penalty[k]; S[2 * kResRows + r] = 0.0f; S[3 * kResRows + r] = 0.0f; S[4 * kResRows + r] = 0.0f;
                S[5 * kResRows + r] = penOk ? 1.0f : 0.0f;
                S[7 * kResRows + r] = 0.0f;                             /* not a Manifold row: no geometric-stiffness term (solver.cpp:393) */
            }
            rowsOn = false;                                            /* the predicate-free accumulation is for all-Manifold bodies */
        } else if (has) {
            const f3 pOwn = L.pos + wOwn;
            contact_constraint_post(c, pre, isA ? pOwn : pOther, isA ? pOther : pOwn, fric);
            if (!owned) { v4u r = {fbits(c.lambda[1]), fbits(c.lambda[2]), c.stick ? 1u : 0u, stampNow}; xst(rV, offV, r); }
            for (int k = 0; k < 3; ++k) {
                const float desired = c.penalty[k] * c.C[k] + c.lambda[k] + 0.0f;     /* solver.cpp:379-381 with stiffness FLT_MAX, motor 0 */
                const float f = rclamp(desired, c.fmin[k], c.fmax[k]);
                const f3 gyro = G[k] * fabsf(f);
                const bool penOk = (c.penalty[k] > 0.0f) & (fabsf(c.penalty[k]) <= FLT_MAX);   /* solver.cpp:387 */
                const int r = lane * 3 + k;
                S[r] = f; S[kResRows + r] = c.penalty[k]; S[2 * kResRows + r] = gyro.x; S[3 * kResRows + r] = gyro.y; S[4 * kResRows + r] = gyro.z;
                S[5 * kResRows + r] = penOk ? 1.0f : 0.0f;
                if (JOINTS) S[7 * kResRows + r] = 1.0f;
                rowsOn &= penOk;
            }
        }
        const bool allOn = __all(rowsOn);
        wave_sync();
        AVBD_MARK(accumulate);
        RS_STAMP(1);
        if (allOn) accumulate_products<true, JOINTS, MULTI>(a, pr, prLine, S, nEnt * 3, J, X.prStride); else accumulate_products<false, JOINTS, MULTI>(a, pr, prLine, S, nEnt * 3, J, X.prStride);
        if (lane < 27) accs[lane] = a.acc;
        wave_sync();
        AVBD_MARK(solve);
        RS_STAMP(2);
        f3 p; q4 q;
#ifdef AVBD_DF_STATS
        if (TWOSTAGE) stat[15] += preDone ? 1 : 0;                   /* tools/two_stage_check.py */
#endif
        if (TWOSTAGE && preDone) solve_pose_post(accs, accs + 32, accs + 64, lane, L, p, q);
        else solve_pose(accs, accs + 32, lane, L, p, q);
#ifdef AVBD_DF_STATS
        race_check_publish(errWord - 2 * B.cap, cntW, body, other, has && moving && !(JOINTS && isJoint), it);
#endif
        if (lane == 0) {
            sanitize3(p, cntW, warn_key(1, it, body, 0), body, WK_POS);
            sanitizeq(q, cntW, warn_key(1, it, body, 1), body, WK_ROT);
            const int o = body * 48;
            v4u r0 = {fbits(p.x), fbits(p.y), fbits(p.z), stampNow}, r1 = {fbits(q.x), fbits(q.y), fbits(q.z), stampNow}, r2 = {fbits(q.w), 0u, 0u, stampNow};
#ifdef AVBD_DF_STATS
            r2.y = (unsigned)__builtin_amdgcn_s_memrealtime();
            trec[3] = r2.y;
#endif
            if (COLO) { volatile v4u AVBD_LDS* r = CL.rec + CL.self * 3; r[0] = r0; r[1] = r1; r[2] = r2; }
            xst(rP, o, r0); xst(rP, o + 16, r1); xst(rP, o + 32, r2);
            if (COLO) CL.flag[CL.self] = stampNow;                    /* (behind the three records: a wave's LDS accesses execute in order, no wait in between) */
        }
        /* lane 0's result becomes the wave's (scalar) copy of the pose */
        L.pos = mk3(bcast0(p.x), bcast0(p.y), bcast0(p.z)); L.rot = mkq(bcast0(q.x), bcast0(q.y), bcast0(q.z), bcast0(q.w));
        AVBD_MARK(dual);
        RS_STAMP(3);

        /* ---- dual update of the joint halves this body owns (solver.cpp:411-430 with Joint rows) ---- */
        if (JOINTS && doDual && has && owned && isJoint) {
            JointRows jr2; jr2.rA = c.rA; jr2.rB = c.rB; jr2.q0 = jq0;
            const q4 rotA = isA ? L.rot : orot, rotB = isA ? orot : L.rot;
            if (isA) joint_compute_constraint(jr2, hasA, L.pos, L.rot, opos, orot);
            else     joint_compute_constraint(jr2, hasA, opos, orot, L.pos, L.rot);
            for (int k = 0; k < 3; ++k) {
                const int row = jPart * 3 + k;
                c.C[k] = jr2.C[row];
                if (comp3(jStiff, k) != FLT_MAX) continue;
                const float lu = rclamp(c.penalty[k] * c.C[k] + c.lambda[k], -FLT_MAX, FLT_MAX);
                const bool active = lu > -FLT_MAX && lu < FLT_MAX;
                c.lambda[k] = lu;
                if (active) {
                    float lw = 0.0f, aw = 0.0f;
                    f3 Jl, Ja;
                    if (hasA) { joint_compute_derivatives(jr2, true, false, rotA, row, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja); }
                    joint_compute_derivatives(jr2, false, false, rotB, row, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja);
                    const float betaRow = row_penalty_gain(lw, aw, P.beta);
                    c.penalty[k] = rmin(c.penalty[k] + betaRow * fabsf(c.C[k]), kPenaltyMax);
                }
            }
            if (moving) {
                v4u r0 = {fbits(c.lambda[0]), fbits(c.lambda[1]), fbits(c.lambda[2]), stampNow};
                v4u r1 = {fbits(c.penalty[0]), fbits(c.penalty[1]), fbits(c.penalty[2]), stampNow << 1};
                xst(rD, offD, r0); xst(rD, offD + 16, r1);
            }
        }
        /* ---- dual update of the contacts this body owns (solver.cpp:411-430) ---- */
        if (doDual && has && owned && !(JOINTS && isJoint)) {
            const f3 wNew = qrotate(L.rot, isA ? c.rA : c.rB);
            const f3 pNew = L.pos + wNew;
            dual_contact_pts(c, isA ? pNew : pOther, isA ? wNew : wOther, isA ? L.invMass : oinv, isA ? pOther : pNew, isA ? wOther : wNew,
                             isA ? oinv : L.invMass, fric, P, alpha);
            if (moving) {
                v4u r0 = {fbits(c.lambda[0]), fbits(c.lambda[1]), fbits(c.lambda[2]), stampNow};
                v4u r1 = {fbits(c.penalty[0]), fbits(c.penalty[1]), fbits(c.penalty[2]), (stampNow << 1) | (c.stick ? 1u : 0u)};
                xst(rD, offD, r0); xst(rD, offD + 16, r1);
            }
        }
        AVBD_MARK(task_end);
        RS_STAMP(4);
#ifdef AVBD_DF_STATS
        if (lane == 0) trec[4] = (unsigned)tprev;
        stat[5] += 1;
#endif
        return 1;
    };
    auto run_task = [&](int it) __attribute__((always_inline)) -> bool { TaskPrep T; prepare_task(it, T); return finish_task(it, T, 0u) != 0; };

    if constexpr (!MULTI) {
        /* one body for the whole launch (the caller checks w < nd) */
        load_task(w, 0);
        for (int it = itFirst; it < itStop; ++it)
            if (!run_task(it)) return;
    } else {
        /* Several bodies per wave.  In order: t = it * slots + slot, a linear extension of the dependency order.  Out of order
         * (X.outOfOrder): the wave runs whichever of its bodies has its inputs published — the dependencies between tasks are
         * carried by the stamped records alone, so any order that respects them gives the same bits — oldest iteration first,
         * so the head of the in-order list (which is what the rest of the launch may be waiting for) is never passed over
         * once it is ready.  Lane k keeps the bookkeeping of the wave's k-th body: its next iteration and ONE record to watch
         * (the input that was missing when the body was last examined); a body is examined in full only when that record
         * has arrived. */
        const int K = w < nd ? (nd - w + W - 1) / W : 0;
        if (!X.outOfOrder || K < 2) {
            for (int it = itFirst; it < itStop; ++it)
                for (int slot = w; slot < nd; slot += W) {
                    load_task(slot, it);
                    RS_STAMP_DRAINED(10);
                    if (!run_task(it)) return;
                    save_task(it == totalIt - 1);
                    RS_STAMP_DRAINED(11);
                }
        } else {
            int bodyK = 0, eBegK = 0, nEntK = 0, itK = itStop, blkOff = -1, seenK = 0;
            unsigned blkStamp = 0;
            if (lane < K) {
                bodyK = order[w + lane * W];
                eBegK = incStart[bodyK]; nEntK = incStart[bodyK + 1] - eBegK;
                itK = itFirst;
            }
            int itLow = itFirst;                                        /* no unfinished body of this wave is at an earlier iteration */
            unsigned idle = 0;
            for (;;) {
                const bool active = itK < itStop;
                if (__ballot(active) == 0) break;
                while (__ballot(active && itK == itLow) == 0) ++itLow;
                bool ok = active;
                if (active && blkOff >= 0) { const v4u r = xld(rP, blkOff); ok = stamp_reached(r.w, blkStamp); }
                int it = -1;
                unsigned long long cand = __ballot(ok);
                while (cand) {
                    int t = itLow;
                    unsigned long long m;
                    while ((m = __ballot(ok && itK == t)) == 0) ++t;
                    const int k = __ffsll((long long)m) - 1;
                    load_header(__builtin_amdgcn_readlane(bodyK, k), __builtin_amdgcn_readlane(eBegK, k), __builtin_amdgcn_readlane(nEntK, k), __builtin_amdgcn_readlane(seenK, k) != 0);
                    if (lane == k) seenK = 1;
                    /* every moving partner's pose of the iteration this visit reads: published?  (The visit / dual records of the
                     * same producer task follow its pose within the task; run_task waits for those.) */
                    unsigned need = 0; bool wants = false;
                    if (has && moving) {
                        if (owned) { wants = true; need = X.base + (unsigned)t + 1u; }
                        else if (t > 0) { wants = true; need = X.base + (unsigned)t; }
                    }
                    bool got = true;
                    if (wants) { const v4u r = xld(rP, offP + 32); got = stamp_reached(r.w, need); }
                    const unsigned long long missing = __ballot(!got);
                    if (missing == 0) {
                        if (lane == k) { ++itK; ++blkStamp; }              /* the same partner is the likeliest to be late again */
                        it = t;
                        break;
                    }
                    const int f = __ffsll((long long)missing) - 1;
                    const int bo = __builtin_amdgcn_readlane(offP + 32, f);
                    const unsigned bs = (unsigned)__builtin_amdgcn_readlane((int)need, f);
                    if (lane == k) { blkOff = bo; blkStamp = bs; ok = false; }
                    cand = __ballot(ok);
                }
                if (it < 0 && X.patience) {
                    /* nothing can run: take the oldest task, do its first half and wait for its inputs — but only for a while.
                     * If they arrive the hand-off is as short as in the in-order sweep; if not, the other bodies get their turn. */
                    const int k = __ffsll((long long)__ballot(active && itK == itLow)) - 1;
                    load_header(__builtin_amdgcn_readlane(bodyK, k), __builtin_amdgcn_readlane(eBegK, k), __builtin_amdgcn_readlane(nEntK, k),
                                __builtin_amdgcn_readlane(seenK, k) != 0);
                    if (lane == k) seenK = 1;
                    load_rest(itLow);
                    TaskPrep T;
                    prepare_task(itLow, T);
                    const int r = finish_task(itLow, T, (unsigned)X.patience);
                    if (r == 0) return;
                    if (r == 2) { if (lane == k) { blkOff = tmoOff; blkStamp = tmoStamp; } }
                    else {
                        if (lane == k) { ++itK; ++blkStamp; }
                        save_task(itLow == totalIt - 1);
                        continue;
                    }
                }
                if (it < 0) {
                    __builtin_amdgcn_s_sleep(4);
                    if ((++idle & 1023u) == 0) {
                        bool dead = (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTimeoutTicks) ||
                                    __hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (__any(dead)) { if (lane == 0) __hip_atomic_store(errWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                    }
                    continue;
                }
                RS_STAMP(9);
                load_rest(it);
                RS_STAMP_DRAINED(10);
                if (!run_task(it)) return;
                save_task(it == totalIt - 1);
                RS_STAMP_DRAINED(11);
            }
        }
    }
#ifdef AVBD_DF_STATS
    if (lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(errWord + 16) + (size_t)(blockIdx.x * kIterWaves + (threadIdx.x >> 6)) * 16;
        stat[6] = __builtin_amdgcn_s_memrealtime() - t0;
        stat[8] = (unsigned long long)nEnt;
        for (int i = 0; i < 16; ++i) o[i] = stat[i];
    }
#endif

    /* ---- once per step: final state back to the tables (plain stores; the next kernel of the stream reads them) ---- */
    if (!MULTI) {
        if (lane == 0) {
            B.plane(BP_POSR)[body] = pack3(L.pos, radius);
            B.plane(BP_QUAT)[body] = packq(L.rot);
            if (COLO && X.crit) X.crit[body] = make_int2(critOther, (int)(unsigned)(__builtin_amdgcn_s_memrealtime() - t0));
        }
        if (has && owned) { if (JOINTS && isJoint) store_joint_half(gf[idx >> 1], jPart, c, true); else store_contact_dual<false>(M, idx, c, true); }
    }
}

} // namespace avbd
