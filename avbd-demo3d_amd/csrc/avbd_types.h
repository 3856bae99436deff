/*
 * avbd_types.h — device-resident state of one solver context (SoA, sized for HBM3E).
 *
 * Replaces the reference's heap objects and intrusive lists (solver.h:47-109, force.cpp:12-69):
 *   Rigid   (264 B object, pointer-chased)  → 18 float4 planes indexed by creation index
 *   Manifold(704 B object)                  → per-manifold header arrays + per-contact float4 planes
 *   body->forces list (nextA/nextB walk)    → CSR adjacency + per-dynamic-body row incidence list,
 *                                              ordered newest-first by a 64-bit creation key
 * All orders that influence results in the reference (broadphase creation order, Gauss-Seidel body
 * order, per-body force order) are functions of creation order, which the key encodes, so the
 * physical placement of a manifold in these arrays is free.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace avbd {

/* float4 planes of the body table */
enum BodyPlane : int {
    BP_POSR = 0,   /* position.xyz, radius            (solver.h:56,69) */
    BP_QUAT,       /* orientation x,y,z,w             (solver.h:57) */
    BP_LINV,       /* linearVelocity.xyz              (solver.h:58) */
    BP_ANGV,       /* angularVelocity.xyz */
    BP_PLIN,       /* prevLinearVelocity.xyz          (solver.h:59) */
    BP_PANG,       /* prevAngularVelocity.xyz */
    BP_IPOS,       /* initialPosition.xyz             (solver.h:61) */
    BP_IQUAT,      /* initialOrientation */
    BP_NPOS,       /* inertialPosition.xyz            (solver.h:63) */
    BP_NQUAT,      /* inertialOrientation */
    BP_SIZE,       /* size.xyz, friction              (solver.h:66,69) */
    BP_MASS,       /* mass, invMass, friction, radius (solver.h:67-70) */
    BP_I0, BP_I1, BP_I2,    /* inertiaTensor columns    (solver.h:68) */
    BP_J0, BP_J1, BP_J2,    /* invInertiaTensor columns */
    BP_COUNT
};

struct BodyTable {
    float4* planes; /* BP_COUNT * cap */
    const int* solved; /* [cap] 1 = this context runs the primal visit of the body: dynamic AND owned by this rank.
                        * Everything else (static bodies, bodies owned by another GPU) is a fixed pose during a sweep. */
    int cap;
    int n;
    __host__ __device__ float4* plane(int p) const { return planes + (size_t)p * cap; }
};

/* One buffer of manifolds; the context owns two and ping-pongs them every step (narrow phase
 * reads last step's manifold, writes the surviving one compacted into the other buffer). */
struct ManifoldTable {
    int2*     pair;    /* bodyA (newer body, larger index), bodyB */
    uint64_t* key;     /* creation key, larger = newer (see make_manifold_key) */
    int*      nc;      /* numContacts 1..4 */
    float*    fric;    /* combinedFriction */
    /* per contact, index 4*m + c */
    float4*   cRA;     /* rA.xyz, C0_n */
    float4*   cRB;     /* rB.xyz, C0_t.x */
    float4*   cN;      /* normal.xyz, C0_t.y */
    float4*   cU1;     /* memo: tangent1 of the stored normal (contact_basis_memo) */
    float4*   cU2;     /* memo: tangent2 of the stored normal */
    float4*   cLam;    /* lambda n,t1,t2, penetration */
    float4*   cPen;    /* penalty n,t1,t2, (unused) */
    int2*     cFS;     /* feature key, stick */
    float4*   cC;      /* last computeConstraint outputs: C n,t1,t2 */
    float4*   cFmin;
    float4*   cFmax;
    int cap;
};

/* Processing order of the narrow phase (k_narrowphase, scenes beyond the fused front end).  A wave whose 64 lanes hold edge contacts, one-point and
 * four-point face manifolds side by side runs every path one after the other; the survivors of step s are therefore listed by CLASS — what the manifold
 * came out as: 3-4 contact points / 2 / 1 on a face / an edge contact — and step s + 1 walks last step's manifolds class by class (most keep their class
 * from step to step), heaviest class first.  Only the ORDER in which lanes pick manifolds changes: slots, keys and every result are what they were. */
constexpr int kNarrowClasses = 4;
constexpr int kNarrowCntStride = 32;                /* ints between two class counters: a 128-byte line each (atomics on one line retire one after the other) */
struct NarrowOrder {
    const int* permCur;   /* [kNarrowClasses][cap]: slots of last step's survivors by class; nullptr: no order (first step, after an upload) */
    const int* cntCur;    /* [kNarrowClasses * kNarrowCntStride] */
    int*       permNext;
    int*       cntNext;   /* zeroed by k_begin_step */
    int        cap;
};

/* Joint / Spring / IgnoreCollision: few, host-created, kept AoS. */
struct GForce {
    int kind, bodyA, bodyB, pad;
    uint64_t key;
    float rA[3], rB[3], q0[4], restLength;
    float C[6], fmin[6], fmax[6], lambda[6], penalty[6], stiffness[6], motor[6];
};

/* Rows of a user-defined Force as its virtuals (getRowCount / computeConstraint / computeDerivatives, solver.h:102-108) left
 * them on the host right before the visit that reads them; GForce.pad = ordinal | rows << 24. */
struct HostRowsDev {
    int rows, pad[3];
    float C[12], fmin[12], fmax[12], lambda[12], penalty[12], stiffness[12], motor[12];
    float JlA[12][3], JaA[12][3], JlB[12][3], JaB[12][3];
};
__host__ __device__ inline int host_force_ordinal(const GForce& f) { return f.pad & 0xFFFFFF; }
__host__ __device__ inline int host_force_rows(const GForce& f) { return (f.pad >> 24) & 0xFF; }
/* incidence entries (3 rows each) a Joint / Spring / IgnoreCollision / host force contributes to each of its swept endpoints */
__host__ __device__ inline int gforce_entries(const GForce& f)
{
    return f.kind == 1 ? 2 : (f.kind == 2 ? 1 : (f.kind == 4 ? (host_force_rows(f) + 2) / 3 : 0));
}
/* kind of a GForce inside an adjacency / incidence reference (two bits): a host force looks like a spring there */
__host__ __device__ inline int gforce_ref_kind(int kind) { return kind == 4 ? 2 : kind; }

struct Counters {
    int numManifolds;   /* manifolds alive in the current buffer */
    int numCandidates;  /* broadphase output this step */
    int overflow;       /* bit0: candidate pool, bit1: manifold pool, bit2: incidence pool */
    int numLevels;
    int numContacts;
    int numIncidences;
    int warnings;
    int dynamicBodies;
    int levelsConverged;
    int residentBad;    /* bit0: a swept body has > 64 incidence entries or a non-Manifold entry, bit1: more swept bodies than resident waves */
    int pad[6];
};

/* What the reference prints when it resets a non-finite quantity (solver.cpp:51-66), kept for the host to print after the
 * step: the first kWarnCap hits of a step, right behind the Counters word they are counted in.  `key` orders them the way
 * the reference's sequential loops would have printed them: phase (predict / iteration sweep / velocity update), iteration,
 * position in the body list (newest first), statement. */
struct WarnRec { unsigned key; int body; float x, y, z; int kind; int pad0, pad1; };
constexpr int kWarnCap = 128;
enum WarnKind : int { WK_LINVEL = 0, WK_ANGVEL, WK_PRED_POS, WK_PRED_ROT, WK_POS, WK_ROT };
__host__ __device__ inline unsigned warn_key(int phase, int it, int body, int sub)
{
    return ((unsigned)phase << 28) | ((unsigned)(it & 63) << 22) | ((0xFFFFFu - (unsigned)body) << 2) | (unsigned)sub;
}

struct DiagAccum { /* uint views of non-negative floats so atomicMax orders them */
    unsigned maxPenetration, maxViolation, maxLinearSpeed, maxAngularSpeed, maxNormalImpulse;
    int activeContacts, activeManifolds, dynamicBodies;
};

struct SolverParams {
    float dt; float gx, gy, gz; int iterations; float alpha, beta, gamma; int postStabilize;
};

/* 64-bit creation key: [63:41] stepIndex at creation, [40] host-created flag, [39:0] payload.
 * Broadphase (solver.cpp:262-270) creates manifolds for A descending, B descending (< A); a later
 * creation is "newer" and sits earlier in every intrusive list, so larger key = earlier in lists. */
__host__ __device__ inline uint64_t make_manifold_key(int step, int bodyA, int bodyB)
{
    const uint64_t M = (1u << 20) - 1;
    return ((uint64_t)(unsigned)step << 41) | ((M - (uint64_t)bodyA) << 20) | (M - (uint64_t)bodyB);
}
__host__ __device__ inline uint64_t make_host_key(int step, uint64_t counter)
{
    return ((uint64_t)(unsigned)step << 41) | (1ull << 40) | (counter & ((1ull << 40) - 1));
}

/* incidence entry: bit31 = body is endpoint A, bits30:29 = kind, bits28:0 = index
 * (manifold: contact slot 4*m+c; joint: 2*g+part (part 0 = linear rows 0-2, 1 = angular rows 3-5);
 * spring: g) */
__host__ __device__ inline int make_inc(int isA, int kind, int index) { return (isA << 31) | (kind << 29) | index; }
__host__ __device__ inline int inc_isA(int e) { return (e >> 31) & 1; }
__host__ __device__ inline int inc_kind(int e) { return (e >> 29) & 3; }
__host__ __device__ inline int inc_index(int e) { return e & ((1 << 29) - 1); }

} // namespace avbd
