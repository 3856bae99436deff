/*
 * avbd_rows.h — the row-based Force API (solver.h:85-109) as device functions.
 *
 * The reference exposes every constraint kind through four virtuals:
 *   getRowCount() / initialize() / computeConstraint(alpha) / computeDerivatives(Jl, Ja, body, row)
 * The solver kernels (avbd_kernels.hip) are written against exactly that interface; what is
 * virtual dispatch in the reference is a switch over `ForceKind` here, and what is a 704-byte
 * heap object there is a register-resident view (`ContactRows`, `JointRows`, `SpringRows`) loaded
 * from / stored to the SoA arrays of avbd_types.h by the calling kernel.
 *
 * A Manifold is evaluated contact by contact (3 rows each): the reference's per-contact loop
 * bodies in manifold.cpp:177-245 / 247-271 are independent of each other, so one lane per contact
 * reproduces the whole-manifold call exactly, including its in-place mutations (normal
 * re-normalisation, friction-cone rescale of lambda, stick flag).
 */
#pragma once
#include "avbd_math.h"
#include "avbd_collide.h"

#pragma clang fp contract(off)

namespace avbd {

constexpr int   kMaxRows            = 12;           /* solver.h:25 */
constexpr float kPenaltyMin         = 20000.0f;     /* solver.h:26 */
constexpr float kPenaltyMax         = 1000000000.0f;/* solver.h:27 */
constexpr float kStickThresh        = 0.02f;        /* solver.h:29 */
constexpr float kPenetrationSlop    = 0.005f;       /* solver.h:36 */
constexpr float kManifoldPenaltyCap = 2000000.0f;   /* solver.cpp:29, manifold.cpp:23 */
constexpr float kNormalMargin       = 0.01f;        /* manifold.cpp:17 */
constexpr float kStickMaxDrift      = 0.015f;       /* manifold.cpp:18 */
constexpr float kStickMinDot        = 0.995f;       /* manifold.cpp:19 */
constexpr float kWarmMaxDrift       = 0.08f;        /* manifold.cpp:20 */
constexpr float kWarmMinDot         = 0.9f;         /* manifold.cpp:21 */
constexpr float kNormalForceCap     = 5000.0f;      /* manifold.cpp:22 */

enum ForceKind : int { FK_MANIFOLD = 0, FK_JOINT = 1, FK_SPRING = 2, FK_IGNORE = 3,
                       FK_HOST = 4 /* a user-defined Force subclass (solver.h:85-109): its virtuals run on the host, the device gets rows.
                                    * Exists as GForce.kind only; incidence / adjacency references use FK_SPRING with the part in bits 24+ */ };

/* getRowCount(): manifold.cpp:66-69, joint.h:42, spring.h:33, ignorecollision.h:19 */
AVBD_HD int force_row_count(int kind, int numContacts)
{
    return kind == FK_MANIFOLD ? numContacts * 3 : (kind == FK_JOINT ? 6 : (kind == FK_SPRING ? 1 : 0));
}

AVBD_HD f3 world_point(f3 pos, q4 rot, f3 local) { return pos + qrotate(rot, local); }   /* manifold.cpp:25-28 */

AVBD_HD f3 normalize_safe(f3 v, f3 fallback)                                             /* manifold.cpp:30-37 */
{
    float lenSq = lensq3(v);
    if (lenSq < kVecEps) return fallback;
    return v / sqrtf(lenSq);
}

AVBD_HD void contact_basis(f3 nIn, f3& n, f3& t1, f3& t2)                                /* manifold.cpp:39-50 */
{
    n = normalize_safe(nIn, mk3(0.0f, 1.0f, 0.0f));
    if (fabsf(n.x) >= fabsf(n.z)) t1 = mk3(-n.y, n.x, 0.0f);
    else t1 = mk3(0.0f, -n.z, n.y);
    t1 = normalize_safe(t1, mk3(1.0f, 0.0f, 0.0f));
    t2 = normalize_safe(cross3(n, t1), mk3(0.0f, 0.0f, 1.0f));
}

/* Tangents of an ALREADY NORMALISED normal: the second half of buildContactBasis (manifold.cpp:43-49). */
AVBD_HD void contact_tangents(f3 n, f3& t1, f3& t2)
{
    if (fabsf(n.x) >= fabsf(n.z)) t1 = mk3(-n.y, n.x, 0.0f);
    else t1 = mk3(0.0f, -n.z, n.y);
    t1 = normalize_safe(t1, mk3(1.0f, 0.0f, 0.0f));
    t2 = normalize_safe(cross3(n, t1), mk3(0.0f, 0.0f, 1.0f));
}

/* buildContactBasis(stored) with a memo: (u1, u2) are the tangents contact_tangents(stored) computed when
 * `stored` was written.  The reference re-normalises the stored normal on every call; when that
 * normalisation is the identity on these bits (|stored| rounds to exactly 1.0f — the usual case after the first
 * call of a step) the whole basis equals the memo bit for bit, because it is a pure function of the same input.
 * Otherwise the basis is recomputed exactly like the reference does and `hit` is false. */
/* BITS: the test `sqrtf(lenSq) == 1.0f` without the square root.  The correctly rounded root of a float is 1.0f for exactly two floats, 1.0f itself and the one
 * above it (sqrt(1 + 2^-23) = 1 + 2^-24 - 2^-49.. rounds down to 1; sqrt(1 - 2^-24) = 1 - 2^-25 - 2^-51.. lies below the midpoint and rounds to 1 - 2^-24;
 * sqrt(1 + 2^-22) rounds to 1 + 2^-23), both far above kVecEps — so the test is `(bits | 1) == 0x3F800001`.  An IEEE square root is ~15 dependent instructions and
 * the test runs in every computeConstraint call.  Round 5 measured it everywhere (bit-identical, 90 GPU tests): 64 000 cubes 4.41 -> 4.35 ms per launch, but
 * Stress1000 0.6147 -> 0.6305 ms per step and 8 000 cubes +0.6 % — fewer instructions, a worse schedule of the kernels that sit at their register limit — so only
 * the several-bodies kernels of the group schedule take this form (iterate_groups<MULTI = true>). */
template <bool BITS = false>
AVBD_HD void contact_basis_memo(f3 stored, f3 u1, f3 u2, f3& n, f3& t1, f3& t2, bool& hit)
{
    const float lenSq = lensq3(stored);
    bool unit;
    if (BITS) { unsigned lb; { float t = lenSq; __builtin_memcpy(&lb, &t, 4); } unit = (lb | 1u) == 0x3F800001u; }
    else unit = !(lenSq < kVecEps) && sqrtf(lenSq) == 1.0f;
    if (unit) { n = stored; t1 = u1; t2 = u2; hit = true; return; } /* stored / 1.0f == stored */
    hit = false;
    n = normalize_safe(stored, mk3(0.0f, 1.0f, 0.0f));
    contact_tangents(n, t1, t2);
}

/* ---- one contact of a Manifold = 3 rows (normal, tangent1, tangent2) ------------------- */
struct ContactRows {
    f3    rA, rB, normal;        /* Manifold::Contact, solver.h:120-130 */
    f3    u1, u2;                /* memo: contact_tangents(normal) for the stored normal (build-only, see contact_basis_memo) */
    float C0n, C0t1, C0t2;
    float penetration;
    bool  stick;
    float lambda[3], penalty[3]; /* Force::lambda/penalty rows base..base+2 */
    float C[3], fmin[3], fmax[3];/* outputs of computeConstraint */
};

/* Manifold::computeConstraint for one contact (manifold.cpp:183-243).  Mutates normal, lambda[1..2],
 * stick, penetration exactly like the reference. */
struct ContactBasis { f3 n, t1, t2; };

/* `db` receives the basis the reference's computeDerivatives would rebuild right after this call from the normal
 * this call stores (manifold.cpp:253-254): the same memo rule applied to the new stored normal. */
/* computeConstraint is evaluated in two halves so that a caller can run the first one before the partner's state is known:
 *   pre  (manifold.cpp:185-187, 200-204): re-normalised normal + tangents, bias terms, normal-force cap — functions of the
 *        contact's own stored state, alpha and the two inverse masses only;
 *   post (manifold.cpp:188-243): everything that involves the two world-space contact points. */
struct ContactPre { f3 n, t1, t2; float biasN, biasT1, biasT2, cap; };

template <bool BITS = false>
AVBD_HD void contact_constraint_pre(ContactRows& c, float alpha, float invMassA, float invMassB, ContactPre& pre, ContactBasis& db)
{
    float biasScale = rclamp(1.0f - alpha, 0.0f, 1.0f);
    bool hit;
    contact_basis_memo<BITS>(c.normal, c.u1, c.u2, pre.n, pre.t1, pre.t2, hit);
    c.normal = pre.n; c.u1 = pre.t1; c.u2 = pre.t2;   /* stored normal and its memo move together */
    if (hit) { db.n = pre.n; db.t1 = pre.t1; db.t2 = pre.t2; }            /* same bits in, same basis out */
    else { bool h2; contact_basis_memo<BITS>(pre.n, pre.t1, pre.t2, db.n, db.t1, db.t2, h2); }
    pre.biasN = biasScale * c.C0n;
    pre.biasT1 = biasScale * c.C0t1;
    pre.biasT2 = biasScale * c.C0t2;
    float invMassSum = invMassA + invMassB;
    float massScale = (invMassSum > 1.0e-6f) ? (1.0f / invMassSum) : 1.0f;
    pre.cap = kNormalForceCap * massScale;
}

/* core on the two world-space contact points pA = posA + rotate(qA, rA), pB = posB + rotate(qB, rB) (manifold.cpp:188-190) */
AVBD_HD void contact_constraint_post(ContactRows& c, const ContactPre& pre, f3 pA, f3 pB, float combinedFriction)
{
    f3 d = pA - pB;
    float separation = dot3(d, pre.n) - kNormalMargin;
    float slip1 = dot3(d, pre.t1);
    float slip2 = dot3(d, pre.t2);

    c.C[0] = separation + pre.biasN;
    float cap = pre.cap;
    c.fmin[0] = -cap;
    c.fmax[0] = 0.0f;

    c.C[1] = slip1 + pre.biasT1;
    c.C[2] = slip2 + pre.biasT2;

    float warmMag = fabsf(rmin(c.lambda[0], 0.0f));
    float trial = c.penalty[0] * c.C[0] + c.lambda[0];
    float trialMag = fabsf(rmin(trial, 0.0f));
    float nMag = rmax(warmMag, trialMag);
    nMag = rmin(nMag, cap);

    float mu = combinedFriction;
    if (!c.stick) mu *= 0.9f;
    float limit = mu * nMag;

    float lt1 = c.lambda[1], lt2 = c.lambda[2];
    float tanMag = sqrtf(lt1 * lt1 + lt2 * lt2);
    if (tanMag > limit && tanMag > 1.0e-8f) {
        float s = limit / tanMag;
        c.lambda[1] *= s;
        c.lambda[2] *= s;
    }
    c.fmin[1] = -limit; c.fmax[1] = limit;
    c.fmin[2] = -limit; c.fmax[2] = limit;

    float slipSq = c.C[1] * c.C[1] + c.C[2] * c.C[2];
    float tanSq = c.lambda[1] * c.lambda[1] + c.lambda[2] * c.lambda[2];
    float stickSq = kStickThresh * kStickThresh;
    c.stick = (slipSq <= stickSq) && (tanSq <= limit * limit + 1.0e-8f);
    c.penetration = rmax(0.0f, -dot3(d, pre.n));
}

template <bool BITS = false>
AVBD_HD void contact_compute_constraint_pts(ContactRows& c, float alpha, f3 pA, f3 pB, float invMassA, float invMassB,
                                            float combinedFriction, ContactBasis& db)
{
    ContactPre pre;
    contact_constraint_pre<BITS>(c, alpha, invMassA, invMassB, pre, db);
    contact_constraint_post(c, pre, pA, pB, combinedFriction);
}

AVBD_HD void contact_compute_constraint(ContactRows& c, float alpha, f3 posA, q4 rotA, float invMassA, f3 posB, q4 rotB,
                                        float invMassB, float combinedFriction, ContactBasis& db)
{
    contact_compute_constraint_pts(c, alpha, world_point(posA, rotA, c.rA), world_point(posB, rotB, c.rB), invMassA, invMassB, combinedFriction, db);
}

/* computeDerivatives from the world-space lever arm wr = rotate(q_body, r_local) (manifold.cpp:265-270) */
AVBD_HD void contact_compute_derivatives_w(const ContactBasis& db, bool isBodyA, f3 wr, f3 Jl[3], f3 Ja[3])
{
    float sign = isBodyA ? 1.0f : -1.0f;
    Jl[0] = db.n * sign;  Ja[0] = cross3(wr, db.n) * sign;
    Jl[1] = db.t1 * sign; Ja[1] = cross3(wr, db.t1) * sign;
    Jl[2] = db.t2 * sign; Ja[2] = cross3(wr, db.t2) * sign;
}

/* Manifold::computeDerivatives for the three rows of one contact (manifold.cpp:247-271).  The
 * reference rebuilds the basis from the STORED normal on every call; the stored normal is the
 * one computeConstraint just wrote, so the basis below is normalised once more, as there. */
AVBD_HD void contact_compute_derivatives(const ContactRows& c, const ContactBasis& db, bool isBodyA, q4 rotBody, f3 Jl[3], f3 Ja[3])
{
    contact_compute_derivatives_w(db, isBodyA, qrotate(rotBody, isBodyA ? c.rA : c.rB), Jl, Ja);
}

/* stand-alone form (row API): rebuild the basis from the stored normal like manifold.cpp:253-254 */
AVBD_HD void contact_compute_derivatives(const ContactRows& c, bool isBodyA, q4 rotBody, f3 Jl[3], f3 Ja[3])
{
    ContactBasis db; bool hit;
    contact_basis_memo(c.normal, c.u1, c.u2, db.n, db.t1, db.t2, hit);
    contact_compute_derivatives(c, db, isBodyA, rotBody, Jl, Ja);
}

/* ---- Manifold::initialize (manifold.cpp:71-175) ----------------------------------------- */
struct ManifoldInit {
    int   numContacts;
    float combinedFriction;
    int   feature[4];
    f3    rA[4], rB[4], normal[4];
    f3    u1[4], u2[4];          /* tangent memo of the stored (normalised) normal */
    float penetration[4], C0n[4], C0t1[4], C0t2[4];
    bool  stick[4];
    float lambda[12], penalty[12];
};

/* the previous step's contacts while the new ones are matched against them (manifold.cpp:76-84); a separate object so
 * that a kernel can keep it (and ManifoldInit) in LDS: both are indexed dynamically and would otherwise live in scratch */
struct ManifoldOld {
    RawContact raw[4];           /* what collide_boxes found this step */
    CollideScratch ws;
    int   feature[4];
    f3    rA[4], rB[4], normal[4];
    bool  stick[4], used[4];
    /* (last step's lambda / penalty stay in registers while the pair is re-collided — manifold_initialize — : 96 B of LDS per pair less) */
};

/* `m` holds the previous step's contacts on entry (numContacts = 0 for a fresh manifold) and the
 * new manifold on exit.  Returns false when the pair no longer touches (the reference deletes it). */
AVBD_HD bool manifold_initialize(ManifoldInit& m, ManifoldOld& old, const BoxPose& A, float frictionA, const BoxPose& B, float frictionB)
{
    m.combinedFriction = sqrtf(frictionA * frictionB);

    int   oldN = m.numContacts;
    int* oldFeature = old.feature; f3* oldRA = old.rA; f3* oldRB = old.rB; f3* oldNormal = old.normal; bool* oldStick = old.stick;
    float oldLambda[12], oldPenalty[12];                               /* registers: read through sel4 below, never indexed dynamically */
    bool* oldUsed = old.used;
    for (int i = 0; i < 4; ++i) oldUsed[i] = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < oldN) {
            oldFeature[i] = m.feature[i]; oldRA[i] = m.rA[i]; oldRB[i] = m.rB[i]; oldNormal[i] = m.normal[i]; oldStick[i] = m.stick[i];
        }
    }
#pragma unroll
    for (int r = 0; r < 12; ++r) { oldLambda[r] = m.lambda[r]; oldPenalty[r] = m.penalty[r]; }
    auto sel4 = [](int j, float a, float b, float c, float d) { return j == 0 ? a : (j == 1 ? b : (j == 2 ? c : d)); };

    RawContact* raw = old.raw;
    m.numContacts = collide_boxes(A, B, raw, old.ws);
    if (m.numContacts == 0) return false;

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i >= m.numContacts) break;
        int base = i * 3;
        m.feature[i] = raw[i].feature; m.rA[i] = raw[i].rA; m.rB[i] = raw[i].rB; m.normal[i] = raw[i].normal;
        for (int k = 0; k < 3; ++k) { m.lambda[base + k] = 0.0f; m.penalty[base + k] = kPenaltyMin; }
        m.stick[i] = false;

        int best = -1;
        for (int j = 0; j < 4; ++j) {
            if (j >= oldN) break;
            if (oldUsed[j]) continue;
            if (m.feature[i] == oldFeature[j]) { best = j; break; }
        }
        if (best >= 0) {
            oldUsed[best] = true;
            f3 newN = normalize_safe(m.normal[i], mk3(0.0f, 1.0f, 0.0f));
            f3 oldNn = normalize_safe(oldNormal[best], newN);
            float nd = dot3(newN, oldNn);
            f3 oldMid = (world_point(A.pos, A.rot, oldRA[best]) + world_point(B.pos, B.rot, oldRB[best])) * 0.5f;
            f3 newMid = (world_point(A.pos, A.rot, m.rA[i]) + world_point(B.pos, B.rot, m.rB[i])) * 0.5f;
            float driftSq = lensq3(newMid - oldMid);
            bool valid = (nd >= kWarmMinDot) && (driftSq <= kWarmMaxDrift * kWarmMaxDrift);
            if (valid) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    m.lambda[base + k] = sel4(best, oldLambda[k], oldLambda[3 + k], oldLambda[6 + k], oldLambda[9 + k]);
                    m.penalty[base + k] = rclamp(sel4(best, oldPenalty[k], oldPenalty[3 + k], oldPenalty[6 + k], oldPenalty[9 + k]), kPenaltyMin, kManifoldPenaltyCap);
                }
            }
            bool reuse = false;
            if (oldStick[best] && valid)
                reuse = (nd >= kStickMinDot) && (driftSq <= kStickMaxDrift * kStickMaxDrift);
            m.stick[i] = oldStick[best] && reuse;
            if (reuse) { m.rA[i] = oldRA[best]; m.rB[i] = oldRB[best]; }
        }

        f3 n, t1, t2;
        contact_basis(m.normal[i], n, t1, t2);
        m.normal[i] = n; m.u1[i] = t1; m.u2[i] = t2;
        f3 pA = world_point(A.pos, A.rot, m.rA[i]);
        f3 pB = world_point(B.pos, B.rot, m.rB[i]);
        f3 d = pA - pB;
        m.C0n[i] = dot3(d, n) - kNormalMargin;
        m.C0t1[i] = dot3(d, t1);
        m.C0t2[i] = dot3(d, t2);
        m.penetration[i] = rmax(0.0f, -dot3(d, n));
    }
    return true;
}

AVBD_HD bool manifold_initialize(ManifoldInit& m, const BoxPose& A, float frictionA, const BoxPose& B, float frictionB)
{
    ManifoldOld old;
    return manifold_initialize(m, old, A, frictionA, B, frictionB);
}

/* ---- Joint (6 rows) — joint.cpp:68-139, bug-compatible angular Jacobian sign --- */
struct JointRows {
    f3 rA, rB; q4 q0;
    float C[6];
    bool fixAngular = false;     /* non-default option (avbd_set_joint_angular_sign): angular rows with the sign that matches C */
};

AVBD_HD void joint_compute_constraint(JointRows& j, bool hasA, f3 posA, q4 rotA, f3 posB, q4 rotB)  /* joint.cpp:68-106 */
{
    f3 pA; q4 qA;
    if (hasA) { qA = rotA; pA = posA + qrotate(qA, j.rA); }
    else { qA = identq(); pA = j.rA; }
    f3 pB = posB + qrotate(rotB, j.rB);
    f3 lc = pA - pB;
    j.C[0] = lc.x; j.C[1] = lc.y; j.C[2] = lc.z;
    q4 cur = qconj(qA) * rotB;
    q4 dq = cur * qconj(j.q0);
    f3 ac = mk3(dq.x, dq.y, dq.z) * 2.0f;
    j.C[3] = ac.x; j.C[4] = ac.y; j.C[5] = ac.z;
    /* fmin/fmax = -/+FLT_MAX for all six rows (joint.cpp:101-105) */
}

AVBD_HD void joint_compute_derivatives(const JointRows& j, bool isBodyA, bool bodyIsWorld, q4 rotBody, int row, f3& Jl, f3& Ja) /* joint.cpp:108-139 */
{
    Jl = zero3(); Ja = zero3();
    float sign = isBodyA ? 1.0f : -1.0f;
    if (isBodyA && bodyIsWorld) return;
    if (row < 3) {
        f3 ax = mk3(row == 0 ? 1.0f : 0.0f, row == 1 ? 1.0f : 0.0f, row == 2 ? 1.0f : 0.0f);
        f3 r = qrotate(rotBody, isBodyA ? j.rA : j.rB);
        Jl = ax * sign;
        Ja = cross3(r, ax) * sign;
    } else {
        int k = row - 3;
        f3 ax = mk3(k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f);
        /* joint.cpp:129-138 as written: +e_k for bodyA, -e_k for bodyB, although C_ang = 2 vec(qA* qB q0*) grows with bodyB's rotation —
         * the reference's sign (bug-compatible default).  The option flips it so that J = dC/dtheta. */
        Ja = ax * (j.fixAngular ? -sign : sign);
    }
}

/* ---- Spring (1 row) — spring.cpp:33-90 -------------------------------------------------- */
struct SpringRows { f3 rA, rB; float restLength; float C; };

AVBD_HD void spring_compute_constraint(SpringRows& s, bool hasA, f3 posA, q4 rotA, f3 posB, q4 rotB)  /* spring.cpp:33-58 */
{
    q4 qA = hasA ? rotA : identq();
    f3 pA = hasA ? posA + qrotate(qA, s.rA) : s.rA;
    f3 pB = posB + qrotate(rotB, s.rB);
    s.C = len3(pA - pB) - s.restLength;
}

AVBD_HD void spring_compute_derivatives(const SpringRows& s, bool isBodyA, bool hasA, f3 posA, q4 rotA, f3 posB, q4 rotB, f3& Jl, f3& Ja) /* spring.cpp:61-90 */
{
    q4 qA = hasA ? rotA : identq();
    f3 pA = hasA ? posA + qrotate(qA, s.rA) : s.rA;
    f3 pB = posB + qrotate(rotB, s.rB);
    f3 d = pA - pB;
    float len = len3(d);
    if (len < kVecEps) { Jl = zero3(); Ja = zero3(); return; }
    f3 n = d / len;
    float sign = isBodyA ? 1.0f : -1.0f;
    Jl = n * sign;
    f3 rw = isBodyA ? qrotate(rotA, s.rA) : qrotate(rotB, s.rB);
    Ja = cross3(rw, n) * sign;
}

/* rowPenaltyGain (solver.cpp:94-125) from the per-endpoint squared Jacobian norms. */
AVBD_HD float row_penalty_gain(float linearWeight, float angularWeight, float beta)
{
    const float angularBetaScale = 0.01f;
    const float epsilon = 1.0e-8f;
    float total = linearWeight + angularWeight;
    if (total < epsilon) return beta;
    float betaLinear = beta;
    float betaAngular = beta * angularBetaScale;
    return (betaLinear * linearWeight + betaAngular * angularWeight) / total;
}

} // namespace avbd
