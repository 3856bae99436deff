/*
 * avbd_primal.h — building blocks of the per-body primal solve (solver.cpp:344-409) and the per-force
 * dual update (solver.cpp:411-430), shared by
 *   - the level-launch kernels k_primal / k_dual (one launch per Gauss-Seidel level; plain loads), and
 *   - the persistent data-flow kernel k_iterate (avbd_iterate.h; every byte that another workgroup may
 *     have written in the same launch is moved with agent-scope `sc1` accesses — template COH=true).
 *
 * One wavefront works on one body:
 *   phase 1  lane e evaluates incidence entry e (<= 3 rows): computeConstraint + computeDerivatives,
 *            stages the row records in LDS, writes the reference's in-place mutations back
 *   phase 2  lanes 0..26 each own one entry of (ll lower, la, aa lower, rhs) and add the row terms
 *            IN LIST ORDER (newest force first, row 0..n) — the reference's summation order
 *   phase 3  lane 0 runs solve6x6 and the pose update
 */
#pragma once
#include "avbd_types.h"
#include "avbd_rows.h"
#include "avbd_util.h"

#pragma clang fp contract(off)

namespace avbd {

/* ---- memory access flavours ---------------------------------------------------------------
 * COH=false: plain global loads/stores (data produced by an earlier launch).
 * COH=true : relaxed agent-scope atomics = global_load/store ... sc1 — bypass this CU's L1 and the
 *            (non-coherent) per-XCD L2 copy; used for hand-offs inside one launch
 *            (cdna_hip_programming.md Guideline 16; MI355X_MICROARCH.md "Valid forms"). */
template <bool COH>
__device__ __forceinline__ float4 ld4(const float4* p)
{
    if constexpr (COH) {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
        unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                           __uint_as_float((unsigned)(b >> 32)));
    } else {
        return *p;
    }
}

template <bool COH>
__device__ __forceinline__ void st4(float4* p, float4 v)
{
    if constexpr (COH) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
        unsigned long long a = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
        unsigned long long b = (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32);
        __hip_atomic_store(q, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *p = v;
    }
}

template <bool COH>
__device__ __forceinline__ int ldi(const int* p)
{
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

template <bool COH>
__device__ __forceinline__ void sti(int* p, int v)
{
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

template <bool COH>
__device__ __forceinline__ float ldf(const float* p)
{
    if constexpr (COH) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else return *p;
}

template <bool COH>
__device__ __forceinline__ void stf(float* p, float v)
{
    if constexpr (COH) __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

template <bool COH>
__device__ __forceinline__ void load_pose(const BodyTable& B, int i, f3& pos, q4& rot)
{
    float4 p = ld4<COH>(B.plane(BP_POSR) + i), q = ld4<COH>(B.plane(BP_QUAT) + i);
    pos = mk3(p.x, p.y, p.z); rot = mkq(q.x, q.y, q.z, q.w);
}

/* ---- one Manifold contact in registers ---------------------------------------------------- */
template <bool COH>
__device__ __forceinline__ void load_contact(const ManifoldTable& M, int s, ContactRows& c)
{
    const float4 ra = M.cRA[s], rb = M.cRB[s];                       /* constant during the iteration loop */
    const float4 nn = ld4<COH>(M.cN + s), lam = ld4<COH>(M.cLam + s), pen = ld4<COH>(M.cPen + s);
    const float4 u1 = ld4<COH>(M.cU1 + s), u2 = ld4<COH>(M.cU2 + s);
    c.u1 = mk3(u1.x, u1.y, u1.z); c.u2 = mk3(u2.x, u2.y, u2.z);
    c.rA = mk3(ra.x, ra.y, ra.z); c.C0n = ra.w;
    c.rB = mk3(rb.x, rb.y, rb.z); c.C0t1 = rb.w;
    c.normal = mk3(nn.x, nn.y, nn.z); c.C0t2 = nn.w;
    c.lambda[0] = lam.x; c.lambda[1] = lam.y; c.lambda[2] = lam.z; c.penetration = lam.w;
    c.penalty[0] = pen.x; c.penalty[1] = pen.y; c.penalty[2] = pen.z;
    c.stick = ldi<COH>(&M.cFS[s].y) != 0;
}

/* the reference's in-place mutations of computeConstraint (manifold.cpp:185-186, 224-231, 241, 243) */
template <bool COH>
__device__ __forceinline__ void store_contact_mutations(const ManifoldTable& M, int s, const ContactRows& c, bool writeRows)
{
    st4<COH>(M.cN + s, make_float4(c.normal.x, c.normal.y, c.normal.z, c.C0t2));
    st4<COH>(M.cU1 + s, make_float4(c.u1.x, c.u1.y, c.u1.z, 0.f));
    st4<COH>(M.cU2 + s, make_float4(c.u2.x, c.u2.y, c.u2.z, 0.f));
    st4<COH>(M.cLam + s, make_float4(c.lambda[0], c.lambda[1], c.lambda[2], c.penetration));
    sti<COH>(&M.cFS[s].y, c.stick ? 1 : 0);
    if (writeRows) { /* API visibility of Force::C/fmin/fmax only; never read back by the solver */
        st4<COH>(M.cC + s, make_float4(c.C[0], c.C[1], c.C[2], 0.f));
        st4<COH>(M.cFmin + s, make_float4(c.fmin[0], c.fmin[1], c.fmin[2], 0.f));
        st4<COH>(M.cFmax + s, make_float4(c.fmax[0], c.fmax[1], c.fmax[2], 0.f));
    }
}

/* ---- per-body constants of one primal visit ------------------------------------------------ */
struct BodyLocal {
    f3 pos; q4 rot; f3 npos; q4 nrot;
    float mass, invMass, invDt2;
    m33 Iw, invIw;
};

template <bool COH>
__device__ __forceinline__ BodyLocal load_body_local(const BodyTable& B, int body, const SolverParams& P)
{
    BodyLocal L;
    load_pose<COH>(B, body, L.pos, L.rot);
    float4 np = B.plane(BP_NPOS)[body], nq = B.plane(BP_NQUAT)[body], mp = B.plane(BP_MASS)[body];
    L.npos = mk3(np.x, np.y, np.z); L.nrot = mkq(nq.x, nq.y, nq.z, nq.w);
    L.mass = mp.x; L.invMass = mp.y;
    float4 i0 = B.plane(BP_I0)[body], i1 = B.plane(BP_I1)[body], i2 = B.plane(BP_I2)[body];
    float4 j0 = B.plane(BP_J0)[body], j1 = B.plane(BP_J1)[body], j2 = B.plane(BP_J2)[body];
    const m33 Ib = mkm(mk3(i0.x, i0.y, i0.z), mk3(i1.x, i1.y, i1.z), mk3(i2.x, i2.y, i2.z));
    const m33 Jb = mkm(mk3(j0.x, j0.y, j0.z), mk3(j1.x, j1.y, j1.z), mk3(j2.x, j2.y, j2.z));
    const m33 R = quat_to_mat(L.rot);
    const m33 Rt = mtranspose(R);
    L.Iw = R * Ib * Rt;       /* rigid.cpp:56-59 */
    L.invIw = R * Jb * Rt;    /* rigid.cpp:51-54 */
    L.invDt2 = 1.0f / (P.dt * P.dt);
    return L;
}

/* World-space inertia and inverse inertia (rigid.cpp:51-59: R * I * transpose(R)), one matrix entry per lane:
 * lanes 0..8 → Iw(i,j), lanes 9..17 → invIw(i,j) with i = l % 3, j = l / 3; entries are exchanged through 18 floats of
 * LDS.  Same expression tree per entry as the reference's mat3 products:
 *   (R*I)(i,k) = (R(i,0) I(0,k) + R(i,1) I(1,k)) + R(i,2) I(2,k);   (M*Rt)(i,j) = (M(i,0) R(j,0) + M(i,1) R(j,1)) + M(i,2) R(j,2) */
__device__ __forceinline__ void world_inertia_parallel(int lane, q4 rot, m33 Ib, m33 Jb, float* xch, m33& Iw, m33& invIw)
{
    const m33 R = quat_to_mat(rot);
    const m33 Rt = mtranspose(R);
    const int e = lane < 9 ? lane : (lane < 18 ? lane - 9 : 0);
    const int i = e % 3, j = e / 3;
    const m33 T = lane < 9 ? Ib : Jb;
    const f3 ri = i == 0 ? Rt.c0 : (i == 1 ? Rt.c1 : Rt.c2);       /* row i of R */
    const f3 rj = j == 0 ? Rt.c0 : (j == 1 ? Rt.c1 : Rt.c2);       /* row j of R = column j of transpose(R) */
    const f3 mi = mk3(dot3(ri, T.c0), dot3(ri, T.c1), dot3(ri, T.c2)); /* row i of R*T */
    const float v = dot3(mi, rj);
    if (lane < 18) xch[lane] = v;
    wave_sync();
    Iw = mkm(mk3(xch[0], xch[1], xch[2]), mk3(xch[3], xch[4], xch[5]), mk3(xch[6], xch[7], xch[8]));
    invIw = mkm(mk3(xch[9], xch[10], xch[11]), mk3(xch[12], xch[13], xch[14]), mk3(xch[15], xch[16], xch[17]));
    wave_sync();
}

/* ---- ordered accumulation ------------------------------------------------------------------ */
constexpr int kRowRec = 14;                 /* Jl[3] Ja[3] f pen gyro[3] flags 1.0f 0.0f */
constexpr int kRowFlagValid = 1, kRowFlagManifold = 2, kRowFlagPenOk = 4;
constexpr int kEntryRec = 3 * kRowRec;

/* accumulator owned by a lane: 0-5 ll lower, 6-14 la, 15-20 aa lower, 21-23 rhs.l, 24-26 rhs.a */
struct AccLane {
    int opA, opB, mode, gy; float acc;   /* mode 0: lhs, 1: lhs aa (+gyro/+0), 2: rhs, 3: idle */
    int opS, opG, onMask, gMask;         /* scale slot (pen | 1.0f), gyro slot (gyro_i | 0.0f), flag bits that enable the add / the gyro add */
};

__device__ __forceinline__ AccLane init_acc_lane(int lane, const BodyLocal& L)
{
    AccLane a; a.opA = 0; a.opB = 0; a.mode = 3; a.gy = -1; a.acc = 0.0f; a.opS = 7; a.opG = 13; a.onMask = kRowFlagPenOk; a.gMask = 0;
    const m33 massMat = diagm(mk3(L.mass, L.mass, L.mass));
    const m33 ll0 = zerom() + massMat * L.invDt2;                     /* solver.cpp:350-361 */
    const m33 aa0 = zerom() + L.Iw * L.invDt2;
    const f3 rl0 = massMat * ((L.pos - L.npos) * L.invDt2);           /* solver.cpp:363 */
    q4 qErr = L.rot * qconj(L.nrot);                                  /* solver.cpp:365-369 */
    f3 rotErr = mk3(qErr.x, qErr.y, qErr.z);
    rotErr = rotErr * 2.0f;
    if (qErr.w < 0.0f) rotErr = -rotErr;
    const f3 ra0 = L.Iw * (rotErr * L.invDt2);
    /* (row i, col j) of a lower triangle, lane order (0,0) (1,0) (2,0) (1,1) (2,1) (2,2) */
    const int k6 = lane < 6 ? lane : (lane >= 15 && lane < 21 ? lane - 15 : 0);
    const int ti = k6 < 3 ? k6 : (k6 < 5 ? k6 - 2 : 2);
    const int tj = k6 < 3 ? 0 : (k6 < 5 ? 1 : 2);
    if (lane < 6) {
        a.mode = 0; a.opA = ti; a.opB = tj;
        a.acc = comp3(tj == 0 ? ll0.c0 : (tj == 1 ? ll0.c1 : ll0.c2), ti);
    } else if (lane < 15) {
        const int k = lane - 6, i = k % 3, j = k / 3;                 /* la(row i, col j) = Jl_i * Ja_j */
        a.mode = 0; a.opA = i; a.opB = 3 + j; a.acc = 0.0f;
    } else if (lane < 21) {
        a.mode = 1; a.opA = 3 + ti; a.opB = 3 + tj; a.gy = (ti == tj) ? (8 + ti) : -1;
        a.opG = (ti == tj) ? (8 + ti) : 13; a.gMask = kRowFlagManifold;
        a.acc = comp3(tj == 0 ? aa0.c0 : (tj == 1 ? aa0.c1 : aa0.c2), ti);
    } else if (lane < 24) {
        const int i = lane - 21; a.mode = 2; a.opA = i; a.opB = 6; a.acc = comp3(rl0, i); a.opS = 12; a.onMask = kRowFlagValid;
    } else if (lane < 27) {
        const int i = lane - 24; a.mode = 2; a.opA = 3 + i; a.opB = 6; a.acc = comp3(ra0, i); a.opS = 12; a.onMask = kRowFlagValid;
    }
    return a;
}

__device__ __forceinline__ void accumulate_rows(AccLane& a, const float* rows, int nEntries)
{
    if (a.mode == 3) return;
    /* The additions must happen in row order, but the LDS reads of different rows are independent: fetch a block of
     * rows into registers first, then run the dependent chain of adds, branch-free.  Per lane (constant over rows):
     *   rhs lanes : acc += (J*f)*1.0f                         when the row is valid                (solver.cpp:383-384)
     *   lhs lanes : acc += (Ja*Jb)*pen                        when pen > 0 and finite               (solver.cpp:387-391)
     *   aa lanes  : ... then += gyro_i (diagonal) or += 0.0f  for manifold rows                      (solver.cpp:393-397)
     * The producer lane already encoded the row predicates in the flag word and provides the constants 1.0f / 0.0f
     * as record slots, so each row costs 5 LDS reads, 2 mul, 2 add, 2 tests and 2 selects. */
    constexpr int kBlock = 6;
    const int nRows = nEntries * 3;           /* the producer zeroes the flags of the two rows past the end */
    const float* pa = rows + a.opA;
    const float* pb = rows + a.opB;
    const float* ps = rows + a.opS;
    const float* pg = rows + a.opG;
    for (int r0 = 0; r0 < nRows; r0 += kBlock) {
        float va[kBlock], vb[kBlock], vs[kBlock], vg[kBlock]; int fl[kBlock];
#pragma unroll
        for (int k = 0; k < kBlock; ++k) {
            const int off = (r0 + k) * kRowRec;
            va[k] = pa[off]; vb[k] = pb[off]; vs[k] = ps[off]; vg[k] = pg[off]; fl[k] = __float_as_int(rows[off + 11]);
        }
#pragma unroll
        for (int k = 0; k < kBlock; ++k) {
            const float nv = a.acc + (va[k] * vb[k]) * vs[k];
            const float nvg = nv + vg[k];
            const float nv2 = (fl[k] & a.gMask) != 0 ? nvg : nv;
            a.acc = (fl[k] & a.onMask) != 0 ? nv2 : a.acc;
        }
    }
}

/* Same sums when every row of the task is a Manifold row with a usable penalty (the normal case; the caller checks it
 * wave-wide): no per-row predicates are needed, only the lane-constant "is this an aa entry" select. */
__device__ __forceinline__ void accumulate_rows_all_on(AccLane& a, const float* rows, int nEntries)
{
    if (a.mode == 3) return;
    constexpr int kBlock = 6;
    const int nRows = nEntries * 3;           /* multiple of 3; rows come in blocks of 6, the tail block is handled below */
    const bool isAA = a.mode == 1;
    const float* pa = rows + a.opA;
    const float* pb = rows + a.opB;
    const float* ps = rows + a.opS;
    const float* pg = rows + a.opG;
    int r0 = 0;
    for (; r0 + kBlock <= nRows; r0 += kBlock) {
        float va[kBlock], vb[kBlock], vs[kBlock], vg[kBlock];
#pragma unroll
        for (int k = 0; k < kBlock; ++k) {
            const int off = (r0 + k) * kRowRec;
            va[k] = pa[off]; vb[k] = pb[off]; vs[k] = ps[off]; vg[k] = pg[off];
        }
#pragma unroll
        for (int k = 0; k < kBlock; ++k) {
            const float nv = a.acc + (va[k] * vb[k]) * vs[k];
            const float nvg = nv + vg[k];
            a.acc = isAA ? nvg : nv;
        }
    }
    if (r0 < nRows) {                          /* three rows left */
        float va[3], vb[3], vs[3], vg[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int off = (r0 + k) * kRowRec;
            va[k] = pa[off]; vb[k] = pb[off]; vs[k] = ps[off]; vg[k] = pg[off];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float nv = a.acc + (va[k] * vb[k]) * vs[k];
            const float nvg = nv + vg[k];
            a.acc = isAA ? nvg : nv;
        }
    }
}

/* rows past the last entry of a block must read as invalid (the staging area keeps older tasks' records) */
__device__ __forceinline__ void clear_row_flags(float* rec)
{
    rec[11] = __int_as_float(0); rec[kRowRec + 11] = __int_as_float(0); rec[2 * kRowRec + 11] = __int_as_float(0);
}

__device__ __forceinline__ void put_row(float* r, f3 Jl, f3 Ja, float f, float pen, f3 gyro, int flags)
{
    r[0] = Jl.x; r[1] = Jl.y; r[2] = Jl.z; r[3] = Ja.x; r[4] = Ja.y; r[5] = Ja.z;
    r[6] = f; r[7] = pen; r[8] = gyro.x; r[9] = gyro.y; r[10] = gyro.z;
    const bool penOk = (pen > 0.0f) & (fabsf(pen) <= FLT_MAX);          /* pen > 0 && isfinite(pen), solver.cpp:387 */
    r[11] = __int_as_float(flags | (penOk ? kRowFlagPenOk : 0)); r[12] = 1.0f; r[13] = 0.0f;
}

/* ---- phase 1 for one Manifold contact held in registers: computeConstraint (mutating c) +
 * computeDerivatives + row records (solver.cpp:372-399 for rows 3c..3c+2) ------------------------- */
__device__ __forceinline__ void eval_contact_core(ContactRows& c, bool isA, const BodyLocal& L, f3 opos, q4 orot, float oinv, float fric,
                                                  float alpha, float* rec)
{
    ContactBasis db;
    if (isA) contact_compute_constraint(c, alpha, L.pos, L.rot, L.invMass, opos, orot, oinv, fric, db);
    else     contact_compute_constraint(c, alpha, opos, orot, oinv, L.pos, L.rot, L.invMass, fric, db);
    f3 Jl[3], Ja[3];
    contact_compute_derivatives(c, db, isA, L.rot, Jl, Ja);
    for (int k = 0; k < 3; ++k) {
        /* stiffness == FLT_MAX for manifold rows, motor == 0 (manifold.cpp:102-106): solver.cpp:379-381 */
        float desired = c.penalty[k] * c.C[k] + c.lambda[k] + 0.0f;
        float f = rclamp(desired, c.fmin[k], c.fmax[k]);
        f3 gyro = abs3(cross3(Ja[k], L.invIw * Ja[k])) * fabsf(f);
        put_row(rec + k * kRowRec, Jl[k], Ja[k], f, c.penalty[k], gyro, kRowFlagValid | kRowFlagManifold);
    }
}

/* The same with the lever arms already rotated: wOwn = rotate(q_own, r_own) (shared by the world point and the Jacobian,
 * manifold.cpp:27 and :265) and pOther = pos_other + rotate(q_other, r_other). */
__device__ __forceinline__ bool eval_contact_pts(ContactRows& c, bool isA, const BodyLocal& L, f3 wOwn, f3 pOther, float oinv, float fric,
                                                 float alpha, float* rec)
{
    const f3 pOwn = L.pos + wOwn;
    ContactBasis db;
    /* one inlined copy for both roles: lanes of a wave mix A-side and B-side contacts, a branch would run both */
    contact_compute_constraint_pts(c, alpha, isA ? pOwn : pOther, isA ? pOther : pOwn, isA ? L.invMass : oinv, isA ? oinv : L.invMass, fric, db);
    f3 Jl[3], Ja[3];
    contact_compute_derivatives_w(db, isA, wOwn, Jl, Ja);
    for (int k = 0; k < 3; ++k) {
        float desired = c.penalty[k] * c.C[k] + c.lambda[k] + 0.0f;        /* solver.cpp:379-381 with stiffness FLT_MAX, motor 0 */
        float f = rclamp(desired, c.fmin[k], c.fmax[k]);
        f3 gyro = abs3(cross3(Ja[k], L.invIw * Ja[k])) * fabsf(f);
        put_row(rec + k * kRowRec, Jl[k], Ja[k], f, c.penalty[k], gyro, kRowFlagValid | kRowFlagManifold);
    }
    /* all three penalties positive and finite (solver.cpp:387)? */
    return ((c.penalty[0] > 0.0f) & (fabsf(c.penalty[0]) <= FLT_MAX)) & ((c.penalty[1] > 0.0f) & (fabsf(c.penalty[1]) <= FLT_MAX)) &
           ((c.penalty[2] > 0.0f) & (fabsf(c.penalty[2]) <= FLT_MAX));
}

/* dual update from rotated lever arms (wA, wB) and world points (pA, pB) */
template <bool BITS = false>
__device__ __forceinline__ void dual_contact_pts(ContactRows& c, f3 pA, f3 wA, float imA, f3 pB, f3 wB, float imB, float fric, const SolverParams& P, float alpha)
{
    ContactBasis db;
    contact_compute_constraint_pts<BITS>(c, alpha, pA, pB, imA, imB, fric, db);
    bool needJ = false;
    bool active[3];
    for (int k = 0; k < 3; ++k) {
        float lu = rclamp(c.penalty[k] * c.C[k] + c.lambda[k], c.fmin[k], c.fmax[k]);
        active[k] = lu > c.fmin[k] && lu < c.fmax[k];
        c.lambda[k] = lu;
        needJ |= active[k];
    }
    if (needJ) {
        f3 JlA[3], JaA[3], JlB[3], JaB[3];
        contact_compute_derivatives_w(db, true, wA, JlA, JaA);
        contact_compute_derivatives_w(db, false, wB, JlB, JaB);
        for (int k = 0; k < 3; ++k) {
            if (!active[k]) continue;
            float lw = 0.0f, aw = 0.0f;                               /* rowPenaltyGain, solver.cpp:94-125 */
            lw += lensq3(JlA[k]); aw += lensq3(JaA[k]);
            lw += lensq3(JlB[k]); aw += lensq3(JaB[k]);
            float betaRow = row_penalty_gain(lw, aw, P.beta);
            c.penalty[k] = rmin(c.penalty[k] + betaRow * fabsf(c.C[k]), kManifoldPenaltyCap);
        }
    }
}

/* ---- phase 1: evaluate one incidence entry (<= 3 rows) for `body` ---------------------------- */
template <bool COH>
__device__ __forceinline__ void eval_entry(int ent, const BodyTable& B, const ManifoldTable& M, GForce* gf, const BodyLocal& L, float alpha,
                                           bool writeRows, float* rec, const HostRowsDev* hostRows = nullptr)
{
    const int kind = inc_kind(ent), idx = inc_index(ent);
    const bool isA = inc_isA(ent) != 0;
    if (kind == FK_MANIFOLD) {
        const int m = idx >> 2;
        const int2 pr = M.pair[m];
        const int other = isA ? pr.y : pr.x;
        f3 opos; q4 orot;
        load_pose<COH>(B, other, opos, orot);
        const float oinv = B.plane(BP_MASS)[other].y;
        ContactRows c;
        load_contact<COH>(M, idx, c);
        eval_contact_core(c, isA, L, opos, orot, oinv, M.fric[m], alpha, rec);
        store_contact_mutations<COH>(M, idx, c, writeRows);
    } else if (kind == FK_JOINT) {
        const int g = idx >> 1, part = idx & 1;
        GForce& f = gf[g];
        const bool hasA = f.bodyA >= 0;
        const int other = isA ? f.bodyB : f.bodyA;
        f3 opos = zero3(); q4 orot = identq();
        if (other >= 0) load_pose<COH>(B, other, opos, orot);
        JointRows j;
        j.rA = mk3(f.rA[0], f.rA[1], f.rA[2]); j.rB = mk3(f.rB[0], f.rB[1], f.rB[2]);
        j.q0 = mkq(f.q0[0], f.q0[1], f.q0[2], f.q0[3]);
        j.fixAngular = (f.pad & 1) != 0;
        if (isA) joint_compute_constraint(j, hasA, L.pos, L.rot, opos, orot);
        else     joint_compute_constraint(j, hasA, opos, orot, L.pos, L.rot);
        for (int k = 0; k < 3; ++k) {
            const int row = part * 3 + k;
            f3 Jl, Ja;
            joint_compute_derivatives(j, isA, false, L.rot, row, Jl, Ja);
            const float stiff = f.stiffness[row];
            const float pen = ldf<COH>(&f.penalty[row]);
            float lw = (stiff == FLT_MAX) ? ldf<COH>(&f.lambda[row]) : 0.0f;
            float desired = pen * j.C[row] + lw + f.motor[row];
            float fr = rclamp(desired, -FLT_MAX, FLT_MAX);
            put_row(rec + k * kRowRec, Jl, Ja, fr, pen, zero3(), kRowFlagValid);
            /* API visibility of Force::C after a post-stabilisation visit (the last computeConstraint of the step, joint.cpp:68-106);
             * both endpoints store, the later visitor's value stays — its stores are ordered behind the earlier visitor's publish */
            if (writeRows) stf<COH>(&f.C[row], j.C[row]);
        }
    } else if (gf[idx & 0xFFFFFF].kind == FK_HOST) {
        /* three rows of a user-defined Force: C / fmin / fmax / lambda and the Jacobians of THIS endpoint are what the subclass's
         * computeConstraint / computeDerivatives produced on the host for this very visit; the row force is solver.cpp:379-381 */
        const GForce& f = gf[idx & 0xFFFFFF];
        const HostRowsDev& h = hostRows[host_force_ordinal(f)];
        const int part = idx >> 24;
        for (int k = 0; k < 3; ++k) {
            const int row = part * 3 + k;
            if (row >= h.rows) { rec[k * kRowRec + 11] = __int_as_float(0); continue; }
            const float* jl = isA ? h.JlA[row] : h.JlB[row];
            const float* ja = isA ? h.JaA[row] : h.JaB[row];
            const float pen = h.penalty[row];
            const float lw = (h.stiffness[row] == FLT_MAX) ? h.lambda[row] : 0.0f;
            const float desired = pen * h.C[row] + lw + h.motor[row];
            const float fr = rclamp(desired, h.fmin[row], h.fmax[row]);
            put_row(rec + k * kRowRec, mk3(jl[0], jl[1], jl[2]), mk3(ja[0], ja[1], ja[2]), fr, pen, zero3(), kRowFlagValid);
        }
    } else { /* FK_SPRING */
        GForce& f = gf[idx];
        const bool hasA = f.bodyA >= 0;
        const int other = isA ? f.bodyB : f.bodyA;
        f3 opos = zero3(); q4 orot = identq();
        if (other >= 0) load_pose<COH>(B, other, opos, orot);
        SpringRows s;
        s.rA = mk3(f.rA[0], f.rA[1], f.rA[2]); s.rB = mk3(f.rB[0], f.rB[1], f.rB[2]); s.restLength = f.restLength;
        f3 pA = isA ? L.pos : opos, pB = isA ? opos : L.pos; q4 qA = isA ? L.rot : orot, qB = isA ? orot : L.rot;
        spring_compute_constraint(s, hasA, pA, qA, pB, qB);
        f3 Jl, Ja;
        spring_compute_derivatives(s, isA, hasA, pA, qA, pB, qB, Jl, Ja);
        const float pen = ldf<COH>(&f.penalty[0]);
        float lw = (f.stiffness[0] == FLT_MAX) ? ldf<COH>(&f.lambda[0]) : 0.0f;
        float desired = pen * s.C + lw + f.motor[0];
        float fr = rclamp(desired, f.fmin[0], f.fmax[0]);
        put_row(rec, Jl, Ja, fr, pen, zero3(), kRowFlagValid);
        if (writeRows) stf<COH>(&f.C[0], s.C);                          /* as for joints (spring.cpp:33-58) */
        rec[kRowRec + 11] = __int_as_float(0);
        rec[2 * kRowRec + 11] = __int_as_float(0);
    }
}

/* ---- phase 3: solve6x6 (solver.cpp:68-83) + pose update (solver.cpp:402-405) ---------------------------
 * Lane-parallel evaluation of exactly the reference's expression trees.  A wavefront issues one instruction per ~4 cycles
 * however few lanes are active, and an IEEE division is ~11 dependent instructions, so the INDEPENDENT divisions of a 3x3
 * solve (maths.h:104: L10 | L20, then L21, then z.x | z.y | z.z) are made side by side by the lanes of a quad, each on its own
 * operands — the same quotients of the same bits — and handed round with quad_perm DPP moves:
 *   quad j = lanes 4j..4j+3 (j = 0..3): sol_j = solve(ll, la.col j) for j < 3, solve(ll, b.l) for j = 3;
 *             then Schur column j = aa.col j - al * sol_j   |   rhs_s = b.a - al * x0
 *   quad 0 : y = solve(schur, rhs_s), x = x0 - AinvB * y, pose update
 * `accs` holds the 27 accumulated entries; `xch` is a 21-float LDS exchange area.  The result (p, q) is valid in lane 0 only
 * (the other lanes run the same instruction stream on operands nobody reads). */
__device__ __forceinline__ float quad_lane(float v, int c)            /* the value lane (lane & ~3) + c holds, c a constant */
{
    const int ctrl = c | (c << 2) | (c << 4) | (c << 6);              /* quad_perm:[c,c,c,c] */
    int r;
    switch (ctrl) {
    case 0x00: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00, 0xf, 0xf, true); break;
    case 0x55: r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55, 0xf, 0xf, true); break;
    default:   r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xaa, 0xf, 0xf, true); break;
    }
    return __int_as_float(r);
}

/* maths.h:104 with the independent divisions spread over the lanes of a quad (c = lane & 3); every lane of the quad returns x */
__device__ __forceinline__ f3 solve_ldlt3_quad(m33 A, f3 b, int c)
{
    const f3 L0 = A.c0;
    bool bad = fabsf(L0.x) < FLT_EPSILON;
    const float d0 = L0.x;
    const float q1 = (c == 0 ? L0.y : L0.z) / d0;                     /* lane 0: L10, lane 1: L20 */
    const float L10 = quad_lane(q1, 0), L20 = quad_lane(q1, 1);
    const f3 L1 = A.c1 - L0 * L10;
    bad |= fabsf(L1.y) < FLT_EPSILON;
    const float d1 = L1.y;
    const float L21 = L1.z / d1;
    const f3 L2 = A.c2 - L0 * L20 - L1 * L21;
    bad |= fabsf(L2.z) < FLT_EPSILON;
    const float d2 = L2.z;
    f3 y; y.x = b.x; y.y = b.y - L10 * y.x; y.z = b.z - L20 * y.x - L21 * y.y;
    const float zq = (c == 0 ? y.x : (c == 1 ? y.y : y.z)) / (c == 0 ? d0 : (c == 1 ? d1 : d2));
    f3 z; z.x = quad_lane(zq, 0); z.y = quad_lane(zq, 1); z.z = quad_lane(zq, 2);
    f3 x; x.z = z.z; x.y = z.y - L21 * x.z; x.x = z.x - L10 * x.y - L20 * x.z;
    return bad ? zero3() : x;
}

/* ---- the same solve in two stages (one body per wave, contacts only) ---------------------------------------------------------
 * lhsLin (ll) and the coupling block (la) are sums of penalty * J J^T: they are final as soon as the body's dual records are in,
 * usually long before the last pose it waits for.  Stage one runs in that wait: the LDL^T factors of ll, AinvB = ll^-1 la and
 * al * AinvB — everything solve6x6 (solver.cpp:68-83) derives from ll and la alone.  Stage two, behind the wait, substitutes the
 * right-hand side through the stored factors (same operations on the same bits as solve_ldlt3_quad) and solves the Schur system:
 * two of the eight division levels and the L1 / L2 rows leave the hop.
 * pre[0..8] AinvB (columns), pre[9..17] al * AinvB (columns), pre[18..20] L10 L20 L21, pre[21..23] d0 d1 d2, pre[24] bad. */
constexpr int kSolvePreFloats = 25;
__device__ __forceinline__ void solve_pose_pre(const float* accs, float* pre, int lane)
{
    m33 ll, la, al;
    ll.c0 = mk3(accs[0], accs[1], accs[2]); ll.c1 = mk3(accs[1], accs[3], accs[4]); ll.c2 = mk3(accs[2], accs[4], accs[5]);
    la.c0 = mk3(accs[6], accs[7], accs[8]); la.c1 = mk3(accs[9], accs[10], accs[11]); la.c2 = mk3(accs[12], accs[13], accs[14]);
    al = mtranspose(la);
    const int j = (lane >> 2) & 3, c = lane & 3;
    const int ro = j < 3 ? 6 + 3 * j : 6;
    const f3 b = mk3(accs[ro], accs[ro + 1], accs[ro + 2]);
    /* solve_ldlt3_quad, keeping the factors */
    const f3 L0 = ll.c0;
    bool bad = fabsf(L0.x) < FLT_EPSILON;
    const float d0 = L0.x;
    const float q1 = (c == 0 ? L0.y : L0.z) / d0;
    const float L10 = quad_lane(q1, 0), L20 = quad_lane(q1, 1);
    const f3 L1 = ll.c1 - L0 * L10;
    bad |= fabsf(L1.y) < FLT_EPSILON;
    const float d1 = L1.y;
    const float L21 = L1.z / d1;
    const f3 L2 = ll.c2 - L0 * L20 - L1 * L21;
    bad |= fabsf(L2.z) < FLT_EPSILON;
    const float d2 = L2.z;
    f3 y; y.x = b.x; y.y = b.y - L10 * y.x; y.z = b.z - L20 * y.x - L21 * y.y;
    const float zq = (c == 0 ? y.x : (c == 1 ? y.y : y.z)) / (c == 0 ? d0 : (c == 1 ? d1 : d2));
    f3 z; z.x = quad_lane(zq, 0); z.y = quad_lane(zq, 1); z.z = quad_lane(zq, 2);
    f3 x; x.z = z.z; x.y = z.y - L21 * x.z; x.x = z.x - L10 * x.y - L20 * x.z;
    const f3 sol = bad ? zero3() : x;
    const f3 prod = al * sol;
    if (lane < 12 && c == 0) {
        pre[j * 3 + 0] = sol.x; pre[j * 3 + 1] = sol.y; pre[j * 3 + 2] = sol.z;
        pre[9 + j * 3 + 0] = prod.x; pre[9 + j * 3 + 1] = prod.y; pre[9 + j * 3 + 2] = prod.z;
    }
    if (lane == 0) { pre[18] = L10; pre[19] = L20; pre[20] = L21; pre[21] = d0; pre[22] = d1; pre[23] = d2; pre[24] = bad ? 1.0f : 0.0f; }
    wave_sync();
}

__device__ __forceinline__ void solve_pose_post(const float* accs, float* xch, const float* pre, int lane, const BodyLocal& L, f3& p, q4& q)
{
    m33 la, al;
    la.c0 = mk3(accs[6], accs[7], accs[8]); la.c1 = mk3(accs[9], accs[10], accs[11]); la.c2 = mk3(accs[12], accs[13], accs[14]);
    al = mtranspose(la);
    const int j = (lane >> 2) & 3, c = lane & 3;
    const float L10 = pre[18], L20 = pre[19], L21 = pre[20], d0 = pre[21], d1 = pre[22], d2 = pre[23];
    const bool bad = pre[24] != 0.0f;
    /* x0 = solve(ll, b.l) through the stored factors: forward substitution, the three divisions side by side, back substitution */
    const f3 b = mk3(accs[21], accs[22], accs[23]);
    f3 y; y.x = b.x; y.y = b.y - L10 * y.x; y.z = b.z - L20 * y.x - L21 * y.y;
    const float zq = (c == 0 ? y.x : (c == 1 ? y.y : y.z)) / (c == 0 ? d0 : (c == 1 ? d1 : d2));
    f3 z; z.x = quad_lane(zq, 0); z.y = quad_lane(zq, 1); z.z = quad_lane(zq, 2);
    f3 x; x.z = z.z; x.y = z.y - L21 * x.z; x.x = z.x - L10 * x.y - L20 * x.z;
    const f3 x0 = bad ? zero3() : x;
    /* Schur column j = aa.col j - al * AinvB.col j   |   rhs_s = b.a - al * x0 */
    const f3 prodX = al * x0;
    const f3 prod = j < 3 ? mk3(pre[9 + j * 3], pre[9 + j * 3 + 1], pre[9 + j * 3 + 2]) : prodX;
    const int b0 = j == 0 ? 15 : (j == 1 ? 16 : (j == 2 ? 17 : 24)), b1 = j == 0 ? 16 : (j == 1 ? 18 : (j == 2 ? 19 : 25)),
              b2 = j == 0 ? 17 : (j == 1 ? 19 : (j == 2 ? 20 : 26));
    const f3 base = mk3(accs[b0], accs[b1], accs[b2]);
    const f3 out = base - prod;
    if (lane < 16 && c == 0) { xch[j * 3 + 0] = out.x; xch[j * 3 + 1] = out.y; xch[j * 3 + 2] = out.z; }
    wave_sync();
    {
        const m33 schur = mkm(mk3(xch[0], xch[1], xch[2]), mk3(xch[3], xch[4], xch[5]), mk3(xch[6], xch[7], xch[8]));
        const f3 rhs_s = mk3(xch[9], xch[10], xch[11]);
        const f3 ys = solve_ldlt3_quad(schur, rhs_s, c);
        const m33 AinvB = mkm(mk3(pre[0], pre[1], pre[2]), mk3(pre[3], pre[4], pre[5]), mk3(pre[6], pre[7], pre[8]));
        const f3 dl = x0 - AinvB * ys;
        p = L.pos - dl;                                               /* solver.cpp:403-405 */
        q4 dq = mkq(ys.x, ys.y, ys.z, 0.0f);
        q = qnormalize(L.rot - (dq * L.rot) * 0.5f);
    }
}

__device__ __forceinline__ void solve_pose(const float* accs, float* xch, int lane, const BodyLocal& L, f3& p, q4& q)
{
    m33 ll, la, al;
    /* lower triangles; the strictly-upper entries are never read by solve (maths.h:104) */
    ll.c0 = mk3(accs[0], accs[1], accs[2]); ll.c1 = mk3(accs[1], accs[3], accs[4]); ll.c2 = mk3(accs[2], accs[4], accs[5]);
    la.c0 = mk3(accs[6], accs[7], accs[8]); la.c1 = mk3(accs[9], accs[10], accs[11]); la.c2 = mk3(accs[12], accs[13], accs[14]);
    al = mtranspose(la); /* al accumulates (Ja_i*Jl_j)*pen = la(j,i) bit for bit (solver.cpp:389-390) */
    const int j = (lane >> 2) & 3, c = lane & 3;
    /* each quad fetches its own right-hand side / Schur base straight from the accumulator area */
    const int ro = j < 3 ? 6 + 3 * j : 21;
    const f3 rhs = mk3(accs[ro], accs[ro + 1], accs[ro + 2]);
    const f3 sol = solve_ldlt3_quad(ll, rhs, c);     /* col0, col1, col2, x0 */
    const f3 prod = al * sol;                        /* al * AinvB.col j  |  al * x0 */
    const int b0 = j == 0 ? 15 : (j == 1 ? 16 : (j == 2 ? 17 : 24)), b1 = j == 0 ? 16 : (j == 1 ? 18 : (j == 2 ? 19 : 25)),
              b2 = j == 0 ? 17 : (j == 1 ? 19 : (j == 2 ? 20 : 26));
    const f3 base = mk3(accs[b0], accs[b1], accs[b2]);
    const f3 out = base - prod;                      /* schur.col j  |  rhs_s */
    if (lane < 16 && c == 0) {
        xch[j * 3 + 0] = out.x; xch[j * 3 + 1] = out.y; xch[j * 3 + 2] = out.z;
        if (j > 0) { xch[12 + (j - 1) * 3 + 0] = sol.x; xch[12 + (j - 1) * 3 + 1] = sol.y; xch[12 + (j - 1) * 3 + 2] = sol.z; }
    }
    wave_sync();
    {
        /* every lane runs this; quad 0 (whose `sol` is column 0 of AinvB) holds the result */
        const m33 schur = mkm(mk3(xch[0], xch[1], xch[2]), mk3(xch[3], xch[4], xch[5]), mk3(xch[6], xch[7], xch[8]));
        const f3 rhs_s = mk3(xch[9], xch[10], xch[11]);
        const f3 y = solve_ldlt3_quad(schur, rhs_s, c);
        const m33 AinvB = mkm(sol, mk3(xch[12], xch[13], xch[14]), mk3(xch[15], xch[16], xch[17]));
        const f3 x0 = mk3(xch[18], xch[19], xch[20]);
        const f3 dl = x0 - AinvB * y;
        p = L.pos - dl;                                               /* solver.cpp:403-405 */
        q4 dq = mkq(y.x, y.y, y.z, 0.0f);
        q = qnormalize(L.rot - (dq * L.rot) * 0.5f);
    }
}

/* ---- dual update of one Manifold contact (solver.cpp:411-430 for rows 3c..3c+2) ------------------- */
__device__ __forceinline__ void dual_contact_core(ContactRows& c, f3 posA, q4 rotA, float imA, f3 posB, q4 rotB, float imB, float fric,
                                                  const SolverParams& P, float alpha)
{
    ContactBasis db;
    contact_compute_constraint(c, alpha, posA, rotA, imA, posB, rotB, imB, fric, db);
    bool needJ = false;
    bool active[3];
    for (int k = 0; k < 3; ++k) {
        float lu = rclamp(c.penalty[k] * c.C[k] + c.lambda[k], c.fmin[k], c.fmax[k]);
        active[k] = lu > c.fmin[k] && lu < c.fmax[k];
        c.lambda[k] = lu;
        needJ |= active[k];
    }
    if (needJ) {
        f3 JlA[3], JaA[3], JlB[3], JaB[3];
        contact_compute_derivatives(c, db, true, rotA, JlA, JaA);   /* both endpoints rebuild the same basis */
        contact_compute_derivatives(c, db, false, rotB, JlB, JaB);
        for (int k = 0; k < 3; ++k) {
            if (!active[k]) continue;
            float lw = 0.0f, aw = 0.0f;                               /* rowPenaltyGain, solver.cpp:94-125 */
            lw += lensq3(JlA[k]); aw += lensq3(JaA[k]);
            lw += lensq3(JlB[k]); aw += lensq3(JaB[k]);
            float betaRow = row_penalty_gain(lw, aw, P.beta);
            c.penalty[k] = rmin(c.penalty[k] + betaRow * fabsf(c.C[k]), kManifoldPenaltyCap);
        }
    }
}

template <bool COH>
__device__ __forceinline__ void store_contact_dual(const ManifoldTable& M, int s, const ContactRows& c, bool writeRows)
{
    st4<COH>(M.cN + s, make_float4(c.normal.x, c.normal.y, c.normal.z, c.C0t2));
    st4<COH>(M.cU1 + s, make_float4(c.u1.x, c.u1.y, c.u1.z, 0.f));
    st4<COH>(M.cU2 + s, make_float4(c.u2.x, c.u2.y, c.u2.z, 0.f));
    st4<COH>(M.cLam + s, make_float4(c.lambda[0], c.lambda[1], c.lambda[2], c.penetration));
    st4<COH>(M.cPen + s, make_float4(c.penalty[0], c.penalty[1], c.penalty[2], 0.f));
    sti<COH>(&M.cFS[s].y, c.stick ? 1 : 0);
    if (writeRows) { /* Force::C/fmin/fmax as the last computeConstraint of the step leaves them (API visibility only) */
        st4<COH>(M.cC + s, make_float4(c.C[0], c.C[1], c.C[2], 0.f));
        st4<COH>(M.cFmin + s, make_float4(c.fmin[0], c.fmin[1], c.fmin[2], 0.f));
        st4<COH>(M.cFmax + s, make_float4(c.fmax[0], c.fmax[1], c.fmax[2], 0.f));
    }
}

template <bool COH>
__device__ __forceinline__ void dual_contact(const BodyTable& B, const ManifoldTable& M, int s, const SolverParams& P, float alpha, bool writeRows)
{
    const int m = s >> 2;
    const int2 pr = M.pair[m];
    f3 posA, posB; q4 rotA, rotB;
    load_pose<COH>(B, pr.x, posA, rotA);
    load_pose<COH>(B, pr.y, posB, rotB);
    const float imA = B.plane(BP_MASS)[pr.x].y, imB = B.plane(BP_MASS)[pr.y].y;
    ContactRows c;
    load_contact<COH>(M, s, c);
    dual_contact_core(c, posA, rotA, imA, posB, rotB, imB, M.fric[m], P, alpha);
    store_contact_dual<COH>(M, s, c, writeRows);
}

/* ---- dual update of one Joint / Spring -------------------------------------------------------------- */
template <bool COH>
__device__ __forceinline__ void dual_gforce(const BodyTable& B, GForce& f, const SolverParams& P, bool writeRows)
{
    if (f.kind == FK_IGNORE || f.kind == FK_HOST) return;            /* (a host force's dual runs on the host: its computeConstraint is a virtual) */
    const bool hasA = f.bodyA >= 0;
    f3 posA = zero3(), posB; q4 rotA = identq(), rotB;
    if (hasA) load_pose<COH>(B, f.bodyA, posA, rotA);
    load_pose<COH>(B, f.bodyB, posB, rotB);
    if (f.kind == FK_JOINT) {
        JointRows j;
        j.rA = mk3(f.rA[0], f.rA[1], f.rA[2]); j.rB = mk3(f.rB[0], f.rB[1], f.rB[2]); j.q0 = mkq(f.q0[0], f.q0[1], f.q0[2], f.q0[3]);
        joint_compute_constraint(j, hasA, posA, rotA, posB, rotB);
        for (int r = 0; r < 6; ++r) {
            if (writeRows) { stf<COH>(&f.C[r], j.C[r]); stf<COH>(&f.fmin[r], -FLT_MAX); stf<COH>(&f.fmax[r], FLT_MAX); }
            if (f.stiffness[r] != FLT_MAX) continue;
            const float pen = ldf<COH>(&f.penalty[r]);
            float lu = rclamp(pen * j.C[r] + ldf<COH>(&f.lambda[r]), -FLT_MAX, FLT_MAX);
            bool active = lu > -FLT_MAX && lu < FLT_MAX;
            stf<COH>(&f.lambda[r], lu);
            if (active) {
                float lw = 0.0f, aw = 0.0f;
                f3 Jl, Ja;
                if (hasA) { joint_compute_derivatives(j, true, false, rotA, r, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja); }
                joint_compute_derivatives(j, false, false, rotB, r, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja);
                float betaRow = row_penalty_gain(lw, aw, P.beta);
                stf<COH>(&f.penalty[r], rmin(pen + betaRow * fabsf(j.C[r]), kPenaltyMax));
            }
        }
    } else { /* spring: rows with finite stiffness are skipped by the dual (solver.cpp:416-418) */
        SpringRows s;
        s.rA = mk3(f.rA[0], f.rA[1], f.rA[2]); s.rB = mk3(f.rB[0], f.rB[1], f.rB[2]); s.restLength = f.restLength;
        spring_compute_constraint(s, hasA, posA, rotA, posB, rotB);
        if (writeRows) stf<COH>(&f.C[0], s.C);
        if (f.stiffness[0] == FLT_MAX) {
            const float pen = ldf<COH>(&f.penalty[0]);
            float lu = rclamp(pen * s.C + ldf<COH>(&f.lambda[0]), f.fmin[0], f.fmax[0]);
            bool active = lu > f.fmin[0] && lu < f.fmax[0];
            stf<COH>(&f.lambda[0], lu);
            if (active) {
                float lw = 0.0f, aw = 0.0f;
                f3 Jl, Ja;
                if (hasA) { spring_compute_derivatives(s, true, hasA, posA, rotA, posB, rotB, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja); }
                spring_compute_derivatives(s, false, hasA, posA, rotA, posB, rotB, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja);
                float betaRow = row_penalty_gain(lw, aw, P.beta);
                stf<COH>(&f.penalty[0], rmin(pen + betaRow * fabsf(s.C), kPenaltyMax));
            }
        }
    }
}

/* ==========================================================================================
 * Level-launch kernels (one launch per Gauss-Seidel level, plain memory accesses)
 * ========================================================================================== */
constexpr int kWavesPerBlock = 4;

#ifndef AVBD_TU_SINGLE   /* (avbd_single.hip compiles only k_iterate_single from these headers: the plain kernels belong to the main translation unit) */
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_primal(BodyTable B, ManifoldTable M, GForce* gf, const int* __restrict__ order,
                                                                 const int* __restrict__ lvlStart, int level, const int* __restrict__ incStart,
                                                                 const int* __restrict__ inc, SolverParams P, float alpha, int writeRows, Counters* cnt, int it,
                                                                 const HostRowsDev* hostRows)
{
    __shared__ float sRows[kWavesPerBlock][66 * kEntryRec];
    __shared__ float sAcc[kWavesPerBlock][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int first = lvlStart[level], last = lvlStart[level + 1];
    const int slot = first + blockIdx.x * kWavesPerBlock + wv;
    if (slot >= last) return; /* wave-uniform */
    const int body = order[slot];
    float* rows = sRows[wv];
    float* accs = sAcc[wv];

    const BodyLocal L = load_body_local<false>(B, body, P);
    AccLane a = init_acc_lane(lane, L);
    const int eBeg = incStart[body], eEnd = incStart[body + 1];
    for (int base = eBeg; base < eEnd; base += 64) {
        const int nHere = min(64, eEnd - base);
        if (lane < nHere) eval_entry<false>(inc[base + lane], B, M, gf, L, alpha, writeRows != 0, rows + lane * kEntryRec, hostRows);
        else if (lane < nHere + 2) clear_row_flags(rows + lane * kEntryRec);
        wave_sync();
        accumulate_rows(a, rows, nHere);
        wave_sync();
    }
    if (lane < 27) accs[lane] = a.acc;
    wave_sync();
    f3 p; q4 q;
    solve_pose(accs, accs + 32, lane, L, p, q);
    if (lane == 0) {
        sanitize3(p, cnt, warn_key(1, it, body, 0), body, WK_POS);     /* solver.cpp:407-408 */
        sanitizeq(q, cnt, warn_key(1, it, body, 1), body, WK_ROT);
        float4 pr = B.plane(BP_POSR)[body];
        B.plane(BP_POSR)[body] = pack3(p, pr.w);
        B.plane(BP_QUAT)[body] = packq(q);
    }
}

__global__ void k_dual(BodyTable B, ManifoldTable M, const Counters* cnt, SolverParams P, float alpha)
{
    const int total = cnt->numManifolds * 4;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += gridDim.x * blockDim.x) {
        if ((s & 3) >= M.nc[s >> 2]) continue;
        dual_contact<false>(B, M, s, P, alpha, true);
    }
}

__global__ void k_dual_gforce(BodyTable B, GForce* gf, int ng, SolverParams P)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    dual_gforce<false>(B, gf[g], P, true);
}

#endif

} // namespace avbd
