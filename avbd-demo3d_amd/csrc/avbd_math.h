/*
 * avbd_math.h — device/host scalar math for the AVBD hot path on gfx950.
 *
 * Bit-exactness contract: every function evaluates the SAME IEEE-754 binary32 expression tree
 * as the corresponding inline in the reference's maths.h (cited per function, paths relative to
 * /root/reference/source/).  Nothing here may be re-associated, fused (build with
 * -ffp-contract=off, also enforced by the pragma below) or replaced by an approximate
 * instruction (build with -fhip-fp32-correctly-rounded-divide-sqrt; no v_rcp/v_rsq shortcuts).
 * The reference's min/max are compare-selects with its NaN behaviour, not fminf/fmaxf.
 */
#pragma once

#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AVBD_HD __host__ __device__ __forceinline__
#else
#define AVBD_HD inline
#endif

#pragma clang fp contract(off)

namespace avbd {

#if !defined(__HIPCC__)
using std::isfinite;
#endif

struct f3 { float x, y, z; };
struct q4 { float x, y, z, w; };
struct m33 { f3 c0, c1, c2; }; /* column-major, maths.h:74-76 */

constexpr float kVecEps = 1e-6f; /* maths.h:18 */

AVBD_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
AVBD_HD f3 zero3() { return mk3(0.0f, 0.0f, 0.0f); }
AVBD_HD q4 mkq(float x, float y, float z, float w) { q4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
AVBD_HD q4 identq() { return mkq(0.0f, 0.0f, 0.0f, 1.0f); }

AVBD_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }           /* maths.h:32 */
AVBD_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }           /* maths.h:33 */
AVBD_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }                                 /* maths.h:31 */
AVBD_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }               /* maths.h:34 */
AVBD_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }               /* maths.h:35 */
AVBD_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                 /* maths.h:42 */
AVBD_HD float lensq3(f3 v) { return dot3(v, v); }                                            /* maths.h:43 */
AVBD_HD float len3(f3 v) { return sqrtf(lensq3(v)); }                                        /* maths.h:44 */
AVBD_HD f3 cross3(f3 a, f3 b)                                                                /* maths.h:46 */
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
AVBD_HD f3 abs3(f3 v) { return mk3(fabsf(v.x), fabsf(v.y), fabsf(v.z)); }                    /* maths.h:47 */
AVBD_HD float comp3(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

AVBD_HD q4 operator+(q4 a, q4 b) { return mkq(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }  /* maths.h:59 */
AVBD_HD q4 operator-(q4 a, q4 b) { return mkq(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }  /* maths.h:60 */
AVBD_HD q4 operator*(q4 q, float s) { return mkq(q.x * s, q.y * s, q.z * s, q.w * s); }       /* maths.h:64 */
AVBD_HD q4 qconj(q4 q) { return mkq(-q.x, -q.y, -q.z, q.w); }                                 /* maths.h:66 */
AVBD_HD q4 qnormalize(q4 q)                                                                   /* maths.h:65 */
{
    float magSq = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    if (magSq < kVecEps) return identq();
    return q * (1.0f / sqrtf(magSq));
}
AVBD_HD q4 operator*(q4 a, q4 b)                                                              /* maths.h:67 */
{
    return mkq(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
               a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
               a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
               a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
AVBD_HD f3 qrotate(q4 q, f3 v)                                                                /* maths.h:68 */
{
    f3 u = mk3(q.x, q.y, q.z);
    f3 t = cross3(u, v) * 2.0f;
    return v + t * q.w + cross3(u, t);
}

AVBD_HD m33 mkm(f3 a, f3 b, f3 c) { m33 r; r.c0 = a; r.c1 = b; r.c2 = c; return r; }
AVBD_HD m33 zerom() { return mkm(zero3(), zero3(), zero3()); }
AVBD_HD m33 diagm(f3 d) { return mkm(mk3(d.x, 0.0f, 0.0f), mk3(0.0f, d.y, 0.0f), mk3(0.0f, 0.0f, d.z)); } /* maths.h:79 */
AVBD_HD m33 mtranspose(m33 m)                                                                 /* maths.h:82 */
{
    return mkm(mk3(m.c0.x, m.c1.x, m.c2.x), mk3(m.c0.y, m.c1.y, m.c2.y), mk3(m.c0.z, m.c1.z, m.c2.z));
}
AVBD_HD f3 operator*(m33 m, f3 v) { return m.c0 * v.x + m.c1 * v.y + m.c2 * v.z; }            /* maths.h:83 */
AVBD_HD m33 operator*(m33 a, m33 b) { return mkm(a * b.c0, a * b.c1, a * b.c2); }             /* maths.h:84 */
AVBD_HD m33 operator*(m33 m, float s) { return mkm(m.c0 * s, m.c1 * s, m.c2 * s); }           /* maths.h:86 */
AVBD_HD m33 operator+(m33 a, m33 b) { return mkm(a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2); }    /* maths.h:85 */
AVBD_HD m33 operator-(m33 a, m33 b) { return mkm(a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2); }    /* maths.h:92 */
AVBD_HD m33 quat_to_mat(q4 q)                                                                 /* maths.h:88 */
{
    float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
    float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    return mkm(mk3(1.0f - 2.0f * (yy + zz), 2.0f * (xy + wz), 2.0f * (xz - wy)),
               mk3(2.0f * (xy - wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz + wx)),
               mk3(2.0f * (xz + wy), 2.0f * (yz - wx), 1.0f - 2.0f * (xx + yy)));
}

/* maths.h:101-103 — compare-selects, NOT fminf/fmaxf (NaN and signed-zero behaviour differ). */
AVBD_HD float rmin(float a, float b) { return a < b ? a : b; }
AVBD_HD float rmax(float a, float b) { return a > b ? a : b; }
AVBD_HD float rclamp(float x, float lo, float hi) { return rmax(lo, rmin(hi, x)); }

/* maths.h:104 — 3x3 LDL^T solve on the lower triangle; zero vector when a pivot is < FLT_EPSILON. */
AVBD_HD f3 solve_ldlt3(m33 A, f3 b)
{
    f3 L0 = A.c0;
    if (fabsf(L0.x) < FLT_EPSILON) return zero3();
    float d0 = L0.x;
    float L10 = L0.y / d0;
    float L20 = L0.z / d0;
    f3 L1 = A.c1 - L0 * L10;
    if (fabsf(L1.y) < FLT_EPSILON) return zero3();
    float d1 = L1.y;
    float L21 = L1.z / d1;
    f3 L2 = A.c2 - L0 * L20 - L1 * L21;
    if (fabsf(L2.z) < FLT_EPSILON) return zero3();
    float d2 = L2.z;
    f3 y; y.x = b.x; y.y = b.y - L10 * y.x; y.z = b.z - L20 * y.x - L21 * y.y;
    f3 z; z.x = y.x / d0; z.y = y.y / d1; z.z = y.z / d2;
    f3 x; x.z = z.z; x.y = z.y - L21 * x.z; x.x = z.x - L10 * x.y - L20 * x.z;
    return x;
}

/* The same solve without early returns: every operation of the full path is executed (a tiny pivot just produces inf/nan
 * that the final select discards), so a wavefront runs one straight instruction stream.  Same results bit for bit. */
AVBD_HD f3 solve_ldlt3_branchless(m33 A, f3 b)
{
    f3 L0 = A.c0;
    bool bad = fabsf(L0.x) < FLT_EPSILON;
    float d0 = L0.x;
    float L10 = L0.y / d0;
    float L20 = L0.z / d0;
    f3 L1 = A.c1 - L0 * L10;
    bad |= fabsf(L1.y) < FLT_EPSILON;
    float d1 = L1.y;
    float L21 = L1.z / d1;
    f3 L2 = A.c2 - L0 * L20 - L1 * L21;
    bad |= fabsf(L2.z) < FLT_EPSILON;
    float d2 = L2.z;
    f3 y; y.x = b.x; y.y = b.y - L10 * y.x; y.z = b.z - L20 * y.x - L21 * y.y;
    f3 z; z.x = y.x / d0; z.y = y.y / d1; z.z = y.z / d2;
    f3 x; x.z = z.z; x.y = z.y - L21 * x.z; x.x = z.x - L10 * x.y - L20 * x.z;
    return bad ? zero3() : x;
}

/* solver.cpp:68-83 — Schur-complement 6x6 solve built from five 3x3 LDL^T solves. */
AVBD_HD void solve_block6(m33 ll, m33 la, m33 al, m33 aa, f3 bl, f3 ba, f3& xl, f3& xa)
{
    f3 col0 = solve_ldlt3(ll, la.c0);
    f3 col1 = solve_ldlt3(ll, la.c1);
    f3 col2 = solve_ldlt3(ll, la.c2);
    m33 AinvB = mkm(col0, col1, col2);
    f3 x0 = solve_ldlt3(ll, bl);
    m33 schur = aa - al * AinvB;
    f3 rhs_s = ba - al * x0;
    f3 y = solve_ldlt3(schur, rhs_s);
    xl = x0 - AinvB * y;
    xa = y;
}

AVBD_HD bool finite3(f3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }
AVBD_HD bool finiteq(q4 q) { return isfinite(q.x) && isfinite(q.y) && isfinite(q.z) && isfinite(q.w); }

} // namespace avbd
