/*
 * avbd_oracle.cpp — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 * See avbd_oracle.h for the role of this file.  Citations are into /root/reference/source/.
 *
 * Storage differs from the reference (flat vectors + index lists instead of intrusive
 * pointer lists) but every traversal order and every floating-point expression tree is the
 * reference's: lists are kept in creation order and walked backwards, which is exactly the
 * head-insertion order of force.cpp:16-29 / rigid.cpp:20-21.
 */
#include "avbd_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace orc {

/* ------------------------------------------------------------------ maths.h ------------ */
static const float VEC_EPS = 1e-6f;                                  /* maths.h:18 */

static inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
static inline V3 vzero() { return v3(0.0f, 0.0f, 0.0f); }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   /* :32 */
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :33 */
static inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }                         /* :31 */
static inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }       /* :34 */
static inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }       /* :35 */
static inline float comp(const V3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         /* :42 */
static inline float lengthSq(V3 v) { return dot(v, v); }
static inline float length(V3 v) { return sqrtf(lengthSq(v)); }
static inline V3 cross(V3 a, V3 b)                                                        /* :46 */
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline V3 vabs(V3 v) { return v3(fabsf(v.x), fabsf(v.y), fabsf(v.z)); }            /* :47 */

static inline Q4 q4(float x, float y, float z, float w) { Q4 r = {x, y, z, w}; return r; }
static inline Q4 qident() { return q4(0.0f, 0.0f, 0.0f, 1.0f); }
static inline Q4 operator+(Q4 a, Q4 b) { return q4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline Q4 operator-(Q4 a, Q4 b) { return q4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline Q4 operator*(Q4 q, float s) { return q4(q.x * s, q.y * s, q.z * s, q.w * s); }   /* :64 */
static inline Q4 qnormalize(Q4 q)                                                              /* :65 */
{
    float magSq = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    if (magSq < VEC_EPS) return qident();
    return q * (1.0f / sqrtf(magSq));
}
static inline Q4 conj(Q4 q) { return q4(-q.x, -q.y, -q.z, q.w); }                              /* :66 */
static inline Q4 operator*(Q4 a, Q4 b)                                                         /* :67 */
{
    return q4(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
              a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
static inline V3 rotate(Q4 q, V3 v)                                                            /* :68 */
{
    V3 u = v3(q.x, q.y, q.z);
    V3 t = cross(u, v) * 2.0f;
    return v + t * q.w + cross(u, t);
}

static inline M3 m3(V3 a, V3 b, V3 c) { M3 r; r.c[0] = a; r.c[1] = b; r.c[2] = c; return r; }
static inline M3 mzero() { return m3(vzero(), vzero(), vzero()); }
static inline M3 mdiag(V3 d) { return m3(v3(d.x, 0, 0), v3(0, d.y, 0), v3(0, 0, d.z)); }       /* :79 */
static inline M3 transpose(const M3& m)                                                        /* :82 */
{
    return m3(v3(m.c[0].x, m.c[1].x, m.c[2].x), v3(m.c[0].y, m.c[1].y, m.c[2].y), v3(m.c[0].z, m.c[1].z, m.c[2].z));
}
static inline V3 operator*(const M3& m, V3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; } /* :83 */
static inline M3 operator*(const M3& a, const M3& b) { return m3(a * b.c[0], a * b.c[1], a * b.c[2]); } /* :84 */
static inline M3 operator*(const M3& m, float s) { return m3(m.c[0] * s, m.c[1] * s, m.c[2] * s); } /* :86 */
static inline M3 operator-(const M3& a, const M3& b) { return m3(a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2]); } /* :92 */
static inline void addTo(M3& a, const M3& b) { a.c[0] = a.c[0] + b.c[0]; a.c[1] = a.c[1] + b.c[1]; a.c[2] = a.c[2] + b.c[2]; } /* :89 */
static inline M3 matFromQuat(Q4 q)                                                             /* :88 */
{
    float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
    float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    return m3(v3(1 - 2 * (yy + zz), 2 * (xy + wz), 2 * (xz - wy)),
              v3(2 * (xy - wz), 1 - 2 * (xx + zz), 2 * (yz + wx)),
              v3(2 * (xz + wy), 2 * (yz - wx), 1 - 2 * (xx + yy)));
}
static inline float fminr(float a, float b) { return a < b ? a : b; }                          /* :101 */
static inline float fmaxr(float a, float b) { return a > b ? a : b; }                          /* :102 */
static inline float clampr(float x, float a, float b) { return fmaxr(a, fminr(b, x)); }        /* :103 */

V3 solve3(const M3& A, V3 b)                                                                   /* :104 */
{
    V3 L0 = A.c[0];
    if (fabsf(L0.x) < FLT_EPSILON) return vzero();
    float d0 = L0.x;
    float L10 = L0.y / d0;
    float L20 = L0.z / d0;
    V3 L1 = A.c[1] - L0 * L10;
    if (fabsf(L1.y) < FLT_EPSILON) return vzero();
    float d1 = L1.y;
    float L21 = L1.z / d1;
    V3 L2 = A.c[2] - L0 * L20 - L1 * L21;
    if (fabsf(L2.z) < FLT_EPSILON) return vzero();
    float d2 = L2.z;
    V3 y; y.x = b.x; y.y = b.y - L10 * y.x; y.z = b.z - L20 * y.x - L21 * y.y;
    V3 z; z.x = y.x / d0; z.y = y.y / d1; z.z = y.z / d2;
    V3 x; x.z = z.z; x.y = z.y - L21 * x.z; x.x = z.x - L10 * x.y - L20 * x.z;
    return x;
}

/* ------------------------------------------------------------------ solver.cpp helpers -- */
static const float MANIFOLD_PENALTY_CAP = 2000000.0f;                 /* solver.cpp:29, manifold.cpp:23 */

static inline M3 outerSolver(V3 a, V3 b) { return m3(a * b.x, a * b.y, a * b.z); } /* solver.cpp:36-39 */

static bool finite3(V3 v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); }
static bool finite4(Q4 q) { return std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z) && std::isfinite(q.w); }

static void sanitizeVec3(V3& v, const char* label, int bodyId)        /* solver.cpp:51-58 */
{
    if (!finite3(v)) {
        std::printf("[Physics] Warning: body %d produced non-finite %s (%.3f, %.3f, %.3f); resetting to zero.\n",
                    bodyId, label, v.x, v.y, v.z);
        v = vzero();
    }
}
static void sanitizeQuat(Q4& q, const char* label, int bodyId)        /* solver.cpp:60-66 */
{
    if (!finite4(q)) {
        std::printf("[Physics] Warning: body %d produced non-finite %s; resetting to identity.\n", bodyId, label);
        q = qident();
    }
}

void solve6(const M3& ll, const M3& la, const M3& al, const M3& aa, V3 bl, V3 ba, V3& xl, V3& xa) /* solver.cpp:68-83 */
{
    V3 col0 = solve3(ll, la.c[0]);
    V3 col1 = solve3(ll, la.c[1]);
    V3 col2 = solve3(ll, la.c[2]);
    M3 AinvB = m3(col0, col1, col2);
    V3 x0 = solve3(ll, bl);
    M3 schur = aa - al * AinvB;
    V3 rhs_s = ba - al * x0;
    V3 y = solve3(schur, rhs_s);
    V3 x = x0 - AinvB * y;
    xl = x; xa = y;
}

static void clampAngularVelocity(V3& w)                               /* solver.cpp:85-92 */
{
    const float maxSpeed = 80.0f;
    float len = length(w);
    if (len > maxSpeed && len > VEC_EPS) {
        float s = maxSpeed / len;
        w = w * s;
    }
}

/* ------------------------------------------------------------------ rigid.cpp ---------- */
void initBoxBody(Body& b, V3 size, float density, float friction, V3 pos, Q4 rot, V3 lin, V3 ang) /* rigid.cpp:12-41 */
{
    b.pos = pos; b.rot = rot; b.linVel = lin; b.angVel = ang; b.prevLinVel = lin; b.prevAngVel = ang;
    b.initPos = vzero(); b.initRot = qident(); b.inerPos = vzero(); b.inerRot = qident();
    b.size = size; b.friction = friction;
    b.mass = size.x * size.y * size.z * density;
    b.invMass = (b.mass > 0.0f) ? 1.0f / b.mass : 0.0f;
    b.radius = length(size) * 0.5f;
    if (b.invMass > 0.0f) {
        float Ixx = (1.0f / 12.0f) * b.mass * (size.y * size.y + size.z * size.z);
        float Iyy = (1.0f / 12.0f) * b.mass * (size.x * size.x + size.z * size.z);
        float Izz = (1.0f / 12.0f) * b.mass * (size.x * size.x + size.y * size.y);
        b.I = m3(v3(Ixx, 0, 0), v3(0, Iyy, 0), v3(0, 0, Izz));
        b.invI = m3(v3(1.0f / Ixx, 0, 0), v3(0, 1.0f / Iyy, 0), v3(0, 0, 1.0f / Izz));
    } else {
        b.I = mzero();
        b.invI = mzero();
    }
}

M3 worldInvInertia(const Body& b) { M3 R = matFromQuat(b.rot); return R * b.invI * transpose(R); } /* rigid.cpp:51-54 */
M3 worldInertia(const Body& b) { M3 R = matFromQuat(b.rot); return R * b.I * transpose(R); }       /* rigid.cpp:56-59 */

/* ------------------------------------------------------------------ collision.cpp ------ */
namespace sat {

const int   MAX_CONTACTS = 4;                                        /* collision.cpp:18 */
const int   MAX_POLY = 16;                                           /* :19 */
const float AXIS_EPS = 1.0e-6f;                                      /* :20 */
const float PLANE_EPS = 1.0e-5f;                                     /* :21 */
const float MERGE_DIST_SQ = 1.0e-6f;                                 /* :22 */
const float PERSIST_MARGIN = kCollisionMargin;                       /* :23 */

struct Box { V3 center, half; V3 axis[3]; };                         /* :31-35 */
struct Best { int type, iA, iB; float sep; V3 nAB; bool valid; };    /* :37-44 */
struct Frame { int axisIndex; V3 normal, center, u, v; float eU, eV; }; /* :46-54 */

static Box makeBox(const Body& b)                                    /* :56-66 */
{
    Box o; o.center = b.pos; o.half = b.size * 0.5f;
    M3 R = matFromQuat(b.rot);
    o.axis[0] = R.c[0]; o.axis[1] = R.c[1]; o.axis[2] = R.c[2];
    return o;
}
static float absDot(V3 a, V3 b) { return fabsf(dot(a, b)); }         /* :68-71 */

static void faceAxes(const Box& b, int ai, V3& u, V3& v, float& eU, float& eV) /* :73-91 */
{
    if (ai == 0) { u = b.axis[1]; v = b.axis[2]; eU = b.half.y; eV = b.half.z; }
    else if (ai == 1) { u = b.axis[0]; v = b.axis[2]; eU = b.half.x; eV = b.half.z; }
    else { u = b.axis[0]; v = b.axis[1]; eU = b.half.x; eV = b.half.y; }
}

static void faceFrame(const Box& b, int ai, V3 outward, Frame& f)    /* :93-100 */
{
    float sign = dot(outward, b.axis[ai]) >= 0.0f ? 1.0f : -1.0f;
    f.axisIndex = ai;
    f.normal = b.axis[ai] * sign;
    f.center = b.center + f.normal * comp(b.half, ai);
    faceAxes(b, ai, f.u, f.v, f.eU, f.eV);
}

static int incidentAxis(const Box& b, V3 refNormal)                  /* :102-116 */
{
    int axis = 0; float best = -FLT_MAX;
    for (int i = 0; i < 3; ++i) {
        float d = absDot(b.axis[i], refNormal);
        if (d > best) { best = d; axis = i; }
    }
    return axis;
}

static void incidentFace(const Box& b, int ai, V3 refNormal, V3 out[4]) /* :118-134 */
{
    float sign = dot(b.axis[ai], refNormal) > 0.0f ? -1.0f : 1.0f;
    V3 fn = b.axis[ai] * sign;
    V3 fc = b.center + fn * comp(b.half, ai);
    V3 u, v; float eU, eV;
    faceAxes(b, ai, u, v, eU, eV);
    out[0] = fc + u * eU + v * eV;
    out[1] = fc - u * eU + v * eV;
    out[2] = fc - u * eU - v * eV;
    out[3] = fc + u * eU - v * eV;
}

static int clipPoly(const V3* in, int n, V3 pn, float po, V3* out)   /* :136-174 */
{
    if (n <= 0) return 0;
    int m = 0;
    V3 a = in[n - 1];
    float da = dot(pn, a) - po;
    for (int i = 0; i < n; ++i) {
        V3 b = in[i];
        float db = dot(pn, b) - po;
        bool aIn = da <= PLANE_EPS;
        bool bIn = db <= PLANE_EPS;
        if (aIn != bIn) {
            float t = 0.0f;
            float denom = da - db;
            if (fabsf(denom) > AXIS_EPS) t = clampr(da / denom, 0.0f, 1.0f);
            if (m < MAX_POLY) out[m++] = a + (b - a) * t;
        }
        if (bIn && m < MAX_POLY) out[m++] = b;
        a = b; da = db;
    }
    return m;
}

static bool addContact(const Body& A, const Body& B, Contact* cs, int& count, V3* mids, V3 xA, V3 xB, int key, V3 nBA) /* :176-206 */
{
    V3 mid = (xA + xB) * 0.5f;
    for (int i = 0; i < count; ++i) {
        V3 d = mid - mids[i];
        if (lengthSq(d) < MERGE_DIST_SQ) return false;
    }
    if (count >= MAX_CONTACTS) return false;
    Contact& c = cs[count];
    c.feature = key;
    c.rA = rotate(conj(A.rot), xA - A.pos);
    c.rB = rotate(conj(B.rot), xB - B.pos);
    c.normal = nBA;
    c.penetration = fmaxr(0.0f, -dot(xA - xB, nBA));
    c.C0_n = 0.0f; c.C0_t = vzero(); c.stick = false;
    mids[count] = mid;
    ++count;
    return true;
}

static bool testAxis(const Box& a, const Box& b, V3 delta, V3 axis, int type, int iA, int iB, Best& best) /* :208-247 */
{
    float lenSq = lengthSq(axis);
    if (lenSq < AXIS_EPS) return true;
    V3 n = axis / sqrtf(lenSq);
    if (dot(n, delta) < 0.0f) n = -n;
    float distance = fabsf(dot(delta, n));
    float rA = a.half.x * absDot(n, a.axis[0]) + a.half.y * absDot(n, a.axis[1]) + a.half.z * absDot(n, a.axis[2]);
    float rB = b.half.x * absDot(n, b.axis[0]) + b.half.y * absDot(n, b.axis[1]) + b.half.z * absDot(n, b.axis[2]);
    float separation = distance - (rA + rB);
    if (separation > PERSIST_MARGIN) return false;
    if (!best.valid || separation > best.sep) {
        best.valid = true; best.type = type; best.iA = iA; best.iB = iB; best.sep = separation; best.nAB = n;
    }
    return true;
}

static void supportEdge(const Box& b, int ai, V3 dir, V3& e0, V3& e1) /* :249-263 */
{
    int a1 = (ai + 1) % 3, a2 = (ai + 2) % 3;
    float s1 = dot(dir, b.axis[a1]) >= 0.0f ? 1.0f : -1.0f;
    float s2 = dot(dir, b.axis[a2]) >= 0.0f ? 1.0f : -1.0f;
    V3 ec = b.center + b.axis[a1] * (comp(b.half, a1) * s1) + b.axis[a2] * (comp(b.half, a2) * s2);
    e0 = ec - b.axis[ai] * comp(b.half, ai);
    e1 = ec + b.axis[ai] * comp(b.half, ai);
}

static void closestSegSeg(V3 p0, V3 p1, V3 q0, V3 q1, V3& c0, V3& c1) /* :265-311 */
{
    V3 d1 = p1 - p0, d2 = q1 - q0, r = p0 - q0;
    float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    float s = 0.0f, t = 0.0f;
    if (a <= AXIS_EPS && e <= AXIS_EPS) { c0 = p0; c1 = q0; return; }
    if (a <= AXIS_EPS) {
        t = clampr(f / e, 0.0f, 1.0f);
    } else {
        float c = dot(d1, r);
        if (e <= AXIS_EPS) {
            s = clampr(-c / a, 0.0f, 1.0f);
        } else {
            float b = dot(d1, d2);
            float denom = a * e - b * b;
            if (fabsf(denom) > AXIS_EPS) s = clampr((b * f - c * e) / denom, 0.0f, 1.0f);
            t = (b * s + f) / e;
            if (t < 0.0f) { t = 0.0f; s = clampr(-c / a, 0.0f, 1.0f); }
            else if (t > 1.0f) { t = 1.0f; s = clampr((b - c) / a, 0.0f, 1.0f); }
        }
    }
    c0 = p0 + d1 * s;
    c1 = q0 + d2 * t;
}

static int faceManifold(const Body& A, const Body& B, const Box& boxA, const Box& boxB, bool refIsA, int refAxis, V3 nAB, Contact* cs) /* :313-394 */
{
    const Box& refBox = refIsA ? boxA : boxB;
    const Box& incBox = refIsA ? boxB : boxA;
    V3 outward = refIsA ? nAB : -nAB;
    V3 nBA = -nAB;
    Frame rf; faceFrame(refBox, refAxis, outward, rf);
    int incAxis = incidentAxis(incBox, rf.normal);
    V3 clip0[MAX_POLY], clip1[MAX_POLY];
    incidentFace(incBox, incAxis, rf.normal, clip0);
    int count = 4;
    V3 n0 = rf.u;    float o0 = dot(n0, rf.center) + rf.eU;
    count = clipPoly(clip0, count, n0, o0, clip1); if (!count) return 0;
    V3 n1 = -rf.u;   float o1 = dot(n1, rf.center) + rf.eU;
    count = clipPoly(clip1, count, n1, o1, clip0); if (!count) return 0;
    V3 n2 = rf.v;    float o2 = dot(n2, rf.center) + rf.eV;
    count = clipPoly(clip0, count, n2, o2, clip1); if (!count) return 0;
    V3 n3 = -rf.v;   float o3 = dot(n3, rf.center) + rf.eV;
    count = clipPoly(clip1, count, n3, o3, clip0); if (!count) return 0;

    int cc = 0; V3 mids[MAX_CONTACTS];
    int prefix = (refIsA ? 0 : 1) << 24;
    prefix |= (refAxis & 0xFF) << 16;
    prefix |= (incAxis & 0xFF) << 8;
    for (int i = 0; i < count && cc < MAX_CONTACTS; ++i) {
        V3 pInc = clip0[i];
        float distance = dot(pInc - rf.center, rf.normal);
        if (distance > PERSIST_MARGIN) continue;
        V3 pRef = pInc - rf.normal * distance;
        V3 xA = refIsA ? pRef : pInc;
        V3 xB = refIsA ? pInc : pRef;
        V3 rel = pRef - rf.center;
        float uC = dot(rel, rf.u), vC = dot(rel, rf.v);
        float uN = (rf.eU > AXIS_EPS) ? (uC / rf.eU) : 0.0f;
        float vN = (rf.eV > AXIS_EPS) ? (vC / rf.eV) : 0.0f;
        int qU = (int)floorf(clampr((uN + 1.0f) * 7.5f, 0.0f, 15.0f));
        int qV = (int)floorf(clampr((vN + 1.0f) * 7.5f, 0.0f, 15.0f));
        int key = prefix | ((qU & 0x0F) << 4) | (qV & 0x0F);
        addContact(A, B, cs, cc, mids, xA, xB, key, nBA);
    }
    return cc;
}

static int edgeContact(const Body& A, const Body& B, const Box& boxA, const Box& boxB, int axA, int axB, V3 nAB, Contact* cs) /* :396-416 */
{
    V3 a0, a1, b0, b1;
    supportEdge(boxA, axA, nAB, a0, a1);
    supportEdge(boxB, axB, -nAB, b0, b1);
    V3 xA, xB;
    closestSegSeg(a0, a1, b0, b1, xA, xB);
    int cc = 0; V3 mids[MAX_CONTACTS];
    int key = (2 << 24) | ((axA & 0xFF) << 8) | (axB & 0xFF);
    addContact(A, B, cs, cc, mids, xA, xB, key, -nAB);
    return cc;
}

} // namespace sat

int collide(const Body& A, const Body& B, Contact* cs)               /* collision.cpp:420-489, flip=false */
{
    using namespace sat;
    Box boxA = makeBox(A), boxB = makeBox(B);
    V3 delta = boxB.center - boxA.center;
    Best bestFace; bestFace.sep = -FLT_MAX; bestFace.valid = false; bestFace.type = 0; bestFace.iA = bestFace.iB = 0; bestFace.nAB = vzero();
    Best bestEdge = bestFace;
    for (int i = 0; i < 3; ++i)
        if (!testAxis(boxA, boxB, delta, boxA.axis[i], 0, i, -1, bestFace)) return 0;
    for (int i = 0; i < 3; ++i)
        if (!testAxis(boxA, boxB, delta, boxB.axis[i], 1, -1, i, bestFace)) return 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            V3 axis = cross(boxA.axis[i], boxB.axis[j]);
            if (!testAxis(boxA, boxB, delta, axis, 2, i, j, bestEdge)) return 0;
        }
    if (!bestFace.valid) return 0;
    Best best = bestFace;
    if (bestEdge.valid) {
        const float relTol = 0.95f, absTol = 0.01f;
        if (relTol * bestEdge.sep > bestFace.sep + absTol) best = bestEdge;
    }
    if (best.type == 2) return edgeContact(A, B, boxA, boxB, best.iA, best.iB, best.nAB, cs);
    if (best.type == 0) return faceManifold(A, B, boxA, boxB, true, best.iA, best.nAB, cs);
    return faceManifold(A, B, boxA, boxB, false, best.iB, best.nAB, cs);
}

/* ------------------------------------------------------------------ manifold.cpp ------- */
namespace mf {
const float NORMAL_MARGIN = 0.01f;            /* manifold.cpp:17 */
const float STICK_MAX_DRIFT = 0.015f;         /* :18 */
const float STICK_MIN_DOT = 0.995f;           /* :19 */
const float WARM_MAX_DRIFT = 0.08f;           /* :20 */
const float WARM_MIN_DOT = 0.9f;              /* :21 */
const float NORMAL_FORCE_CAP = 5000.0f;       /* :22 */

static V3 worldPoint(const Body& b, V3 local) { return b.pos + rotate(b.rot, local); } /* :25-28 */
static V3 normalizeSafe(V3 v, V3 fallback)    /* :30-37 */
{
    float lenSq = lengthSq(v);
    if (lenSq < VEC_EPS) return fallback;
    return v / sqrtf(lenSq);
}
static void basis(V3 nIn, V3& n, V3& t1, V3& t2) /* :39-50 */
{
    n = normalizeSafe(nIn, v3(0.0f, 1.0f, 0.0f));
    if (fabsf(n.x) >= fabsf(n.z)) t1 = v3(-n.y, n.x, 0.0f);
    else t1 = v3(0.0f, -n.z, n.y);
    t1 = normalizeSafe(t1, v3(1.0f, 0.0f, 0.0f));
    t2 = normalizeSafe(cross(n, t1), v3(0.0f, 0.0f, 1.0f));
}
} // namespace mf

/* ------------------------------------------------------------------ World --------------- */
World::World()
    : enableDiagnostics(false), logFrequency(60), stepIndex(0), nextBodyId(1)
{
    std::memset(&last, 0, sizeof(last));
    defaultParams();
}

void World::defaultParams()                    /* solver.cpp:240-253 */
{
    dt = 1.0f / 60.0f;
    gravity = v3(0.0f, -10.0f, 0.0f);
    iterations = 10;
    alpha = 0.95f;
    beta = 100000.0f;
    gamma = 0.99f;
    postStabilize = false;
    if (logFrequency <= 0) logFrequency = 60;
    std::memset(&last, 0, sizeof(last));
}

/* solver.cpp:145-228 — ray pick of the nearest dynamic box.  The reference walks its body list, which is newest
 * first (rigid.cpp:20-21), and keeps a hit only when strictly nearer, so equal distances go to the newer body. */
int World::pick(V3 origin, V3 dir, V3& local) const
{
    const float epsilon = 1.0e-6f;
    float dirLenSq = lengthSq(dir);
    if (dirLenSq < epsilon) return -1;
    V3 rayDir = dir / sqrtf(dirLenSq);
    float bestT = FLT_MAX;
    int bestBody = -1;
    V3 bestLocal = vzero();
    for (int i = (int)bodies.size() - 1; i >= 0; --i) {
        const Body& b = bodies[i];
        if (b.invMass <= 0.0f) continue;
        Q4 invRot = conj(b.rot);
        V3 lo = rotate(invRot, origin - b.pos);
        V3 ld = rotate(invRot, rayDir);
        V3 half = b.size * 0.5f;
        float tEnter = 0.0f, tExit = FLT_MAX;
        bool hit = true;
        for (int axis = 0; axis < 3; ++axis) {
            float o = comp(lo, axis), d = comp(ld, axis), lim = comp(half, axis);
            if (fabsf(d) < epsilon) {
                if (o < -lim || o > lim) { hit = false; break; }
                continue;
            }
            float invD = 1.0f / d;
            float t0 = (-lim - o) * invD, t1 = (lim - o) * invD;
            if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
            tEnter = tEnter > t0 ? tEnter : t0;   /* maths.h:102 max */
            tExit = tExit < t1 ? tExit : t1;      /* maths.h:101 min */
            if (tEnter > tExit) { hit = false; break; }
        }
        if (!hit) continue;
        float tHit = (tEnter >= 0.0f) ? tEnter : tExit;
        if (tHit < 0.0f) continue;
        if (tHit < bestT) { bestT = tHit; bestBody = i; bestLocal = lo + ld * tHit; }
    }
    if (bestBody >= 0) local = bestLocal;
    return bestBody;
}

void World::clear()                            /* solver.cpp:230-238; Rigid::next_id is NOT reset (rigid.cpp:10) */
{
    bodies.clear(); pool.clear(); forceList.clear(); freeSlots.clear();
    stepIndex = 0;
    std::memset(&last, 0, sizeof(last));
}

int World::addBody(V3 size, float density, float friction, V3 pos, Q4 rot, V3 lin, V3 ang)
{
    Body b;
    b.id = nextBodyId++;
    initBoxBody(b, size, density, friction, pos, rot, lin, ang);
    bodies.push_back(b);
    return (int)bodies.size() - 1;
}

int World::newForce(int kind, int a, int b)    /* force.cpp:12-41 (+ manifold.cpp:54-64) */
{
    int id;
    if (!freeSlots.empty()) { id = freeSlots.back(); freeSlots.pop_back(); }
    else { id = (int)pool.size(); pool.emplace_back(); }
    Force& f = pool[id];
    std::memset(&f, 0, sizeof(f));
    f.kind = kind; f.bodyA = a; f.bodyB = b;
    forceList.push_back(id);
    if (a >= 0) bodies[a].forces.push_back(id);
    if (b >= 0) bodies[b].forces.push_back(id);
    for (int i = 0; i < kMaxRows; ++i) {
        f.stiffness[i] = 0.0f; f.lambda[i] = 0.0f; f.penalty[i] = 0.0f; f.motor[i] = 0.0f;
        f.fmin[i] = -FLT_MAX; f.fmax[i] = FLT_MAX; f.C[i] = 0.0f;
    }
    f.numContacts = 0; f.combinedFriction = 0.0f;
    f.q0 = qident();
    return id;
}

static void eraseValue(std::vector<int>& v, int id)
{
    for (size_t i = v.size(); i-- > 0;)
        if (v[i] == id) { v.erase(v.begin() + (long)i); return; }
}

void World::deleteForce(int id)                /* force.cpp:43-69 */
{
    Force& f = pool[id];
    eraseValue(forceList, id);
    if (f.bodyA >= 0) eraseValue(bodies[f.bodyA].forces, id);
    if (f.bodyB >= 0) eraseValue(bodies[f.bodyB].forces, id);
    freeSlots.push_back(id);
}

bool World::isConstrainedTo(int a, int other) const /* rigid.cpp:61-69 */
{
    const std::vector<int>& L = bodies[a].forces;
    for (size_t i = L.size(); i-- > 0;) {
        const Force& f = pool[L[i]];
        if (f.bodyB == other || f.bodyA == other) return true;
    }
    return false;
}

int World::addJoint(int a, int b, V3 rA, V3 rB, float linStiff, float angStiff) /* joint.cpp:11-63 */
{
    int id = newForce(JOINT, a, b);
    Force& f = pool[id];
    if (a >= 0) {
        f.rA = rA; f.rB = rB;
        f.q0 = conj(bodies[a].rot) * bodies[b].rot;
    } else {
        /* world joint: rA is the world anchor, rB the anchor in bodyB's frame (joint.cpp:46-52) */
        f.rA = rA;
        f.rB = transpose(matFromQuat(bodies[b].rot)) * (rA - bodies[b].pos);
        f.q0 = bodies[b].rot;
    }
    for (int i = 0; i < 3; ++i) { f.stiffness[i] = linStiff; f.lambda[i] = 0; f.penalty[i] = kPenaltyMin; }
    for (int i = 3; i < 6; ++i) { f.stiffness[i] = angStiff; f.lambda[i] = 0; f.penalty[i] = kPenaltyMin; }
    return id;
}

int World::addSpring(int a, int b, V3 rA, V3 rB, float stiffness, float rest) /* spring.cpp:10-31 */
{
    int id = newForce(SPRING, a, b);
    Force& f = pool[id];
    f.rA = rA; f.rB = rB; f.restLength = rest;
    f.stiffness[0] = stiffness;
    if (f.restLength < 0) {
        V3 pA = bodies[a].pos + rotate(bodies[a].rot, rA);
        V3 pB = bodies[b].pos + rotate(bodies[b].rot, rB);
        f.restLength = length(pA - pB);
    }
    f.lambda[0] = 0.0f; f.penalty[0] = kPenaltyMin; f.fmin[0] = -FLT_MAX; f.fmax[0] = FLT_MAX;
    return id;
}

int World::addIgnore(int a, int b) { return newForce(IGNORE, a, b); } /* ignorecollision.h:15-16 */

int World::rowCount(const Force& f) const
{
    switch (f.kind) {
    case MANIFOLD: return f.numContacts * 3;     /* manifold.cpp:66-69 */
    case JOINT: return 6;                        /* joint.h:42 */
    case SPRING: return 1;                       /* spring.h:33 */
    default: return 0;                           /* ignorecollision.h:19 */
    }
}

bool World::initialize(Force& f)
{
    if (f.kind != MANIFOLD) return true;         /* joint.cpp:63-66, spring.h:34, ignorecollision.h:20 */
    using namespace mf;
    /* manifold.cpp:71-175 */
    const Body& A = bodies[f.bodyA];
    const Body& B = bodies[f.bodyB];
    f.combinedFriction = sqrtf(A.friction * B.friction);

    Contact oldC[4]; float oldLambda[12], oldPenalty[12];
    bool oldUsed[4] = {false, false, false, false};
    int oldN = f.numContacts;
    for (int i = 0; i < oldN; ++i) {
        oldC[i] = f.contacts[i];
        for (int k = 0; k < 3; ++k) { int r = i * 3 + k; oldLambda[r] = f.lambda[r]; oldPenalty[r] = f.penalty[r]; }
    }
    f.numContacts = collide(A, B, f.contacts);
    if (f.numContacts == 0) return false;

    for (int i = 0; i < f.numContacts; ++i) {
        int base = i * 3;
        for (int k = 0; k < 3; ++k) {
            int r = base + k;
            f.stiffness[r] = FLT_MAX; f.lambda[r] = 0.0f; f.penalty[r] = kPenaltyMin; f.motor[r] = 0.0f;
        }
        f.contacts[i].stick = false;
        int best = -1;
        for (int j = 0; j < oldN; ++j) {
            if (oldUsed[j]) continue;
            if (f.contacts[i].feature == oldC[j].feature) { best = j; break; }
        }
        if (best >= 0) {
            oldUsed[best] = true;
            V3 newN = normalizeSafe(f.contacts[i].normal, v3(0.0f, 1.0f, 0.0f));
            V3 oldNn = normalizeSafe(oldC[best].normal, newN);
            float nd = dot(newN, oldNn);
            V3 oldMid = (worldPoint(A, oldC[best].rA) + worldPoint(B, oldC[best].rB)) * 0.5f;
            V3 newMid = (worldPoint(A, f.contacts[i].rA) + worldPoint(B, f.contacts[i].rB)) * 0.5f;
            float driftSq = lengthSq(newMid - oldMid);
            bool valid = (nd >= WARM_MIN_DOT) && (driftSq <= WARM_MAX_DRIFT * WARM_MAX_DRIFT);
            if (valid) {
                for (int k = 0; k < 3; ++k) {
                    int r = base + k, o = best * 3 + k;
                    f.lambda[r] = oldLambda[o];
                    f.penalty[r] = clampr(oldPenalty[o], kPenaltyMin, MANIFOLD_PENALTY_CAP);
                }
            }
            bool reuse = false;
            if (oldC[best].stick && valid)
                reuse = (nd >= STICK_MIN_DOT) && (driftSq <= STICK_MAX_DRIFT * STICK_MAX_DRIFT);
            f.contacts[i].stick = oldC[best].stick && reuse;
            if (reuse) { f.contacts[i].rA = oldC[best].rA; f.contacts[i].rB = oldC[best].rB; }
        }
        V3 n, t1, t2;
        basis(f.contacts[i].normal, n, t1, t2);
        f.contacts[i].normal = n;
        V3 pA = worldPoint(A, f.contacts[i].rA);
        V3 pB = worldPoint(B, f.contacts[i].rB);
        V3 d = pA - pB;
        f.contacts[i].C0_n = dot(d, n) - NORMAL_MARGIN;
        f.contacts[i].C0_t.x = dot(d, t1);
        f.contacts[i].C0_t.y = dot(d, t2);
        f.contacts[i].C0_t.z = 0.0f;
        f.contacts[i].penetration = fmaxr(0.0f, -dot(d, n));
    }
    return true;
}

void World::computeConstraint(Force& f, float alphaIn)
{
    if (f.kind == MANIFOLD) {                    /* manifold.cpp:177-245 */
        using namespace mf;
        const Body& A = bodies[f.bodyA];
        const Body& B = bodies[f.bodyB];
        float biasScale = clampr(1.0f - alphaIn, 0.0f, 1.0f);
        for (int i = 0; i < f.numContacts; ++i) {
            int base = i * 3;
            Contact& c = f.contacts[i];
            V3 n, t1, t2;
            basis(c.normal, n, t1, t2);
            c.normal = n;
            V3 pA = worldPoint(A, c.rA);
            V3 pB = worldPoint(B, c.rB);
            V3 d = pA - pB;
            float separation = dot(d, n) - NORMAL_MARGIN;
            float slip1 = dot(d, t1);
            float slip2 = dot(d, t2);
            f.C[base + 0] = separation + biasScale * c.C0_n;
            float invMassSum = A.invMass + B.invMass;
            float massScale = (invMassSum > 1.0e-6f) ? (1.0f / invMassSum) : 1.0f;
            float cap = NORMAL_FORCE_CAP * massScale;
            f.fmin[base + 0] = -cap;
            f.fmax[base + 0] = 0.0f;
            f.C[base + 1] = slip1 + biasScale * c.C0_t.x;
            f.C[base + 2] = slip2 + biasScale * c.C0_t.y;
            float warmMag = fabsf(fminr(f.lambda[base + 0], 0.0f));
            float trial = f.penalty[base + 0] * f.C[base + 0] + f.lambda[base + 0];
            float trialMag = fabsf(fminr(trial, 0.0f));
            float nMag = fmaxr(warmMag, trialMag);
            nMag = fminr(nMag, cap);
            float mu = f.combinedFriction;
            if (!c.stick) mu *= 0.9f;
            float limit = mu * nMag;
            float lt1 = f.lambda[base + 1], lt2 = f.lambda[base + 2];
            float tanMag = sqrtf(lt1 * lt1 + lt2 * lt2);
            if (tanMag > limit && tanMag > 1.0e-8f) {
                float s = limit / tanMag;
                f.lambda[base + 1] *= s;
                f.lambda[base + 2] *= s;
            }
            f.fmin[base + 1] = -limit; f.fmax[base + 1] = limit;
            f.fmin[base + 2] = -limit; f.fmax[base + 2] = limit;
            float slipSq = f.C[base + 1] * f.C[base + 1] + f.C[base + 2] * f.C[base + 2];
            float tanSq = f.lambda[base + 1] * f.lambda[base + 1] + f.lambda[base + 2] * f.lambda[base + 2];
            float stickSq = kStickThresh * kStickThresh;
            c.stick = (slipSq <= stickSq) && (tanSq <= limit * limit + 1.0e-8f);
            c.penetration = fmaxr(0.0f, -dot(d, n));
        }
    } else if (f.kind == JOINT) {                /* joint.cpp:68-106 */
        V3 pA, pB; Q4 qA;
        const Body& B = bodies[f.bodyB];
        if (f.bodyA >= 0) { qA = bodies[f.bodyA].rot; pA = bodies[f.bodyA].pos + rotate(qA, f.rA); }
        else { qA = qident(); pA = f.rA; }
        pB = B.pos + rotate(B.rot, f.rB);
        V3 lc = pA - pB;
        f.C[0] = lc.x; f.C[1] = lc.y; f.C[2] = lc.z;
        Q4 cur = conj(qA) * B.rot;
        Q4 dq = cur * conj(f.q0);
        V3 ac = v3(dq.x, dq.y, dq.z) * 2.0f;
        f.C[3] = ac.x; f.C[4] = ac.y; f.C[5] = ac.z;
        for (int i = 0; i < 6; ++i) { f.fmin[i] = -FLT_MAX; f.fmax[i] = FLT_MAX; }
    } else if (f.kind == SPRING) {               /* spring.cpp:33-58 (H_ll is never read: omitted) */
        const Body& B = bodies[f.bodyB];
        Q4 qA = f.bodyA >= 0 ? bodies[f.bodyA].rot : qident();
        V3 pA = f.bodyA >= 0 ? bodies[f.bodyA].pos + rotate(qA, f.rA) : f.rA;
        V3 pB = B.pos + rotate(B.rot, f.rB);
        f.C[0] = length(pA - pB) - f.restLength;
    }
}

void World::computeDerivatives(const Force& f, V3& Jl, V3& Ja, int body, int row) const
{
    if (f.kind == MANIFOLD) {                    /* manifold.cpp:247-271 */
        using namespace mf;
        const Contact& c = f.contacts[row / 3];
        int type = row % 3;
        V3 n, t1, t2;
        basis(c.normal, n, t1, t2);
        V3 b = type == 0 ? n : (type == 1 ? t1 : t2);
        bool isA = body == f.bodyA;
        float sign = isA ? 1.0f : -1.0f;
        V3 wr = isA ? rotate(bodies[f.bodyA].rot, c.rA) : rotate(bodies[f.bodyB].rot, c.rB);
        Jl = b * sign;
        Ja = cross(wr, b) * sign;
    } else if (f.kind == JOINT) {                /* joint.cpp:108-139 */
        Jl = vzero(); Ja = vzero();
        bool isA = body == f.bodyA;
        float sign = isA ? 1.0f : -1.0f;
        if (isA && f.bodyA < 0) return;
        if (row < 3) {
            V3 ax = v3(row == 0 ? 1.0f : 0.0f, row == 1 ? 1.0f : 0.0f, row == 2 ? 1.0f : 0.0f);
            V3 r = isA ? rotate(bodies[body].rot, f.rA) : rotate(bodies[body].rot, f.rB);
            Jl = ax * sign;
            Ja = cross(r, ax) * sign;
        } else {
            int k = row - 3;
            V3 ax = v3(k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f);
            Jl = vzero();
            Ja = ax * sign;
        }
    } else if (f.kind == SPRING) {               /* spring.cpp:61-90 */
        const Body& B = bodies[f.bodyB];
        Q4 qA = f.bodyA >= 0 ? bodies[f.bodyA].rot : qident();
        V3 pA = f.bodyA >= 0 ? bodies[f.bodyA].pos + rotate(qA, f.rA) : f.rA;
        V3 pB = B.pos + rotate(B.rot, f.rB);
        V3 d = pA - pB;
        float len = length(d);
        if (len < VEC_EPS) { Jl = vzero(); Ja = vzero(); return; }
        V3 n = d / len;
        bool isA = body == f.bodyA;
        float sign = isA ? 1.0f : -1.0f;
        Jl = n * sign;
        if (body >= 0) {
            V3 rw = isA ? rotate(bodies[body].rot, f.rA) : rotate(bodies[body].rot, f.rB);
            Ja = cross(rw, n) * sign;
        } else {
            Ja = vzero();
        }
    } else {
        /* ignorecollision.h:22: no-op — outputs are left untouched like the reference */
    }
}

static float rowPenaltyGain(const World& w, const Force& f, int row, float beta) /* solver.cpp:94-125 */
{
    const float angularBetaScale = 0.01f;
    const float epsilon = 1.0e-8f;
    float lw = 0.0f, aw = 0.0f;
    int ends[2] = {f.bodyA, f.bodyB};
    for (int e = 0; e < 2; ++e) {
        if (ends[e] < 0) continue;
        V3 Jl, Ja;
        w.computeDerivatives(f, Jl, Ja, ends[e], row);
        lw += lengthSq(Jl);
        aw += lengthSq(Ja);
    }
    float total = lw + aw;
    if (total < epsilon) return beta;
    float bl = beta;
    float ba = beta * angularBetaScale;
    return (bl * lw + ba * aw) / total;
}

void World::step()                               /* solver.cpp:255-514 */
{
    ++stepIndex;
    Diag stats; std::memset(&stats, 0, sizeof(stats));
    const int nb = (int)bodies.size();

    /* 1. broadphase (solver.cpp:262-270): list order = descending creation index */
    for (int a = nb - 1; a >= 0; --a) {
        for (int b = a - 1; b >= 0; --b) {
            V3 dp = bodies[a].pos - bodies[b].pos;
            float r = bodies[a].radius + bodies[b].radius;
            if (dot(dp, dp) <= r * r && !isConstrainedTo(a, b)) newForce(MANIFOLD, a, b);
        }
    }

    /* 2. initialise + warm-start (solver.cpp:273-296): newest force first */
    for (long i = (long)forceList.size() - 1; i >= 0; --i) {
        int id = forceList[(size_t)i];
        Force& f = pool[id];
        if (!initialize(f)) { deleteForce(id); continue; }
        int rows = rowCount(f);
        for (int r = 0; r < rows; ++r) {
            if (postStabilize) {
                f.penalty[r] = clampr(f.penalty[r] * gamma, kPenaltyMin, kPenaltyMax);
            } else {
                f.lambda[r] *= alpha * gamma;
                f.penalty[r] = clampr(f.penalty[r] * gamma, kPenaltyMin, kPenaltyMax);
            }
            if (f.stiffness[r] > 0.0f && f.stiffness[r] < FLT_MAX) f.penalty[r] = fminr(f.penalty[r], f.stiffness[r]);
        }
    }

    /* 3. predict (solver.cpp:299-337) */
    for (int k = nb - 1; k >= 0; --k) {
        Body& b = bodies[k];
        clampAngularVelocity(b.angVel);
        b.initPos = b.pos; b.initRot = b.rot;
        b.inerPos = b.pos; b.inerRot = b.rot;
        if (b.invMass > 0.0f) {
            ++stats.dynamicBodies;
            sanitizeVec3(b.linVel, "linear velocity", b.id);
            sanitizeVec3(b.angVel, "angular velocity", b.id);
            b.inerPos = b.pos + b.linVel * dt + gravity * (dt * dt);
            Q4 omega = q4(b.angVel.x, b.angVel.y, b.angVel.z, 0.0f);
            b.inerRot = qnormalize(b.rot + (omega * b.rot) * (0.5f * dt));
            float gLen = length(gravity);
            float aw = 0.0f;
            if (gLen > 1e-5f) {
                V3 accel = (b.linVel - b.prevLinVel) / dt;
                float proj = dot(accel, gravity / gLen);
                aw = clampr(proj / gLen, 0.0f, 1.0f);
                if (!std::isfinite(aw)) aw = 0.0f;
            }
            b.pos = b.pos + (b.linVel * dt + gravity * (aw * dt * dt));
            b.rot = b.inerRot;
            sanitizeVec3(b.pos, "predicted position", b.id);
            sanitizeQuat(b.rot, "predicted orientation", b.id);
        } else {
            b.pos = b.inerPos; b.rot = b.inerRot;
        }
    }

    /* 4. iterations (solver.cpp:340-431) */
    int total = iterations + (postStabilize ? 1 : 0);
    for (int it = 0; it < total; ++it) {
        float curAlpha = postStabilize ? (it < iterations ? 1.0f : 0.0f) : alpha;

        for (int k = nb - 1; k >= 0; --k) {      /* primal, solver.cpp:344-409 */
            Body& b = bodies[k];
            if (b.invMass <= 0.0f) continue;
            M3 ll = mzero(), la = mzero(), al = mzero(), aa = mzero();
            V3 rl = vzero(), ra = vzero();
            M3 massMat = mdiag(v3(b.mass, b.mass, b.mass));
            M3 Iw = worldInertia(b);
            float invDt2 = 1.0f / (dt * dt);
            addTo(ll, massMat * invDt2);
            addTo(aa, Iw * invDt2);
            rl = massMat * ((b.pos - b.inerPos) * invDt2);
            Q4 qErr = b.rot * conj(b.inerRot);
            V3 rotErr = v3(qErr.x, qErr.y, qErr.z);
            rotErr = rotErr * 2.0f;
            if (qErr.w < 0.0f) rotErr = -rotErr;
            ra = Iw * (rotErr * invDt2);

            for (size_t fi = b.forces.size(); fi-- > 0;) {
                Force& f = pool[b.forces[fi]];
                computeConstraint(f, curAlpha);
                int rows = rowCount(f);
                for (int r = 0; r < rows; ++r) {
                    V3 Jl, Ja;
                    computeDerivatives(f, Jl, Ja, k, r);
                    float lw = (f.stiffness[r] == FLT_MAX) ? f.lambda[r] : 0.0f;
                    float desired = f.penalty[r] * f.C[r] + lw + f.motor[r];
                    float fr = clampr(desired, f.fmin[r], f.fmax[r]);
                    rl = rl + Jl * fr;
                    ra = ra + Ja * fr;
                    float pen = f.penalty[r];
                    if (pen > 0.0f && std::isfinite(pen)) {
                        addTo(ll, outerSolver(Jl, Jl) * pen);
                        addTo(la, outerSolver(Jl, Ja) * pen);
                        addTo(al, outerSolver(Ja, Jl) * pen);
                        addTo(aa, outerSolver(Ja, Ja) * pen);
                        if (f.kind == MANIFOLD) {
                            M3 invIw = worldInvInertia(b);
                            V3 gyro = vabs(cross(Ja, invIw * Ja)) * fabsf(fr);
                            addTo(aa, mdiag(gyro));
                        }
                    }
                }
            }
            V3 dl, da;
            solve6(ll, la, al, aa, rl, ra, dl, da);
            b.pos = b.pos - dl;
            Q4 dq = q4(da.x, da.y, da.z, 0.0f);
            b.rot = qnormalize(b.rot - (dq * b.rot) * 0.5f);
            sanitizeVec3(b.pos, "position", b.id);
            sanitizeQuat(b.rot, "orientation", b.id);
        }

        if (it < iterations) {                   /* dual, solver.cpp:411-430 */
            for (size_t i = forceList.size(); i-- > 0;) {
                Force& f = pool[forceList[i]];
                computeConstraint(f, curAlpha);
                int rows = rowCount(f);
                for (int r = 0; r < rows; ++r) {
                    if (f.stiffness[r] != FLT_MAX) continue;
                    float lu = clampr(f.penalty[r] * f.C[r] + f.lambda[r], f.fmin[r], f.fmax[r]);
                    bool active = lu > f.fmin[r] && lu < f.fmax[r];
                    f.lambda[r] = lu;
                    if (active) {
                        float betaRow = rowPenaltyGain(*this, f, r, beta);
                        float cap = (f.kind == MANIFOLD) ? MANIFOLD_PENALTY_CAP : kPenaltyMax;
                        f.penalty[r] = fminr(f.penalty[r] + betaRow * fabsf(f.C[r]), cap);
                    }
                }
            }
        }
    }

    /* 5. velocities (solver.cpp:434-469) */
    const float linearDamping = 0.995f, angularDamping = 0.97f;
    for (int k = nb - 1; k >= 0; --k) {
        Body& b = bodies[k];
        if (b.invMass <= 0.0f) continue;
        b.prevLinVel = b.linVel; b.prevAngVel = b.angVel;
        b.linVel = (b.pos - b.initPos) / dt;
        Q4 dq = b.rot * conj(b.initRot);
        V3 av = v3(dq.x, dq.y, dq.z);
        av = av * (2.0f / dt);
        if (dq.w < 0.0f) av = -av;
        b.angVel = av;
        b.linVel = b.linVel * linearDamping;
        b.angVel = b.angVel * angularDamping;
        sanitizeVec3(b.linVel, "linear velocity", b.id);
        sanitizeVec3(b.angVel, "angular velocity", b.id);
        stats.maxLinearSpeed = fmaxr(stats.maxLinearSpeed, length(b.linVel));
        stats.maxAngularSpeed = fmaxr(stats.maxAngularSpeed, length(b.angVel));
    }

    /* 6. diagnostics (solver.cpp:472-497) */
    for (size_t i = forceList.size(); i-- > 0;) {
        const Force& f = pool[forceList[i]];
        if (f.kind != MANIFOLD) continue;
        stats.activeManifolds++;
        stats.activeContacts += f.numContacts;
        const Body& A = bodies[f.bodyA];
        const Body& B = bodies[f.bodyB];
        for (int c = 0; c < f.numContacts; ++c) {
            const Contact& ct = f.contacts[c];
            V3 pA = A.pos + rotate(A.rot, ct.rA);
            V3 pB = B.pos + rotate(B.rot, ct.rB);
            float separation = dot(pA - pB, ct.normal);
            float pen = fmaxr(0.0f, -separation);
            float viol = fmaxr(0.0f, kPenetrationSlop - separation);
            stats.maxPenetration = fmaxr(stats.maxPenetration, pen);
            stats.maxConstraintViolation = fmaxr(stats.maxConstraintViolation, viol);
            stats.maxNormalImpulse = fmaxr(stats.maxNormalImpulse, fabsf(f.lambda[c * 3 + 0]));
        }
    }
    last = stats;

    if (enableDiagnostics) {                     /* solver.cpp:499-513 */
        int freq = logFrequency > 0 ? logFrequency : 1;
        if (stepIndex % freq == 0) {
            std::printf("[Physics] step %d | manifolds: %d | contacts: %d | dyn bodies: %d | maxPen: %.6f | maxDrift: %.6f | maxLin: %.3f | maxAng: %.3f | maxLambda: %.3f\n",
                        stepIndex, stats.activeManifolds, stats.activeContacts, stats.dynamicBodies, stats.maxPenetration,
                        stats.maxConstraintViolation, stats.maxLinearSpeed, stats.maxAngularSpeed, stats.maxNormalImpulse);
        }
    }
}

/* ------------------------------------------------------------------ scenes.h ----------- */
const char* const kSceneNames[9] = {"Empty", "Ground", "Stack", "Pyramid", "Wall", "TwoBlockDrop",
                                    "Stress1000", "Rod (WIP)", "Soft Body (WIP)"};   /* scenes.h:200-210 */

static Q4 quatAxisAngle(V3 axis, float angle)    /* maths.h:58 */
{
    float h = angle * 0.5f;
    float s = sinf(h);
    Q4 q; q.w = cosf(h); q.x = axis.x * s; q.y = axis.y * s; q.z = axis.z * s;
    return q;
}

static void ground(World& w)                     /* scenes.h:27-31 */
{
    w.clear();
    w.addBody(v3(100, 1, 100), 0.0f, 0.5f, v3(0, -0.5f, 0), qident(), vzero(), vzero());
}

static unsigned hashU(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static float hash01(unsigned x) { return (hashU(x) & 0x00FFFFFFU) / 16777215.0f; }  /* scenes.h:108-115 */

static void stressGrid(World& w, int nx, int ny, int nz, bool packed) /* scenes.h:87-132 with NX,NY,NZ = nx,ny,nz */
{
    ground(w);
    w.iterations = 20; w.beta = 30000.0f; w.gamma = 0.995f;
    const float spacingXZ = 1.15f;
    const float spacingY = packed ? 1.02f : 2.0f;
    const float startY = packed ? 0.52f : 20.0f;
    const float jitterXZ = 0.04f;
    const float jitterY = packed ? 0.0f : 0.25f;
    for (int y = 0; y < ny; ++y)
        for (int z = 0; z < nz; ++z)
            for (int x = 0; x < nx; ++x) {
                unsigned seed = (unsigned)(x + nx * (z + nz * y) + 1);
                float jx = (hash01(seed * 9781U) * 2.0f - 1.0f) * jitterXZ;
                float jz = (hash01(seed * 6271U) * 2.0f - 1.0f) * jitterXZ;
                float jy = hash01(seed * 3343U) * jitterY;
                float px = (x - (nx - 1) * 0.5f) * spacingXZ + jx;
                float py = startY + y * spacingY + jy;
                float pz = (z - (nz - 1) * 0.5f) * spacingXZ + jz;
                w.addBody(v3(1, 1, 1), 1.0f, 0.5f, v3(px, py, pz), qident(), vzero(), vzero());
            }
}

static void stressN(World& w, int n, bool packed) { stressGrid(w, n, n, n, packed); }

int buildScene(World& w, const char* name)
{
    if (!std::strncmp(name, "StressGrid:", 11) || !std::strncmp(name, "PackedGrid:", 11)) {
        int nx = 0, ny = 0, nz = 0;
        if (std::sscanf(name + 11, "%d,%d,%d", &nx, &ny, &nz) != 3 || nx < 1 || ny < 1 || nz < 1) return -1;
        stressGrid(w, nx, ny, nz, name[0] == 'P');
        return 102;
    }
    if (!std::strncmp(name, "StressN:", 8)) { stressN(w, std::atoi(name + 8), false); return 100; }
    if (!std::strncmp(name, "Packed:", 7)) { stressN(w, std::atoi(name + 7), true); return 101; }
    int idx = -1;
    for (int i = 0; i < 9; ++i) if (!std::strcmp(kSceneNames[i], name)) { idx = i; break; }
    switch (idx) {
    case 0: w.clear(); break;                                             /* scenes.h:23-25 */
    case 1: ground(w); break;
    case 2:                                                               /* scenes.h:33-40 */
        ground(w);
        for (int i = 0; i < 10; ++i) w.addBody(v3(1, 1, 1), 1.0f, 0.5f, v3(0, i * 1.1f + 0.5f, 0), qident(), vzero(), vzero());
        break;
    case 3: {                                                             /* scenes.h:42-53 */
        ground(w);
        const int S = 10;
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S - y; ++x) {
                float xp = (x - (S - y - 1) * 0.5f) * 1.1f;
                float yp = y * 1.05f + 0.5f;
                w.addBody(v3(1, 1, 1), 1.0f, 0.5f, v3(xp, yp, 0), qident(), vzero(), vzero());
            }
        break;
    }
    case 4: {                                                             /* scenes.h:55-72 */
        ground(w);
        const int W = 8, H = 8;
        const V3 brick = v3(1.0f, 0.5f, 0.5f);
        const float spacingX = 1.03f, spacingY = 0.52f;
        const float baseY = brick.y * 0.5f;
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                float xo = (i % 2 == 0) ? 0.0f : 0.5f * spacingX;
                float x = (j - (W - 1) * 0.5f) * spacingX + xo;
                float y = i * spacingY + baseY;
                w.addBody(brick, 1.0f, 0.4f, v3(x, y, -5), qident(), vzero(), vzero());
            }
        break;
    }
    case 5: {                                                             /* scenes.h:74-85 */
        ground(w);
        w.addBody(v3(1, 1, 1), 1.0f, 0.5f, v3(0.0f, 0.5f, 0.0f), qident(), vzero(), vzero());
        Q4 tilt = quatAxisAngle(v3(0.0f, 0.0f, 1.0f), 0.45f);
        w.addBody(v3(1, 1, 1), 1.0f, 0.5f, v3(0.18f, 2.2f, 0.0f), tilt, vzero(), v3(0.0f, 0.0f, 1.0f));
        break;
    }
    case 6: stressN(w, 10, false); break;                                 /* scenes.h:87-132 */
    case 7: {                                                             /* scenes.h:138-152 */
        w.clear();
        for (int i = 0; i < 15; ++i)
            w.addBody(v3(0.25f, 1, 0.25f), (i == 0) ? 0.0f : 1.0f, 0.5f, v3(0, 10.0f - i * 1.0f, 0), qident(), vzero(), vzero());
        break;
    }
    case 8: {                                                             /* scenes.h:154-181 */
        ground(w);
        const int W = 10, H = 10;
        for (int i = 0; i < W; ++i)
            for (int j = 0; j < H; ++j)
                w.addBody(v3(0.5f, 0.5f, 0.5f), 1.0f, 0.3f, v3(i * 0.6f - W * 0.3f, j * 0.6f + 2.0f, 0), qident(), vzero(), vzero());
        break;
    }
    default: return -1;
    }
    return idx;
}

} // namespace orc
