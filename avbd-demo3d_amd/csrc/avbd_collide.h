/*
 * avbd_collide.h — OBB-vs-OBB SAT narrow phase producing <=4 contact points with feature keys.
 *
 * Device restatement of Manifold::collide and its helpers (collision.cpp:56-489; flip=false is
 * the only mode Solver::step uses).  One lane evaluates one pair; everything lives in registers
 * / private scratch (two 16-vertex clip polygons).  Arithmetic follows the reference expression
 * by expression — see avbd_math.h for the bit-exactness contract.
 */
#pragma once
#include "avbd_math.h"

#pragma clang fp contract(off)

namespace avbd {

constexpr int   kMaxContacts     = 4;        /* collision.cpp:18 */
constexpr int   kMaxPolyVerts    = 16;       /* collision.cpp:19 */
constexpr float kSatAxisEps      = 1.0e-6f;  /* collision.cpp:20 */
constexpr float kPlaneEps        = 1.0e-5f;  /* collision.cpp:21 */
constexpr float kMergeDistSq     = 1.0e-6f;  /* collision.cpp:22 */
constexpr float kCollisionMargin = 0.02f;    /* solver.h:28 = CONTACT_PERSISTENCE_MARGIN collision.cpp:23 */

struct BoxPose { f3 pos; q4 rot; f3 size; };

struct RawContact {      /* what addContact (collision.cpp:176-206) fills in */
    int feature;
    f3  rA, rB, normal;
    float penetration;
};

struct Obb { f3 center, half, ax0, ax1, ax2; };                  /* collision.cpp:31-35 */

/* (component-wise selects of VALUES: `i == 0 ? b.ax0 : ...` on the structs becomes a select of addresses followed by a load, which
 * forces both boxes of a pair into scratch memory on the device) */
AVBD_HD f3 obb_axis(const Obb& b, int i)
{
    const float x0 = b.ax0.x, y0 = b.ax0.y, z0 = b.ax0.z, x1 = b.ax1.x, y1 = b.ax1.y, z1 = b.ax1.z, x2 = b.ax2.x, y2 = b.ax2.y, z2 = b.ax2.z;
    return mk3(i == 0 ? x0 : (i == 1 ? x1 : x2), i == 0 ? y0 : (i == 1 ? y1 : y2), i == 0 ? z0 : (i == 1 ? z1 : z2));
}

AVBD_HD Obb make_obb(const BoxPose& p)                           /* collision.cpp:56-66 */
{
    Obb o;
    o.center = p.pos;
    o.half = p.size * 0.5f;
    m33 R = quat_to_mat(p.rot);
    o.ax0 = R.c0; o.ax1 = R.c1; o.ax2 = R.c2;
    return o;
}

AVBD_HD float abs_dot(f3 a, f3 b) { return fabsf(dot3(a, b)); }  /* collision.cpp:68-71 */

AVBD_HD float half_extent(const Obb& b, int i) { const float hx = b.half.x, hy = b.half.y, hz = b.half.z; return i == 0 ? hx : (i == 1 ? hy : hz); }

AVBD_HD void face_axes(const Obb& b, int ai, f3& u, f3& v, float& eU, float& eV) /* collision.cpp:73-91 */
{
    /* face ai spans the two other axes, lower index first: (1,2) (0,2) (0,1) */
    const int iu = ai == 0 ? 1 : 0, iv = ai == 2 ? 1 : 2;
    u = obb_axis(b, iu); v = obb_axis(b, iv);
    eU = half_extent(b, iu); eV = half_extent(b, iv);
}

struct SatBest { int type, iA, iB; float sep; f3 nAB; bool valid; }; /* collision.cpp:37-44 */

/* collision.cpp:208-247.  Returns false when the axis separates by more than the margin. */
AVBD_HD bool sat_test_axis(const Obb& a, const Obb& b, f3 delta, f3 axis, int type, int iA, int iB, SatBest& best)
{
    float lenSq = lensq3(axis);
    if (lenSq < kSatAxisEps) return true;
    f3 n = axis / sqrtf(lenSq);
    if (dot3(n, delta) < 0.0f) n = -n;
    float distance = fabsf(dot3(delta, n));
    float rA = a.half.x * abs_dot(n, a.ax0) + a.half.y * abs_dot(n, a.ax1) + a.half.z * abs_dot(n, a.ax2);
    float rB = b.half.x * abs_dot(n, b.ax0) + b.half.y * abs_dot(n, b.ax1) + b.half.z * abs_dot(n, b.ax2);
    float separation = distance - (rA + rB);
    if (separation > kCollisionMargin) return false;
    if (!best.valid || separation > best.sep) {
        best.valid = true; best.type = type; best.iA = iA; best.iB = iB; best.sep = separation; best.nAB = n;
    }
    return true;
}

/* collision.cpp:136-174 — Sutherland–Hodgman clip of a convex polygon against one plane. */
/* `cap`: vertices the output holds.  The reference's buffers hold kMaxPolyVerts = 16 and are never full: a quad clipped against four half-planes has
 * at most eight vertices (each clip of a convex polygon adds at most one).  The one-lane-per-pair narrow phase keeps buffers of kLanePolyVerts = 8 in
 * LDS (it is what bounds how many pairs a CU holds); should rounding ever produce a ninth vertex, `over` is raised — the step then fails with a capacity
 * error instead of silently dropping a vertex the reference would have kept. */
constexpr int kLanePolyVerts = 8;
AVBD_HD int clip_polygon(const f3* in, int n, f3 pn, float po, f3* out, int cap, int& over)
{
    if (n <= 0) return 0;
    int m = 0;
    f3 a = in[n - 1];
    float da = dot3(pn, a) - po;
    for (int i = 0; i < n; ++i) {
        f3 b = in[i];
        float db = dot3(pn, b) - po;
        bool aIn = da <= kPlaneEps;
        bool bIn = db <= kPlaneEps;
        if (aIn != bIn) {
            float t = 0.0f;
            float denom = da - db;
            if (fabsf(denom) > kSatAxisEps) t = rclamp(da / denom, 0.0f, 1.0f);
            if (m < cap) out[m++] = a + (b - a) * t; else if (m < kMaxPolyVerts) over = 1;
        }
        if (bIn) { if (m < cap) out[m++] = b; else if (m < kMaxPolyVerts) over = 1; }
        a = b; da = db;
    }
    return m;
}

/* collision.cpp:176-206 */
AVBD_HD bool add_contact(const BoxPose& A, const BoxPose& B, RawContact* cs, int& count, f3* mids, f3 xA, f3 xB, int key, f3 nBA)
{
    f3 mid = (xA + xB) * 0.5f;
    for (int i = 0; i < count; ++i) {
        f3 d = mid - mids[i];
        if (lensq3(d) < kMergeDistSq) return false;
    }
    if (count >= kMaxContacts) return false;
    RawContact& c = cs[count];
    c.feature = key;
    c.rA = qrotate(qconj(A.rot), xA - A.pos);
    c.rB = qrotate(qconj(B.rot), xB - B.pos);
    c.normal = nBA;
    c.penetration = rmax(0.0f, -dot3(xA - xB, nBA));
    mids[count] = mid;
    ++count;
    return true;
}

/* working storage of the manifold builders: indexed dynamically, so a kernel keeps it in LDS (scratch otherwise) */
struct CollideScratch { f3 clip0[kLanePolyVerts], clip1[kLanePolyVerts]; f3 mids[kMaxContacts]; int overflow; };

/* collision.cpp:313-394 — reference-face clipping manifold. */
AVBD_HD int build_face_manifold(const BoxPose& A, const BoxPose& B, const Obb& boxA, const Obb& boxB, bool refIsA,
                                int refAxis, f3 nAB, RawContact* cs, CollideScratch& ws)
{
    const Obb& refBox = refIsA ? boxA : boxB;
    const Obb& incBox = refIsA ? boxB : boxA;
    f3 outward = refIsA ? nAB : -nAB;
    f3 nBA = -nAB;

    /* buildFaceFrame, collision.cpp:93-100 */
    f3 rax = obb_axis(refBox, refAxis);
    float fsign = dot3(outward, rax) >= 0.0f ? 1.0f : -1.0f;
    f3 fN = rax * fsign;
    f3 fC = refBox.center + fN * half_extent(refBox, refAxis);
    f3 fU, fV; float eU, eV;
    face_axes(refBox, refAxis, fU, fV, eU, eV);

    /* chooseIncidentFaceAxis, collision.cpp:102-116 */
    int incAxis = 0;
    {
        float bestD = -FLT_MAX;
        for (int i = 0; i < 3; ++i) {
            float d = abs_dot(obb_axis(incBox, i), fN);
            if (d > bestD) { bestD = d; incAxis = i; }
        }
    }

    f3* clip0 = ws.clip0; f3* clip1 = ws.clip1;
    {   /* buildIncidentFace, collision.cpp:118-134 */
        f3 iax = obb_axis(incBox, incAxis);
        float s = dot3(iax, fN) > 0.0f ? -1.0f : 1.0f;
        f3 n = iax * s;
        f3 c = incBox.center + n * half_extent(incBox, incAxis);
        f3 u, v; float xu, xv;
        face_axes(incBox, incAxis, u, v, xu, xv);
        clip0[0] = c + u * xu + v * xv;
        clip0[1] = c - u * xu + v * xv;
        clip0[2] = c - u * xu - v * xv;
        clip0[3] = c + u * xu - v * xv;
    }
    int count = 4;
    f3 n0 = fU;   float o0 = dot3(n0, fC) + eU;
    count = clip_polygon(clip0, count, n0, o0, clip1, kLanePolyVerts, ws.overflow); if (!count) return 0;
    f3 n1 = -fU;  float o1 = dot3(n1, fC) + eU;
    count = clip_polygon(clip1, count, n1, o1, clip0, kLanePolyVerts, ws.overflow); if (!count) return 0;
    f3 n2 = fV;   float o2 = dot3(n2, fC) + eV;
    count = clip_polygon(clip0, count, n2, o2, clip1, kLanePolyVerts, ws.overflow); if (!count) return 0;
    f3 n3 = -fV;  float o3 = dot3(n3, fC) + eV;
    count = clip_polygon(clip1, count, n3, o3, clip0, kLanePolyVerts, ws.overflow); if (!count) return 0;

    int cc = 0;
    f3* mids = ws.mids;
    int prefix = (refIsA ? 0 : 1) << 24;
    prefix |= (refAxis & 0xFF) << 16;
    prefix |= (incAxis & 0xFF) << 8;
    for (int i = 0; i < count && cc < kMaxContacts; ++i) {
        f3 pInc = clip0[i];
        float distance = dot3(pInc - fC, fN);
        if (distance > kCollisionMargin) continue;
        f3 pRef = pInc - fN * distance;
        f3 xA = refIsA ? pRef : pInc;
        f3 xB = refIsA ? pInc : pRef;
        f3 rel = pRef - fC;
        float uC = dot3(rel, fU), vC = dot3(rel, fV);
        float uN = (eU > kSatAxisEps) ? (uC / eU) : 0.0f;
        float vN = (eV > kSatAxisEps) ? (vC / eV) : 0.0f;
        int qU = (int)floorf(rclamp((uN + 1.0f) * 7.5f, 0.0f, 15.0f));
        int qV = (int)floorf(rclamp((vN + 1.0f) * 7.5f, 0.0f, 15.0f));
        int key = prefix | ((qU & 0x0F) << 4) | (qV & 0x0F);
        add_contact(A, B, cs, cc, mids, xA, xB, key, nBA);
    }
    return cc;
}

/* collision.cpp:249-263 */
AVBD_HD void support_edge(const Obb& b, int ai, f3 dir, f3& e0, f3& e1)
{
    int a1 = (ai + 1) % 3, a2 = (ai + 2) % 3;
    f3 x1 = obb_axis(b, a1), x2 = obb_axis(b, a2), x0 = obb_axis(b, ai);
    float s1 = dot3(dir, x1) >= 0.0f ? 1.0f : -1.0f;
    float s2 = dot3(dir, x2) >= 0.0f ? 1.0f : -1.0f;
    f3 ec = b.center + x1 * (half_extent(b, a1) * s1) + x2 * (half_extent(b, a2) * s2);
    e0 = ec - x0 * half_extent(b, ai);
    e1 = ec + x0 * half_extent(b, ai);
}

/* collision.cpp:265-311 */
AVBD_HD void closest_on_segments(f3 p0, f3 p1, f3 q0, f3 q1, f3& c0, f3& c1)
{
    f3 d1 = p1 - p0, d2 = q1 - q0, r = p0 - q0;
    float a = dot3(d1, d1), e = dot3(d2, d2), f = dot3(d2, r);
    float s = 0.0f, t = 0.0f;
    if (a <= kSatAxisEps && e <= kSatAxisEps) { c0 = p0; c1 = q0; return; }
    if (a <= kSatAxisEps) {
        t = rclamp(f / e, 0.0f, 1.0f);
    } else {
        float c = dot3(d1, r);
        if (e <= kSatAxisEps) {
            s = rclamp(-c / a, 0.0f, 1.0f);
        } else {
            float b = dot3(d1, d2);
            float denom = a * e - b * b;
            if (fabsf(denom) > kSatAxisEps) s = rclamp((b * f - c * e) / denom, 0.0f, 1.0f);
            t = (b * s + f) / e;
            if (t < 0.0f) { t = 0.0f; s = rclamp(-c / a, 0.0f, 1.0f); }
            else if (t > 1.0f) { t = 1.0f; s = rclamp((b - c) / a, 0.0f, 1.0f); }
        }
    }
    c0 = p0 + d1 * s;
    c1 = q0 + d2 * t;
}

/* collision.cpp:396-416 */
AVBD_HD int build_edge_contact(const BoxPose& A, const BoxPose& B, const Obb& boxA, const Obb& boxB, int axA, int axB,
                               f3 nAB, RawContact* cs, CollideScratch& ws)
{
    f3 a0, a1, b0, b1;
    support_edge(boxA, axA, nAB, a0, a1);
    support_edge(boxB, axB, -nAB, b0, b1);
    f3 xA, xB;
    closest_on_segments(a0, a1, b0, b1, xA, xB);
    int cc = 0;
    f3* mids = ws.mids;
    int key = (2 << 24) | ((axA & 0xFF) << 8) | (axB & 0xFF);
    add_contact(A, B, cs, cc, mids, xA, xB, key, -nAB);
    return cc;
}

/* Manifold::collide, collision.cpp:420-489 with flip=false. */
AVBD_HD int collide_boxes(const BoxPose& A, const BoxPose& B, RawContact* cs, CollideScratch& ws)
{
    ws.overflow = 0;
    Obb boxA = make_obb(A), boxB = make_obb(B);
    f3 delta = boxB.center - boxA.center;
    SatBest bestFace; bestFace.sep = -FLT_MAX; bestFace.valid = false; bestFace.type = 0; bestFace.iA = 0; bestFace.iB = 0; bestFace.nAB = zero3();
    SatBest bestEdge = bestFace;

    for (int i = 0; i < 3; ++i)
        if (!sat_test_axis(boxA, boxB, delta, obb_axis(boxA, i), 0, i, -1, bestFace)) return 0;
    for (int i = 0; i < 3; ++i)
        if (!sat_test_axis(boxA, boxB, delta, obb_axis(boxB, i), 1, -1, i, bestFace)) return 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            f3 axis = cross3(obb_axis(boxA, i), obb_axis(boxB, j));
            if (!sat_test_axis(boxA, boxB, delta, axis, 2, i, j, bestEdge)) return 0;
        }
    if (!bestFace.valid) return 0;

    SatBest best = bestFace;
    if (bestEdge.valid) {
        const float relTol = 0.95f, absTol = 0.01f;          /* collision.cpp:463-467 */
        if (relTol * bestEdge.sep > bestFace.sep + absTol) best = bestEdge;
    }
    if (best.type == 2) return build_edge_contact(A, B, boxA, boxB, best.iA, best.iB, best.nAB, cs, ws);
    /* one inlined copy of the face builder for both roles (lanes of a wave mix them) */
    const bool refIsA = best.type == 0;
    return build_face_manifold(A, B, boxA, boxB, refIsA, refIsA ? best.iA : best.iB, best.nAB, cs, ws);
}

AVBD_HD int collide_boxes(const BoxPose& A, const BoxPose& B, RawContact* cs)
{
    CollideScratch ws;
    return collide_boxes(A, B, cs, ws);
}

} // namespace avbd
