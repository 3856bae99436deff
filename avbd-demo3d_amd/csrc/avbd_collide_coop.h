/*
 * avbd_collide_coop.h — wave-cooperative narrow phase: SIXTEEN LANES PER PAIR (four pairs per wavefront).
 *
 * Manifold::initialize + Manifold::collide (manifold.cpp:71-175, collision.cpp:56-489) for one candidate / persistent pair, with
 * the lanes of a 16-lane group spread over what is independent inside a pair:
 *   SAT            lane k < 15 tests axis k (3 face axes of A, 3 of B, 9 edge-edge cross products); the winner is then chosen
 *                  by replaying the reference's loop over the fifteen results in its order (first strict maximum wins,
 *                  collision.cpp:239-245), so ties and skipped axes resolve exactly as there;
 *   clipping       Sutherland-Hodgman against the four side planes, one lane per polygon EDGE: a lane emits (intersection?,
 *                  end vertex?) and a 16-lane prefix sum gives the output positions, i.e. the reference's output order
 *                  (collision.cpp:136-174), the 16-vertex cap included;
 *   contacts       one lane per clipped vertex for the projection / feature key; the merge-by-distance and the four-contact
 *                  cap are order dependent (collision.cpp:176-206) and are replayed serially over the <= 16 candidates;
 *   initialize     the feature matching consumes old contacts in order (manifold.cpp:98-110) and is replayed serially (<= 4 x 4
 *                  compares); everything per contact after it — warm-start gates, stick anchors, tangent basis, C0 — runs
 *                  one lane per contact, and that lane also stores its contact.
 * Same expression trees as avbd_collide.h / avbd_rows.h (which stay as the one-lane form behind the known-answer entry point
 * of the host mirror); pinned by the same vectors: 400 reference pairs, 4 000 random pairs against the oracle, every scene.
 * gfx950 only (cross-lane: ds_bpermute shuffles and ballots inside a 64-wide wavefront).
 */
#pragma once
#include "avbd_rows.h"

#pragma clang fp contract(off)

namespace avbd {

/* per 16-lane group, in LDS: everything that is indexed dynamically */
struct CoopScratch {
    f3 clip[2][kMaxPolyVerts];
    RawContact raw[kMaxContacts];
    /* the previous step's contacts while the new ones are matched against them (manifold.cpp:76-84) */
    int   oldFeature[4], oldStick[4];
    f3    oldRA[4], oldRB[4], oldNormal[4];
    float oldLambda[12], oldPenalty[12];
};

__device__ __forceinline__ f3 sel3(bool c, f3 a, f3 b) { return mk3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
__device__ __forceinline__ Obb obb_select(bool c, const Obb& a, const Obb& b)
{
    Obb r; r.center = sel3(c, a.center, b.center); r.half = sel3(c, a.half, b.half); r.ax0 = sel3(c, a.ax0, b.ax0); r.ax1 = sel3(c, a.ax1, b.ax1); r.ax2 = sel3(c, a.ax2, b.ax2);
    return r;
}
__device__ __forceinline__ float gshfl(float v, int srcLane) { return __shfl(v, srcLane, 64); }
__device__ __forceinline__ int   gshfl(int v, int srcLane) { return __shfl(v, srcLane, 64); }
__device__ __forceinline__ f3    gshfl(f3 v, int srcLane) { return mk3(gshfl(v.x, srcLane), gshfl(v.y, srcLane), gshfl(v.z, srcLane)); }
__device__ __forceinline__ unsigned gballot16(bool p, int gBase) { return (unsigned)((__ballot(p) >> gBase) & 0xFFFFull); }
#ifdef AVBD_DF_STATS
#define COOP_STAMP(k) do { if (dbg) dbg[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define COOP_DBG_PARAM , unsigned long long* dbg = nullptr
#define COOP_DBG_ARG , dbg
#else
#define COOP_STAMP(k) do { } while (0)
#define COOP_DBG_PARAM
#define COOP_DBG_ARG
#endif
__device__ __forceinline__ void gsync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

/* collision.cpp:136-174 with one lane per polygon edge; returns the new vertex count (uniform over the group) */
__device__ __forceinline__ int coop_clip(const f3* in, int n, f3 pn, float po, f3* out, int gl, int gBase)
{
    if (n <= 0) return 0;
    int emitA = 0, emitB = 0;
    f3 vA = zero3(), vB = zero3();
    if (gl < n) {
        const f3 a = in[gl == 0 ? n - 1 : gl - 1], b = in[gl];
        const float da = dot3(pn, a) - po, db = dot3(pn, b) - po;
        const bool aIn = da <= kPlaneEps, bIn = db <= kPlaneEps;
        if (aIn != bIn) {
            float t = 0.0f;
            const float denom = da - db;
            if (fabsf(denom) > kSatAxisEps) t = rclamp(da / denom, 0.0f, 1.0f);
            vA = a + (b - a) * t;
            emitA = 1;
        }
        if (bIn) { vB = b; emitB = 1; }
    }
    /* vertices emitted by the lanes in front of this one = position in the reference's output: two ballots instead of a prefix sum
     * through the LDS crossbar (four dependent round trips per clip, four clips per pair) */
    const unsigned mA = gballot16(emitA != 0, gBase), mB = gballot16(emitB != 0, gBase), below = (1u << gl) - 1u;
    const int pos = __popc(mA & below) + __popc(mB & below);
    if (emitA && pos < kMaxPolyVerts) out[pos] = vA;                 /* (every push checks the cap on its own, like the reference) */
    if (emitB && pos + emitA < kMaxPolyVerts) out[pos + emitA] = vB;
    const int total = __popc(mA) + __popc(mB);
    gsync();
    return total < kMaxPolyVerts ? total : kMaxPolyVerts;
}

/* Manifold::collide (collision.cpp:420-489, flip = false).  Contacts land in ws.raw; the count is uniform over the group. */
__device__ __forceinline__ int collide_boxes_coop(const BoxPose& A, const BoxPose& B, CoopScratch& ws, int gl, int gBase COOP_DBG_PARAM)
{
    const Obb boxA = make_obb(A), boxB = make_obb(B);
    const f3 delta = boxB.center - boxA.center;

    /* ---- SAT: lane k tests axis k ---- */
    f3 axis = zero3();
    if (gl < 3) axis = obb_axis(boxA, gl);
    else if (gl < 6) axis = obb_axis(boxB, gl - 3);
    else if (gl < 15) axis = cross3(obb_axis(boxA, (gl - 6) / 3), obb_axis(boxB, (gl - 6) % 3));
    bool tested = false, separating = false;
    float separation = 0.0f;
    f3 n = zero3();
    if (gl < 15) {                                                      /* testAxis, collision.cpp:208-247 */
        const float lenSq = lensq3(axis);
        if (!(lenSq < kSatAxisEps)) {
            n = axis / sqrtf(lenSq);
            if (dot3(n, delta) < 0.0f) n = -n;
            const float distance = fabsf(dot3(delta, n));
            const float rA = boxA.half.x * abs_dot(n, boxA.ax0) + boxA.half.y * abs_dot(n, boxA.ax1) + boxA.half.z * abs_dot(n, boxA.ax2);
            const float rB = boxB.half.x * abs_dot(n, boxB.ax0) + boxB.half.y * abs_dot(n, boxB.ax1) + boxB.half.z * abs_dot(n, boxB.ax2);
            separation = distance - (rA + rB);
            tested = true;
            separating = separation > kCollisionMargin;
        }
    }
    COOP_STAMP(4);
    if (gballot16(separating, gBase)) return 0;                        /* (the reference returns at the first such axis: same outcome) */

    /* the reference's two running maxima, replayed in its order over the fifteen results */
    int faceK = -1, edgeK = -1; float faceSep = -FLT_MAX, edgeSep = -FLT_MAX;
    const unsigned testedMask = gballot16(tested, gBase);
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const bool tk = (testedMask >> k) & 1u;
        const float sk = gshfl(separation, gBase + k);
        if (k < 6) { if (tk && (faceK < 0 || sk > faceSep)) { faceK = k; faceSep = sk; } }
        else       { if (tk && (edgeK < 0 || sk > edgeSep)) { edgeK = k; edgeSep = sk; } }
    }
    if (faceK < 0) return 0;
    int bestK = faceK;
    if (edgeK >= 0) {
        const float relTol = 0.95f, absTol = 0.01f;                     /* collision.cpp:463-467 */
        if (relTol * edgeSep > faceSep + absTol) bestK = edgeK;
    }
    const f3 nAB = gshfl(n, gBase + bestK);
    COOP_STAMP(5);

    if (bestK >= 6) {
        /* ---- edge-edge: one contact from the closest points of the two support edges (collision.cpp:396-416) ---- */
        const int axA = (bestK - 6) / 3, axB = (bestK - 6) % 3;
        f3 a0, a1, b0, b1;
        support_edge(boxA, axA, nAB, a0, a1);
        support_edge(boxB, axB, -nAB, b0, b1);
        f3 xA, xB;
        closest_on_segments(a0, a1, b0, b1, xA, xB);
        if (gl == 0) {
            RawContact& c = ws.raw[0];
            c.feature = (2 << 24) | ((axA & 0xFF) << 8) | (axB & 0xFF);
            c.rA = qrotate(qconj(A.rot), xA - A.pos);
            c.rB = qrotate(qconj(B.rot), xB - B.pos);
            c.normal = -nAB;
            c.penetration = rmax(0.0f, -dot3(xA - xB, -nAB));
        }
        gsync();
        return 1;
    }

    /* ---- reference-face clipping manifold (collision.cpp:313-394) ---- */
    const bool refIsA = bestK < 3;
    const int refAxis = refIsA ? bestK : bestK - 3;
    /* (selected by value, field by field: a reference to one of two objects would put both into scratch memory) */
    const Obb refBox = obb_select(refIsA, boxA, boxB);
    const Obb incBox = obb_select(refIsA, boxB, boxA);
    const f3 outward = refIsA ? nAB : -nAB;
    const f3 nBA = -nAB;
    const f3 rax = obb_axis(refBox, refAxis);                         /* buildFaceFrame, collision.cpp:93-100 */
    const float fsign = dot3(outward, rax) >= 0.0f ? 1.0f : -1.0f;
    const f3 fN = rax * fsign;
    const f3 fC = refBox.center + fN * half_extent(refBox, refAxis);
    f3 fU, fV; float eU, eV;
    face_axes(refBox, refAxis, fU, fV, eU, eV);
    int incAxis = 0;                                                    /* chooseIncidentFaceAxis, collision.cpp:102-116 */
    {
        float bestD = -FLT_MAX;
        for (int i = 0; i < 3; ++i) {
            const float d = abs_dot(obb_axis(incBox, i), fN);
            if (d > bestD) { bestD = d; incAxis = i; }
        }
    }
    f3* clip0 = ws.clip[0]; f3* clip1 = ws.clip[1];
    if (gl < 4) {                                                       /* buildIncidentFace, collision.cpp:118-134: one corner per lane */
        const f3 iax = obb_axis(incBox, incAxis);
        const float s = dot3(iax, fN) > 0.0f ? -1.0f : 1.0f;
        const f3 nn = iax * s;
        const f3 c = incBox.center + nn * half_extent(incBox, incAxis);
        f3 u, v; float xu, xv;
        face_axes(incBox, incAxis, u, v, xu, xv);
        clip0[gl] = gl == 0 ? c + u * xu + v * xv : (gl == 1 ? c - u * xu + v * xv : (gl == 2 ? c - u * xu - v * xv : c + u * xu - v * xv));
    }
    gsync();
    COOP_STAMP(6);
    int count = 4;
    count = coop_clip(clip0, count, fU, dot3(fU, fC) + eU, clip1, gl, gBase);   if (!count) return 0;
    count = coop_clip(clip1, count, -fU, dot3(-fU, fC) + eU, clip0, gl, gBase); if (!count) return 0;
    count = coop_clip(clip0, count, fV, dot3(fV, fC) + eV, clip1, gl, gBase);   if (!count) return 0;
    count = coop_clip(clip1, count, -fV, dot3(-fV, fC) + eV, clip0, gl, gBase); if (!count) return 0;

    COOP_STAMP(7);
    /* one lane per clipped vertex: projection onto the reference face, feature key */
    bool near_ = false; f3 xA = zero3(), xB = zero3(), mid = zero3(); int key = 0;
    if (gl < count) {
        const f3 pInc = clip0[gl];
        const float distance = dot3(pInc - fC, fN);
        near_ = !(distance > kCollisionMargin);
        const f3 pRef = pInc - fN * distance;
        xA = refIsA ? pRef : pInc;
        xB = refIsA ? pInc : pRef;
        const f3 rel = pRef - fC;
        const float uC = dot3(rel, fU), vC = dot3(rel, fV);
        const float uN = (eU > kSatAxisEps) ? (uC / eU) : 0.0f;
        const float vN = (eV > kSatAxisEps) ? (vC / eV) : 0.0f;
        const int qU = (int)floorf(rclamp((uN + 1.0f) * 7.5f, 0.0f, 15.0f));
        const int qV = (int)floorf(rclamp((vN + 1.0f) * 7.5f, 0.0f, 15.0f));
        key = (((refIsA ? 0 : 1) << 24) | ((refAxis & 0xFF) << 16) | ((incAxis & 0xFF) << 8)) | ((qU & 0x0F) << 4) | (qV & 0x0F);
        mid = (xA + xB) * 0.5f;
    }
    /* addContact's merge-by-distance and four-contact cap, in vertex order (collision.cpp:176-206, 376-392) */
    int cc = 0, mySlot = -1;
    f3 m0 = zero3(), m1 = zero3(), m2 = zero3(), m3 = zero3();
    /* (which candidates are near is one ballot: one round trip through the LDS crossbar per candidate instead of two) */
    const unsigned nearMask = gballot16(near_, gBase);
    for (int i = 0; i < count && cc < kMaxContacts; ++i) {
        if (!((nearMask >> i) & 1u)) continue;
        const f3 mi = gshfl(mid, gBase + i);
        bool merged = false;
        if (cc > 0 && lensq3(mi - m0) < kMergeDistSq) merged = true;
        if (cc > 1 && lensq3(mi - m1) < kMergeDistSq) merged = true;
        if (cc > 2 && lensq3(mi - m2) < kMergeDistSq) merged = true;
        if (merged) continue;
        if (gl == i) mySlot = cc;
        if (cc == 0) m0 = mi; else if (cc == 1) m1 = mi; else if (cc == 2) m2 = mi; else m3 = mi;
        ++cc;
    }
    (void)m3;
    COOP_STAMP(8);
    if (mySlot >= 0) {
        RawContact& c = ws.raw[mySlot];
        c.feature = key;
        c.rA = qrotate(qconj(A.rot), xA - A.pos);
        c.rB = qrotate(qconj(B.rot), xB - B.pos);
        c.normal = nBA;
        c.penetration = rmax(0.0f, -dot3(xA - xB, nBA));
    }
    gsync();
    return cc;
}

/* one contact of the new manifold, held by lane c < numContacts of the group */
struct CoopContact {
    int feature; bool stick;
    f3 rA, rB, normal, u1, u2;
    float penetration, C0n, C0t1, C0t2;
    float lambda[3], penalty[3];
};

/* Manifold::initialize (manifold.cpp:71-175).  The previous step's `oldN` contacts are in ws.old*; on return lane c < numContacts
 * holds contact c.  Returns the new contact count (0: the pair no longer touches, the reference deletes the manifold). */
/* what Manifold::initialize does with the `nNew` contacts collide_boxes_coop has left in ws.raw (nNew > 0; the two halves are separate so that
 * the caller can request the manifold's slot in between: the round trip of that atomic then runs beside this function instead of behind it) */
__device__ __forceinline__ void manifold_adopt_coop(int oldN, int nNew, CoopScratch& ws, const BoxPose& A, const BoxPose& B, CoopContact& out, int gl, int gBase COOP_DBG_PARAM)
{
    /* feature matching: new contact i takes the first old contact with its feature key that no earlier new contact took
     * (the eight keys are read side by side, then matched in registers) */
    int myOld = -1;
    {
        static_assert(kMaxContacts == 4, "four new and four old feature keys");
        int fNew[4], fOld[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { fNew[i] = ws.raw[i].feature; fOld[i] = ws.oldFeature[i]; }
        unsigned used = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= nNew) break;
            int best = -1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= oldN) break;
                if ((used >> j) & 1u) continue;
                if (fNew[i] == fOld[j]) { best = j; break; }
            }
            if (best >= 0) used |= 1u << best;
            if (gl == i) myOld = best;
        }
    }
    COOP_STAMP(10);
    if (gl < nNew) {
        const RawContact& r = ws.raw[gl];
        out.feature = r.feature; out.rA = r.rA; out.rB = r.rB; out.normal = r.normal;
        for (int k = 0; k < 3; ++k) { out.lambda[k] = 0.0f; out.penalty[k] = kPenaltyMin; }
        out.stick = false;
        if (myOld >= 0) {
            const f3 newN = normalize_safe(out.normal, mk3(0.0f, 1.0f, 0.0f));
            const f3 oldNn = normalize_safe(ws.oldNormal[myOld], newN);
            const float nd = dot3(newN, oldNn);
            const f3 oldMid = (world_point(A.pos, A.rot, ws.oldRA[myOld]) + world_point(B.pos, B.rot, ws.oldRB[myOld])) * 0.5f;
            const f3 newMid = (world_point(A.pos, A.rot, out.rA) + world_point(B.pos, B.rot, out.rB)) * 0.5f;
            const float driftSq = lensq3(newMid - oldMid);
            const bool valid = (nd >= kWarmMinDot) && (driftSq <= kWarmMaxDrift * kWarmMaxDrift);
            if (valid) {
                for (int k = 0; k < 3; ++k) {
                    out.lambda[k] = ws.oldLambda[myOld * 3 + k];
                    out.penalty[k] = rclamp(ws.oldPenalty[myOld * 3 + k], kPenaltyMin, kManifoldPenaltyCap);
                }
            }
            bool reuse = false;
            const bool oldStick = ws.oldStick[myOld] != 0;
            if (oldStick && valid) reuse = (nd >= kStickMinDot) && (driftSq <= kStickMaxDrift * kStickMaxDrift);
            out.stick = oldStick && reuse;
            if (reuse) { out.rA = ws.oldRA[myOld]; out.rB = ws.oldRB[myOld]; }
        }
        f3 nn, t1, t2;
        contact_basis(out.normal, nn, t1, t2);
        out.normal = nn; out.u1 = t1; out.u2 = t2;
        const f3 d = world_point(A.pos, A.rot, out.rA) - world_point(B.pos, B.rot, out.rB);
        out.C0n = dot3(d, nn) - kNormalMargin;
        out.C0t1 = dot3(d, t1);
        out.C0t2 = dot3(d, t2);
        out.penetration = rmax(0.0f, -dot3(d, nn));
    }
}

/* Manifold::initialize in one piece */
__device__ __forceinline__ int manifold_initialize_coop(int oldN, CoopScratch& ws, const BoxPose& A, const BoxPose& B, CoopContact& out, int gl, int gBase COOP_DBG_PARAM)
{
    const int nNew = collide_boxes_coop(A, B, ws, gl, gBase COOP_DBG_ARG);
    COOP_STAMP(9);
    if (nNew == 0) return 0;
    manifold_adopt_coop(oldN, nNew, ws, A, B, out, gl, gBase COOP_DBG_ARG);
    return nNew;
}

} // namespace avbd
