/*
 * avbd_kernels.hip — gfx950 kernels for the phases of Solver::step() (solver.cpp:255-514).
 *
 *   K1  k_broadphase        solver.cpp:262-270   O(n^2) sphere test, LDS-tiled, + isConstrainedTo (rigid.cpp:61-69)
 *   K2  k_narrowphase       manifold.cpp:71-175 + collision.cpp:420-489 + warm-start decay solver.cpp:281-293
 *       k_gforce_prepare    same decay for Joint/Spring rows
 *       k_scan / k_fill_adj / k_sort_expand / k_levels   — replace the intrusive lists of force.cpp
 *   K4  k_predict           solver.cpp:299-337
 *   K5  k_primal            solver.cpp:344-409   one wavefront per body, one Gauss-Seidel level per launch
 *   K6  k_dual / k_dual_gforce   solver.cpp:411-430
 *   K7/K8 k_tail            solver.cpp:434-497 (velocities + contact diagnostics)
 *
 * Exactness: the primal sweep of the reference is sequential over bodies (newest first) and each
 * body accumulates its rows force by force, row by row.  Here bodies are grouped into dependency
 * levels (a body's level is 1 + the max level of its later-created dynamic neighbours), launched
 * level by level, and inside a body the per-row contributions are computed by parallel lanes but
 * ADDED in the reference's order by one lane per accumulator entry.  Results are bit-identical
 * to the serial sweep.  No MFMA: there is no dense contraction on this path.
 */
#include "avbd_types.h"
#include "avbd_rows.h"
#include "avbd_util.h"
#include "avbd_collide_coop.h"

#pragma clang fp contract(off)

namespace avbd {

constexpr int kResLiteMaxEntriesDecl = 33;   /* = kResLiteMaxEntries (avbd_resident.h), needed before that header is included */


/* Rigid::isConstrainedTo (rigid.cpp:61-69) on the CSR lists: is `other` among adjOther[s .. e)?  Eight entries per trip, loaded side by side
 * (an entry-by-entry loop with an early exit is a chain of dependent memory round trips — six of them for a cube inside a pile — in front of
 * every sphere-overlapping pair of the broadphase). */
__device__ __forceinline__ bool adj_lists(const int* __restrict__ adjOther, int s, int e, int other)
{
    for (int k = s; k < e; k += 8) {
        int v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = adjOther[min(k + q, e - 1)];
        bool hit = false;
#pragma unroll
        for (int q = 0; q < 8; ++q) hit |= v[q] == other;
        if (hit) return true;
    }
    return false;
}

/* ------------------------------------------------------------------------------------------
 * K1 broadphase.  Grid = T(T+1)/2 tiles of 256x256 (T = ceil(n/256)); tile (ti,tj), tj<=ti, tests
 * bodies A in tile ti against bodies B in tile tj with B < A (A is the later-created body: it
 * precedes B in the reference's list).  16 B per body staged in LDS.
 * ------------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_broadphase(BodyTable B, const int* __restrict__ adjStart, const int* __restrict__ adjOther,
                                                     int2* cand, int candCap, Counters* cnt, int keepUnsolvedPairs)
{
    __shared__ float4 sB[256];
    const float4* posr = B.plane(BP_POSR);
    const int n = B.n;
    const int bid = blockIdx.x;
    int ti = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    while (ti * (ti + 1) / 2 > bid) --ti;
    const int tj = bid - ti * (ti + 1) / 2;
    const int i = ti * 256 + threadIdx.x;
    const int j0 = tj * 256;
    {
        int j = j0 + threadIdx.x;
        sB[threadIdx.x] = (j < n) ? posr[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (i >= n) return;
    const float4 a = posr[i];
    const f3 pa = mk3(a.x, a.y, a.z);
    const bool solvedI = B.solved[i] != 0;
    int jEnd = n - j0; if (jEnd > 256) jEnd = 256;
    if (i - j0 < jEnd) jEnd = i - j0; /* only j < i */
    for (int jj = 0; jj < jEnd; ++jj) {
        const float4 b = sB[jj];
        f3 dp = pa - mk3(b.x, b.y, b.z);
        float r = a.w + b.w;
        if (dot3(dp, dp) <= r * r) {
            const int j = j0 + jj;
            /* sharded runs: a pair belongs to this GPU when it solves one of its bodies (pairs nobody solves, e.g.
             * static-static, are kept by rank 0 only so that diagnostics count them once) */
            if (!solvedI && !B.solved[j]) {
                const bool bothStatic = !(B.plane(BP_MASS)[i].y > 0.0f) && !(B.plane(BP_MASS)[j].y > 0.0f);
                if (!(keepUnsolvedPairs && bothStatic)) continue;
            }
            /* Rigid::isConstrainedTo: does any live force connect i and j?  scan the shorter list */
            int si = adjStart[i], ei = adjStart[i + 1], sj = adjStart[j], ej = adjStart[j + 1];
            int s, e, other;
            if (ei - si <= ej - sj) { s = si; e = ei; other = j; } else { s = sj; e = ej; other = i; }
            if (!adj_lists(adjOther, s, e, other)) {
                int slot = atomicAdd(&cnt->numCandidates, 1);
                if (slot < candCap) cand[slot] = make_int2(i, j);
                else atomicOr(&cnt->overflow, 1);
            }
        }
    }
}

/* The same pair enumeration for small scenes (a few thousand bodies): one thread per (body A, 32-body slice of the
 * earlier-created bodies), no LDS staging — the whole position table sits in L2 — so that a 1000-body scene still
 * spreads over the chip instead of ten workgroups. */
__global__ __launch_bounds__(256) void k_broadphase_small(BodyTable B, const int* __restrict__ adjStart, const int* __restrict__ adjOther,
                                                           int2* cand, int candCap, Counters* cnt, int keepUnsolvedPairs, int slices)
{
    const float4* posr = B.plane(BP_POSR);
    const int n = B.n;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / slices, sl = t - i * slices;
    if (i >= n) return;
    const int jBeg = sl * 32;
    int jEnd = jBeg + 32; if (jEnd > i) jEnd = i;
    if (jBeg >= jEnd) return;
    const float4 a = posr[i];
    const f3 pa = mk3(a.x, a.y, a.z);
    const bool solvedI = B.solved[i] != 0;
    for (int j = jBeg; j < jEnd; ++j) {
        const float4 b = posr[j];
        f3 dp = pa - mk3(b.x, b.y, b.z);
        float r = a.w + b.w;
        if (dot3(dp, dp) <= r * r) {
            if (!solvedI && !B.solved[j]) {
                const bool bothStatic = !(B.plane(BP_MASS)[i].y > 0.0f) && !(B.plane(BP_MASS)[j].y > 0.0f);
                if (!(keepUnsolvedPairs && bothStatic)) continue;
            }
            int si = adjStart[i], ei = adjStart[i + 1], sj = adjStart[j], ej = adjStart[j + 1];
            int s, e, other;
            if (ei - si <= ej - sj) { s = si; e = ei; other = j; } else { s = sj; e = ej; other = i; }
            if (!adj_lists(adjOther, s, e, other)) {
                int slot = atomicAdd(&cnt->numCandidates, 1);
                if (slot < candCap) cand[slot] = make_int2(i, j);
                else atomicOr(&cnt->overflow, 1);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * K1g grid broadphase, scenes above 4096 bodies: the SAME pair set as the O(n^2) loop — every pair
 * (A later-created, B earlier) whose bounding spheres overlap and that shares no force — found through a hashed uniform
 * grid.  The creation keys make the order in which candidates are appended irrelevant.  Bodies whose radius exceeds the
 * cell (the ground slab) are on a short "big" list and are tested against everybody; the others are binned by centre into
 * cells of edge >= 2 * largest small radius, so two overlapping small spheres are never more than one cell apart on any axis.
 * Different cells may share a bucket of the hash table (that only adds pairs to the exact sphere test); a body scans each
 * distinct bucket once.
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ int grid_coord(float x, float invCell) { return (int)floorf(x * invCell); }
__device__ __forceinline__ unsigned grid_bucket(int ix, int iy, int iz, unsigned mask)
{
    /* lattice coordinates are small consecutive integers: mix them properly (murmur3 finaliser) before masking */
    unsigned h = (unsigned)ix * 73856093u + (unsigned)iy * 19349663u + (unsigned)iz * 83492791u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h & mask;
}

/* every small body pushes itself on the linked list of its bucket (heads start at -1; order inside a bucket is irrelevant) */
__global__ void k_grid_build(BodyTable B, const unsigned char* __restrict__ isBig, float invCell, unsigned mask, int* __restrict__ cellHead, int* __restrict__ cellNext)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.n) return;
    if (isBig[i]) { cellNext[i] = -1; return; }
    const float4 p = B.plane(BP_POSR)[i];
    const unsigned h = grid_bucket(grid_coord(p.x, invCell), grid_coord(p.y, invCell), grid_coord(p.z, invCell), mask);
    cellNext[i] = atomicExch(&cellHead[h], i);
}

/* the pair test of solver.cpp:262-270 + Rigid::isConstrainedTo (rigid.cpp:61-69) for one ordered pair i > j */
template <bool SPHERES_TESTED = false>
__device__ __forceinline__ bool broadphase_pair_passes(const BodyTable& B, int i, int j, const int* __restrict__ adjStart, const int* __restrict__ adjOther,
                                                       int keepUnsolvedPairs)
{
    if (!SPHERES_TESTED) {
        const float4* posr = B.plane(BP_POSR);
        const float4 a = posr[i], b = posr[j];
        const f3 dp = mk3(a.x, a.y, a.z) - mk3(b.x, b.y, b.z);
        const float r = a.w + b.w;
        if (!(dot3(dp, dp) <= r * r)) return false;
    }
    if (!B.solved[i] && !B.solved[j]) {
        const bool bothStatic = !(B.plane(BP_MASS)[i].y > 0.0f) && !(B.plane(BP_MASS)[j].y > 0.0f);
        if (!(keepUnsolvedPairs && bothStatic)) return false;
    }
    const int si = adjStart[i], ei = adjStart[i + 1], sj = adjStart[j], ej = adjStart[j + 1];
    int s, e, other;
    if (ei - si <= ej - sj) { s = si; e = ei; other = j; } else { s = sj; e = ej; other = i; }
    return !adj_lists(adjOther, s, e, other);
}

constexpr int kGridCellsPerBody = 14;
/* one thread per (body, one of 14 cells around it): the centre thread also takes the body's pairs with the big bodies.
 * Appending is what bounds a broadphase here (a falling 64 000-cube grid yields 250 000 candidates per step, and atomics on
 * one word retire at ~10 ns each): every thread first COUNTS its candidates — keeping the first three in registers — the workgroup
 * reserves its range with ONE atomic, then the threads write (a thread with more than three walks its list a second time).
 * Different cells may share a bucket of the hash table: a body of the bucket counts for the thread whose cell it really lies in
 * (its centre is in the thread's hand for the sphere test anyway), so no thread has to know which buckets its 26 siblings scan. */
__global__ __launch_bounds__(256) void k_grid_pairs(BodyTable B, const unsigned char* __restrict__ isBig, const int* __restrict__ bigList, int nBig, float invCell, unsigned mask,
                             const int* __restrict__ cellHead, const int* __restrict__ cellNext, const int* __restrict__ adjStart,
                             const int* __restrict__ adjOther, int2* cand, int candCap, Counters* cnt, int keepUnsolvedPairs)
{
    __shared__ int sCount, sBase;
    /* HALF the neighbourhood: the body's own cell (pairs with earlier-created bodies) and the 13 cells behind it in (z, y, x) order (pairs with every body
     * there, whichever is the later-created one) — an unordered pair of bodies in two neighbouring cells is met exactly once, from the cell that comes first */
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(t / kGridCellsPerBody), k = (int)(t - (long long)i * kGridCellsPerBody) + 13;
    const bool live = i < B.n;
    const bool big = live && isBig[i];
    const bool own = k == 13;
    if (threadIdx.x == 0) sCount = 0;
    int head = -1, tx = 0, ty = 0, tz = 0;
    float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && !big) {
        pi = B.plane(BP_POSR)[i];
        tx = grid_coord(pi.x, invCell) + (k % 3) - 1; ty = grid_coord(pi.y, invCell) + ((k / 3) % 3) - 1; tz = grid_coord(pi.z, invCell) + (k / 9) - 1;
        head = cellHead[grid_bucket(tx, ty, tz, mask)];
    }
    __syncthreads();
    /* the pair test for a body j of the thread's bucket: in the thread's cell, earlier-created, spheres overlap, no force between them */
    auto cell_pair_passes = [&](int j) -> bool {
        if (own ? !(j < i) : j == i) return false;
        const float4 b = B.plane(BP_POSR)[j];
        const f3 dp = mk3(pi.x, pi.y, pi.z) - mk3(b.x, b.y, b.z);
        const float r = pi.w + b.w;
        if (!(dot3(dp, dp) <= r * r)) return false;
        if (grid_coord(b.x, invCell) != tx || grid_coord(b.y, invCell) != ty || grid_coord(b.z, invCell) != tz) return false;   /* another cell of this bucket */
        return broadphase_pair_passes<true>(B, i > j ? i : j, i > j ? j : i, adjStart, adjOther, keepUnsolvedPairs);
    };
    int mine = 0;
    int2 h0 = make_int2(0, 0), h1 = h0, h2 = h0;
    auto note = [&](int hi, int lo) { if (mine == 0) h0 = make_int2(hi, lo); else if (mine == 1) h1 = make_int2(hi, lo); else if (mine == 2) h2 = make_int2(hi, lo); ++mine; };
    if (live && k == 13) {                                            /* pairs with a big body: once per unordered pair */
        for (int g = 0; g < nBig; ++g) {
            const int o = bigList[g];
            if (o == i || (big && !(i < o))) continue;
            const int hi = i > o ? i : o, lo = i > o ? o : i;
            if (broadphase_pair_passes(B, hi, lo, adjStart, adjOther, keepUnsolvedPairs)) note(hi, lo);
        }
    }
    for (int j = head; j >= 0; j = cellNext[j])
        if (cell_pair_passes(j)) note(i > j ? i : j, i > j ? j : i);
    int offset = mine ? atomicAdd(&sCount, mine) : 0;                  /* LDS: position inside the workgroup's range */
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = sCount;
        const int base = total ? atomicAdd(&cnt->numCandidates, total) : 0;
        if (base + total > candCap) atomicOr(&cnt->overflow, 1);
        sBase = base;
    }
    __syncthreads();
    offset += sBase;
    if (mine == 0) return;
    if (mine <= 3) {
        if (offset < candCap) cand[offset] = h0;
        if (mine > 1 && offset + 1 < candCap) cand[offset + 1] = h1;
        if (mine > 2 && offset + 2 < candCap) cand[offset + 2] = h2;
        return;
    }
    int w = 0;                                                         /* (rare) more than three: the same tests again, in the same order */
    auto put = [&](int hi, int lo) { const int slot = offset + w; if (slot < candCap) cand[slot] = make_int2(hi, lo); ++w; };
    if (k == 13) {
        for (int g = 0; g < nBig; ++g) {
            const int o = bigList[g];
            if (o == i || (big && !(i < o))) continue;
            const int hi = i > o ? i : o, lo = i > o ? o : i;
            if (broadphase_pair_passes(B, hi, lo, adjStart, adjOther, keepUnsolvedPairs)) put(hi, lo);
        }
    }
    for (int j = head; j >= 0; j = cellNext[j])
        if (cell_pair_passes(j)) put(i > j ? i : j, i > j ? j : i);
}

/* ------------------------------------------------------------------------------------------
 * K2 narrow phase + Manifold::initialize + warm-start decay.  One lane per pair: last step's
 * manifolds (src) followed by this step's candidates.  Survivors are appended to dst.
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ BoxPose load_box(const BodyTable& B, int i, float& friction, float& invMass)
{
    BoxPose p;
    p.pos = ld3(B.plane(BP_POSR), i);
    p.rot = ldq(B.plane(BP_QUAT), i);
    float4 s = B.plane(BP_SIZE)[i];
    p.size = mk3(s.x, s.y, s.z);
    friction = s.w;
    invMass = B.plane(BP_MASS)[i].y;
    return p;
}

/* One pair by the sixteen lanes of a group (avbd_collide_coop.h): last step's manifold `item` of `src` (item >= 0) or a fresh
 * candidate (a, b); survivors are appended to dst, one lane per contact.  item / a / b are uniform over the group. */
__device__ __forceinline__ void narrow_pair(const BodyTable& B, const ManifoldTable& src, const ManifoldTable& dst, int item, int a, int b, int step,
                                            CoopScratch& ws, Counters* cnt, int* nextCount, int* degForce, int* degInc, const SolverParams& P, int gl, int gBase,
                                            unsigned long long* dbg = nullptr)
{
    uint64_t key;
    int oldN = 0;
    if (item >= 0) {
        const int2 pr = src.pair[item]; a = pr.x; b = pr.y; key = src.key[item];
        oldN = src.nc[item];
        if (gl < oldN) {                                              /* one lane per old contact */
            const int s = item * 4 + gl;
            const float4 ra = src.cRA[s], rb = src.cRB[s], nn = src.cN[s], lam = src.cLam[s], pen = src.cPen[s];
            const int2 fs = src.cFS[s];
            ws.oldFeature[gl] = fs.x; ws.oldStick[gl] = fs.y != 0 ? 1 : 0;
            ws.oldRA[gl] = mk3(ra.x, ra.y, ra.z); ws.oldRB[gl] = mk3(rb.x, rb.y, rb.z); ws.oldNormal[gl] = mk3(nn.x, nn.y, nn.z);
            ws.oldLambda[gl * 3 + 0] = lam.x; ws.oldLambda[gl * 3 + 1] = lam.y; ws.oldLambda[gl * 3 + 2] = lam.z;
            ws.oldPenalty[gl * 3 + 0] = pen.x; ws.oldPenalty[gl * 3 + 1] = pen.y; ws.oldPenalty[gl * 3 + 2] = pen.z;
        }
    } else {
        key = make_manifold_key(step, a, b);
    }
    gsync();
    float fa, fb, ima, imb;
    const BoxPose A = load_box(B, a, fa, ima);
    const BoxPose Bp = load_box(B, b, fb, imb);
#ifdef AVBD_DF_STATS
    if (dbg) dbg[1] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(A.pos.x == 12345.0f);
#endif
    const bool solvedA = B.solved[a] != 0, solvedB = B.solved[b] != 0;   /* (for the degree counts at the end: requested with the boxes) */
    CoopContact cc;
    const int nc = collide_boxes_coop(A, Bp, ws, gl, gBase
#ifdef AVBD_DF_STATS
                                      , dbg
#endif
                                      );
#ifdef AVBD_DF_STATS
    if (dbg) { dbg[9] = __builtin_amdgcn_s_memrealtime(); if (!nc) dbg[3] = dbg[2] = dbg[9]; }
#endif
    if (nc == 0) return;
    /* the manifold survives (manifold.cpp:112-117): its slot is requested now, and the round trip of that atomic runs beside the rest of initialize */
    int slot = 0;
    if (gl == 0) slot = atomicAdd(nextCount, 1);
    manifold_adopt_coop(oldN, nc, ws, A, Bp, cc, gl, gBase
#ifdef AVBD_DF_STATS
                        , dbg
#endif
                        );
#ifdef AVBD_DF_STATS
    if (dbg) dbg[2] = __builtin_amdgcn_s_memrealtime();
#endif
    const float combinedFriction = sqrtf(fa * fb);                     /* manifold.cpp:73 */

    /* warm-start decay, solver.cpp:281-293 (manifold rows have stiffness FLT_MAX: no cap) */
    if (gl < nc) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!P.postStabilize) cc.lambda[k] *= P.alpha * P.gamma;
            cc.penalty[k] = rclamp(cc.penalty[k] * P.gamma, kPenaltyMin, kPenaltyMax);
        }
    }
    slot = gshfl(slot, gBase);
    if (slot >= dst.cap) { if (gl == 0) atomicOr(&cnt->overflow, 2); return; }
    if (gl == 0) {
        dst.pair[slot] = make_int2(a, b);
        dst.key[slot] = key;
        dst.nc[slot] = nc;
        dst.fric[slot] = combinedFriction;
        atomicAdd(&degForce[a], 1);
        atomicAdd(&degForce[b], 1);
        int inc = 0;
        if (solvedA) { atomicAdd(&degInc[a], nc); inc += nc; }
        if (solvedB) { atomicAdd(&degInc[b], nc); inc += nc; }
        atomicAdd(reinterpret_cast<unsigned long long*>(&cnt->numContacts), ((unsigned long long)(unsigned)inc << 32) | (unsigned long long)(unsigned)nc);
    }
    if (gl < nc) {
        const int s = slot * 4 + gl;
        dst.cRA[s] = pack3(cc.rA, cc.C0n);
        dst.cRB[s] = pack3(cc.rB, cc.C0t1);
        dst.cN[s] = pack3(cc.normal, cc.C0t2);
        dst.cU1[s] = pack3(cc.u1, 0.f);
        dst.cU2[s] = pack3(cc.u2, 0.f);
        dst.cLam[s] = make_float4(cc.lambda[0], cc.lambda[1], cc.lambda[2], cc.penetration);
        dst.cPen[s] = make_float4(cc.penalty[0], cc.penalty[1], cc.penalty[2], 0.0f);
        dst.cFS[s] = make_int2(cc.feature, cc.stick ? 1 : 0);
        dst.cC[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        dst.cFmin[s] = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, 0.f);
        dst.cFmax[s] = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
    }
#ifdef AVBD_DF_STATS
    if (dbg) dbg[3] = __builtin_amdgcn_s_memrealtime();
#endif
}

constexpr int kPairsPerWave = 4;       /* sixteen lanes per pair */

/* One pair per LANE (the throughput form, for scenes beyond 4096 bodies: tens of thousands of pairs fill the machine, and a pair
 * costs a third of the wave time it costs when sixteen lanes share it): last step's manifold `item` of `src` (item >= 0) or a fresh candidate (a, b); survivors are appended to dst */
__device__ __forceinline__ void narrow_pair_lane(const BodyTable& B, const ManifoldTable& src, const ManifoldTable& dst, int item, int a, int b, int step,
                                            ManifoldInit& mi, ManifoldOld& old, Counters* cnt, int* nextCount, int* degForce, int* degInc, const SolverParams& P,
                                            const NarrowOrder& O, unsigned long long* dbg = nullptr)
{
    uint64_t key;
    if (item >= 0) {
        int2 pr = src.pair[item]; a = pr.x; b = pr.y; key = src.key[item];
        mi.numContacts = src.nc[item];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < mi.numContacts) {
                int s = item * 4 + c;
                float4 ra = src.cRA[s], rb = src.cRB[s], nn = src.cN[s], lam = src.cLam[s], pen = src.cPen[s];
                int2 fs = src.cFS[s];
                mi.feature[c] = fs.x; mi.stick[c] = fs.y != 0;
                mi.rA[c] = mk3(ra.x, ra.y, ra.z); mi.rB[c] = mk3(rb.x, rb.y, rb.z); mi.normal[c] = mk3(nn.x, nn.y, nn.z);
                mi.lambda[c * 3 + 0] = lam.x; mi.lambda[c * 3 + 1] = lam.y; mi.lambda[c * 3 + 2] = lam.z;
                mi.penalty[c * 3 + 0] = pen.x; mi.penalty[c * 3 + 1] = pen.y; mi.penalty[c * 3 + 2] = pen.z;
            }
        }
    } else {
        key = make_manifold_key(step, a, b);
        mi.numContacts = 0;
    }
    float fa, fb, ima, imb;
    BoxPose A = load_box(B, a, fa, ima);
    BoxPose Bp = load_box(B, b, fb, imb);
#ifdef AVBD_DF_STATS
    if (dbg) dbg[1] = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(A.pos.x == 12345.0f);
#endif
    const bool alive = manifold_initialize(mi, old, A, fa, Bp, fb);
    if (old.ws.overflow) atomicOr(&cnt->overflow, 64);                 /* a clipped polygon with more than eight vertices (clip_polygon) */
#ifdef AVBD_DF_STATS
    if (dbg) { dbg[2] = __builtin_amdgcn_s_memrealtime(); if (!alive) dbg[3] = dbg[2]; }
#endif
    if (!alive) return;

    /* warm-start decay, solver.cpp:281-293 (manifold rows have stiffness FLT_MAX: no cap) */
    const int rows = mi.numContacts * 3;
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        if (r < rows) {
            if (!P.postStabilize) mi.lambda[r] *= P.alpha * P.gamma;
            mi.penalty[r] = rclamp(mi.penalty[r] * P.gamma, kPenaltyMin, kPenaltyMax);
        }
    }

    int slot = atomicAdd(nextCount, 1);
    if (slot >= dst.cap) { atomicOr(&cnt->overflow, 2); return; }
    if (O.permNext) {                                                 /* next step's processing order: this survivor's slot on the list of its class */
        const int cls = mi.numContacts >= 3 ? 0 : (mi.numContacts == 2 ? 1 : ((mi.feature[0] >> 24) == 2 ? 3 : 2));
#pragma unroll
        for (int k = 0; k < kNarrowClasses; ++k) {
            const unsigned long long m = __ballot(cls == k);          /* (of the lanes that got here) */
            if (cls == k) {
                const int leader = __ffsll((long long)m) - 1, lane = (int)(threadIdx.x & 63);
                int base = 0;
                if (lane == leader) base = atomicAdd(&O.cntNext[k * kNarrowCntStride], __popcll(m));
                base = __shfl(base, leader);
                O.permNext[(size_t)k * O.cap + base + __popcll(m & ((1ull << lane) - 1ull))] = slot;
            }
        }
    }
    dst.pair[slot] = make_int2(a, b);
    dst.key[slot] = key;
    dst.nc[slot] = mi.numContacts;
    dst.fric[slot] = mi.combinedFriction;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < mi.numContacts) {
            int s = slot * 4 + c;
            dst.cRA[s] = pack3(mi.rA[c], mi.C0n[c]);
            dst.cRB[s] = pack3(mi.rB[c], mi.C0t1[c]);
            dst.cN[s] = pack3(mi.normal[c], mi.C0t2[c]);
            dst.cU1[s] = pack3(mi.u1[c], 0.f);
            dst.cU2[s] = pack3(mi.u2[c], 0.f);
            dst.cLam[s] = make_float4(mi.lambda[c * 3], mi.lambda[c * 3 + 1], mi.lambda[c * 3 + 2], mi.penetration[c]);
            dst.cPen[s] = make_float4(mi.penalty[c * 3], mi.penalty[c * 3 + 1], mi.penalty[c * 3 + 2], 0.0f);
            dst.cFS[s] = make_int2(mi.feature[c], mi.stick[c] ? 1 : 0);
            dst.cC[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            dst.cFmin[s] = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, 0.f);
            dst.cFmax[s] = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
        }
    }
    atomicAdd(&degForce[a], 1);
    atomicAdd(&degForce[b], 1);
    int inc = 0;
    if (B.solved[a]) { atomicAdd(&degInc[a], mi.numContacts); inc += mi.numContacts; }
    if (B.solved[b]) { atomicAdd(&degInc[b], mi.numContacts); inc += mi.numContacts; }
    /* numContacts and numIncidences are neighbours (an aligned 8 bytes): ONE atomic for both — atomics on one cache line retire one after the other,
     * about 13 ns each whoever issues them, and a wave's worth of survivors brings one per word (5 300 waves at 64 000 cubes: 69 us per word) */
    static_assert(offsetof(Counters, numContacts) % 8 == 0 && offsetof(Counters, numIncidences) == offsetof(Counters, numContacts) + 4, "one 64-bit add for both counts");
    atomicAdd(reinterpret_cast<unsigned long long*>(&cnt->numContacts), ((unsigned long long)(unsigned)inc << 32) | (unsigned long long)(unsigned)mi.numContacts);
#ifdef AVBD_DF_STATS
    if (dbg) dbg[3] = __builtin_amdgcn_s_memrealtime();
#endif
}

/* (one wave per SIMD: capped at 256 registers for two, the kernel spills 23 and runs at the same speed — 240 us at 64 000 cubes either way: it is
 * bound by instruction issue, not by latency; sixteen lanes per pair, the small scenes' form, is 2.4x slower here) */
/* LANES = 32: with a few thousand pairs and 1024 SIMDs a half-filled wavefront costs nothing and halves the branch divergence.  LANES = 64 (round 4) once
 * the pairs outnumber the lanes of the machine several times: the per-pair LDS (584 B since the clip buffers hold the eight vertices a clipped quad can have and
 * last step's lambda / penalty stay in registers; 872 B before) lets four 64-lane workgroups share a CU — one wave per SIMD, full waves. */
template <int LANES>
__global__ __launch_bounds__(LANES) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_narrowphase(BodyTable B, ManifoldTable src, ManifoldTable dst, const int2* __restrict__ cand,
                                                      int candCap, Counters* cnt, int* nextCount, int* degForce, int* degInc, SolverParams P, int step, NarrowOrder O)
{
    const int nOld = cnt->numManifolds;
    int nCand = cnt->numCandidates; if (nCand > candCap) nCand = candCap;
    const int total = nOld + nCand;
    /* last step's manifolds class by class (NarrowOrder): the lists must hold exactly the manifolds of src */
    int c0 = 0, c1 = 0, c2 = 0;
    bool ordered = O.permCur != nullptr;
    if (ordered) {
        c0 = O.cntCur[0]; c1 = c0 + O.cntCur[kNarrowCntStride]; c2 = c1 + O.cntCur[2 * kNarrowCntStride];
        ordered = c2 + O.cntCur[3 * kNarrowCntStride] == nOld;
    }
    /* the dynamically indexed per-pair state lives in LDS instead of scratch */
    static_assert(sizeof(ManifoldOld) * 64 * 4 <= 160 * 1024, "four 64-lane workgroups of k_narrowphase share a CU's LDS");
    __shared__ ManifoldOld sOld[LANES];
    ManifoldInit miPriv;
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += gridDim.x * blockDim.x) {
        int a = 0, b = 0, slot = -1;
        if (item >= nOld) { int2 pr = cand[item - nOld]; a = pr.x; b = pr.y; }
        else if (ordered) {
            const int k = item >= c2 ? 3 : (item >= c1 ? 2 : (item >= c0 ? 1 : 0));
            slot = O.permCur[(size_t)k * O.cap + (item - (k == 3 ? c2 : (k == 2 ? c1 : (k == 1 ? c0 : 0))))];
        } else slot = item;
        narrow_pair_lane(B, src, dst, slot, a, b, step, miPriv, sOld[threadIdx.x], cnt, nextCount, degForce, degInc, P, O);
    }
}

/* Small scenes: broadphase and narrow phase as ONE launch.  The first `persistBlocks` workgroups re-collide last step's
 * manifolds (they do not depend on the broadphase); every other workgroup enumerates 32 (body, 32-body slice) pair sets,
 * keeps its sphere-overlapping, unconstrained pairs in LDS and then collides them itself, one pair per lane.  Creation
 * keys make the order in which manifolds are appended irrelevant (avbd_types.h: make_manifold_key). */
/* Two shapes of the launch (THREADS per workgroup, SLICE bodies per broadphase thread): 128 / 8 up to kFrontWideFrom bodies, 256 / 16 beyond (a quarter of the
 * workgroups: the step of 1 331 / 1 728 / 2 744 / 3 375 cubes 0.708 -> 0.703 / 0.691 -> 0.687 / 0.871 -> 0.852 / 1.015 -> 0.994 ms; 1 000 and 512 cubes: the same either way, their
 * tail is the workgroups that re-collide last step's manifolds).  Option front_wide_from. */
constexpr int kFrontWideFrom = 1100;
constexpr int kFrontSlice = 8;                     /* bodies one thread of the fused broadphase tests: short slices = few candidates per workgroup, and a workgroup collides its own */
constexpr int kFrontThreads = 128;                  /* (2 650 one-wave workgroups took 27 us to dispatch at Stress1000: the pairs of block 0 were done after 8) */
constexpr int kFrontPairs = kPairsPerWave * (kFrontThreads / 64);   /* pairs a workgroup collides side by side */
constexpr int kFrontSliceWide = 16, kFrontThreadsWide = 256, kFrontPairsWide = kPairsPerWave * (kFrontThreadsWide / 64);
#ifdef AVBD_DF_STATS
constexpr int kFrontStampBlocks = 8192;
__device__ unsigned long long gFrontBlockStamps[16 * kFrontStampBlocks];   /* first / last instant of every workgroup of the last k_front_small (avbd_debug_front_blocks) */
#endif
template <int THREADS, int SLICE>
__global__ __launch_bounds__(THREADS) void k_front_small(BodyTable B, ManifoldTable src, ManifoldTable dst,
                                                    const int* __restrict__ adjStart, const int* __restrict__ adjOther, Counters* cnt, int* nextCount,
                                                    int* degForce, int* degInc, SolverParams P, int step, int persistBlocks, int keepUnsolvedPairs, int slices)
{
    constexpr int PAIRS = kPairsPerWave * (THREADS / 64);
    __shared__ CoopScratch sWs[PAIRS];
    __shared__ int2 sHits[THREADS * SLICE];                           /* a workgroup can never find more */
    __shared__ int sNumHits;
    const int gl = threadIdx.x & 15, gBase = threadIdx.x & 48, gj = threadIdx.x >> 4;   /* gBase: first lane of the group inside its wave */
#ifdef AVBD_DF_STATS
    if (threadIdx.x == 0 && blockIdx.x < kFrontStampBlocks) { for (int k = 1; k < 16; ++k) gFrontBlockStamps[16 * blockIdx.x + k] = 0; gFrontBlockStamps[16 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if ((int)blockIdx.x < persistBlocks) {
#ifdef AVBD_DF_STATS
        unsigned long long* dbg = (blockIdx.x == 0 && threadIdx.x == 0) ? reinterpret_cast<unsigned long long*>(dst.cC + (size_t)dst.cap * 4 - 8) : nullptr;
        if (dbg) dbg[0] = __builtin_amdgcn_s_memrealtime();
        if (!dbg && (threadIdx.x & 15) == 0 && blockIdx.x < kFrontStampBlocks && (threadIdx.x >> 4) < 1) dbg = gFrontBlockStamps + 16 * blockIdx.x;   /* [1..3]: after loads / initialize / end of group 0's first pair, [4..10]: inside initialize */
#endif
        const int nOld = cnt->numManifolds;
        /* four pairs per wavefront, sixteen lanes each (avbd_collide_coop.h) */
        for (int item = blockIdx.x * PAIRS + gj; item < nOld; item += persistBlocks * PAIRS)
            narrow_pair(B, src, dst, item, 0, 0, step, sWs[gj], cnt, nextCount, degForce, degInc, P, gl, gBase
#ifdef AVBD_DF_STATS
                        , dbg
#endif
                        );
#ifdef AVBD_DF_STATS
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x < kFrontStampBlocks) gFrontBlockStamps[16 * blockIdx.x + 15] = __builtin_amdgcn_s_memrealtime();
#endif
        return;
    }
    if (threadIdx.x == 0) sNumHits = 0;
    __syncthreads();
    const float4* posr = B.plane(BP_POSR);
    const int n = B.n;
    const int t = ((int)blockIdx.x - persistBlocks) * THREADS + threadIdx.x;
    const int i = t / slices, sl = t - i * slices;
    const int jBeg = sl * SLICE;
    int jEnd = jBeg + SLICE; if (jEnd > i) jEnd = i;
    if (i < n && jBeg < jEnd) {
        const float4 a = posr[i];
        const f3 pa = mk3(a.x, a.y, a.z);
        const bool solvedI = B.solved[i] != 0;
        float4 bs[SLICE];                                       /* the slice's spheres, requested side by side */
#pragma unroll
        for (int q = 0; q < SLICE; ++q) bs[q] = posr[min(jBeg + q, jEnd - 1)];
        const int si = adjStart[i], ei = adjStart[i + 1];
#pragma unroll
        for (int q = 0; q < SLICE; ++q) {
            const int j = jBeg + q;
            if (j >= jEnd) break;
            const float4 b = bs[q];
            f3 dp = pa - mk3(b.x, b.y, b.z);
            float r = a.w + b.w;
            if (dot3(dp, dp) <= r * r) {
                if (!solvedI && !B.solved[j]) {
                    const bool bothStatic = !(B.plane(BP_MASS)[i].y > 0.0f) && !(B.plane(BP_MASS)[j].y > 0.0f);
                    if (!(keepUnsolvedPairs && bothStatic)) continue;
                }
                const int sj = adjStart[j], ej = adjStart[j + 1];
                int s, e, other;
                if (ei - si <= ej - sj) { s = si; e = ei; other = j; } else { s = sj; e = ej; other = i; }
                if (!adj_lists(adjOther, s, e, other)) sHits[atomicAdd(&sNumHits, 1)] = make_int2(i, j);
            }
        }
    }
    __syncthreads();
    const int nHits = sNumHits;
    if (threadIdx.x == 0 && nHits) atomicAdd(&cnt->numCandidates, nHits);
    for (int h = gj; h < nHits; h += PAIRS) {
        const int2 pr = sHits[h];
        narrow_pair(B, src, dst, -1, pr.x, pr.y, step, sWs[gj], cnt, nextCount, degForce, degInc, P, gl, gBase);
    }
#ifdef AVBD_DF_STATS
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < kFrontStampBlocks) gFrontBlockStamps[16 * blockIdx.x + 15] = __builtin_amdgcn_s_memrealtime();
#endif
}

/* Joint / Spring / IgnoreCollision: initialize() is a no-op returning true; apply the row decay
 * (solver.cpp:281-293 incl. the finite-stiffness cap) and count adjacency degrees. */
__global__ void k_gforce_prepare(BodyTable B, GForce* gf, int ng, int* degForce, int* degInc, SolverParams P, int decay)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    GForce& f = gf[g];
    int rows = f.kind == FK_HOST ? 0 : force_row_count(f.kind, 0);   /* (a host force's rows are decayed on the host, solver.cpp:281-293) */
    if (decay) {
        for (int r = 0; r < rows; ++r) {
            if (!P.postStabilize) f.lambda[r] *= P.alpha * P.gamma;
            f.penalty[r] = rclamp(f.penalty[r] * P.gamma, kPenaltyMin, kPenaltyMax);
            if (f.stiffness[r] > 0.0f && f.stiffness[r] < FLT_MAX) f.penalty[r] = rmin(f.penalty[r], f.stiffness[r]);
        }
    }
    int entries = gforce_entries(f);
    if (f.bodyA >= 0) { atomicAdd(&degForce[f.bodyA], 1); if (B.solved[f.bodyA]) atomicAdd(&degInc[f.bodyA], entries); }
    if (f.bodyB >= 0) { atomicAdd(&degForce[f.bodyB], 1); if (B.solved[f.bodyB]) atomicAdd(&degInc[f.bodyB], entries); }
}

/* inclusive scan over the 1024 threads of the workgroup: wave shuffles + one 16-entry exchange (two barriers) */
__device__ __forceinline__ unsigned long long block_scan_inclusive_u64(unsigned long long v, unsigned long long* sWaveTotals /* [16] */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned lo = __shfl_up((unsigned)v, off), hi = __shfl_up((unsigned)(v >> 32), off);
        if (lane >= off) v += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 63) sWaveTotals[wv] = v;
    __syncthreads();
    unsigned long long base = 0;
    for (int w = 0; w < wv; ++w) base += sWaveTotals[w];
    __syncthreads();
    return v + base;
}

/* ---- scans over more bodies than one workgroup holds: one tile of 4096 per workgroup, decoupled look-back ----------------
 * Every workgroup publishes the sum of its tile as ONE 16-byte record {low, high, stamp, 0} (sc1 store: written through to the
 * level all XCDs share, observed untorn) and adds up the records of the tiles before it (sc1 loads, polled until they carry
 * this launch's stamp).  Workgroups are dispatched in index order, so a workgroup only ever waits for ones that already run.
 * `stamp` is unique per launch (the context counts them), so the records are never reset.  (The single-workgroup forms of
 * these scans took 98 + 125 us per step at 64 000 bodies: sixteen tiles one after the other, each a memory round trip and
 * two workgroup barriers.) */
typedef unsigned tile_v4u __attribute__((ext_vector_type(4)));
constexpr int kTileBodies = 4096;
__device__ __forceinline__ unsigned long long tile_exclusive_prefix(tile_v4u* agg, int tile, unsigned long long mine, unsigned stamp, unsigned long long* sBcast)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)agg, 0, (tile + 1) * 16, 0x00020000);
    if (threadIdx.x == 0) {
        const tile_v4u rec = {(unsigned)mine, (unsigned)(mine >> 32), stamp, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(rec, r, tile * 16, 0, 16);
    }
    if (threadIdx.x < 64) {
        unsigned long long prefix = 0;
        for (int base = 0; base < tile; base += 64) {
            const int p = base + (int)threadIdx.x;
            unsigned long long v = 0;
            bool have = p >= tile;
            while (!__all(have)) {
                if (!have) {
                    const tile_v4u rec = __builtin_amdgcn_raw_buffer_load_b128(r, p * 16, 0, 16);
                    if (rec.z == stamp) { v = ((unsigned long long)rec.y << 32) | rec.x; have = true; }
                }
                if (!__all(have)) __builtin_amdgcn_s_sleep(2);
            }
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
                v += ((unsigned long long)hi << 32) | lo;
            }
            prefix += v;
        }
        if (threadIdx.x == 0) *sBcast = prefix;
    }
    __syncthreads();
    const unsigned long long prefix = *sBcast;
    __syncthreads();
    return prefix;
}

/* Exclusive scans of the two degree arrays (n+1 outputs each), and cursor reset: every thread takes four consecutive bodies
 * (coalesced), one shuffle-based workgroup scan per tile, the offset of the tile by look-back.  `ndCached` >= 0: the sweep
 * order of this scene is already on the device (it only changes with the bodies) — the step's counters that go with it are
 * set here instead of by k_order_by_index. */
__global__ __launch_bounds__(1024) void k_scan(int n, int* degForce, int* degInc, int* adjStart, int* incStart, Counters* cnt, const int* nextCount,
                                               tile_v4u* agg, unsigned stamp, int ndCached, int residentWaves, int manifoldCap)
{
    __shared__ unsigned long long sWaveTotals[16];
    __shared__ unsigned long long sBcast;
    const int t = threadIdx.x, tile = blockIdx.x;
    const int i0 = tile * kTileBodies + 4 * t;
    unsigned long long v[4];                                          /* high word: incidence entries, low word: forces */
    unsigned long long mine = 0;
    for (int k = 0; k < 4; ++k) {
        const int i = i0 + k;
        v[k] = i < n ? (((unsigned long long)(unsigned)degInc[i] << 32) | (unsigned)degForce[i]) : 0ull;
        mine += v[k];
    }
    const unsigned long long incl = block_scan_inclusive_u64(mine, sWaveTotals);
    __shared__ unsigned long long sTileSum;
    if (t == 1023) sTileSum = incl;
    __syncthreads();
    const unsigned long long before = tile_exclusive_prefix(agg, tile, sTileSum, stamp, &sBcast);
    unsigned long long off = before + incl - mine;
    for (int k = 0; k < 4; ++k) {
        const int i = i0 + k;
        if (i < n) { adjStart[i] = (int)(unsigned)off; incStart[i] = (int)(off >> 32); degForce[i] = 0; degInc[i] = 0; /* become fill cursors */ }
        off += v[k];
    }
    if (t == 1023 && tile == (int)gridDim.x - 1) {
        const unsigned long long total = before + incl;
        adjStart[n] = (int)(unsigned)total; incStart[n] = (int)(total >> 32); cnt->numManifolds = min(*nextCount, manifoldCap);   /* (the narrow phase counts the survivors it had no slot for as well: Counters::overflow bit 1 says so) */
        if (ndCached >= 0) {
            atomicExch(&cnt->numLevels, ndCached > 0 ? 1 : 0);
            if (ndCached > residentWaves) atomicOr(&cnt->residentBad, 2);
        }
    }
}

/* A position in a body's adjacency list.  Bodies this context sweeps have a handful of forces each: one returning atomic per entry.  A body that
 * is not swept — the ground under a 64 000-cube pile carries 6 400 manifolds — would serialise thousands of returning atomics on ONE word
 * (13 ns each: 68 of the launch's 68 us): the lanes of a wave that target the same such body take their positions from one atomic. */
__device__ __forceinline__ int adj_position(int body, bool want, const int* __restrict__ solved, const int* __restrict__ adjStart, int* cursor)
{
    const int lane = threadIdx.x & 63;
    const bool shared = want && !solved[body];
    int pos = -1;
    if (want && !shared) pos = adjStart[body] + atomicAdd(&cursor[body], 1);
    unsigned long long m = __ballot(shared);
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        const int v = __shfl(body, l, 64);
        const unsigned long long same = __ballot(shared && body == v);
        int base = 0;
        if (lane == l) base = atomicAdd(&cursor[v], (int)__popcll(same));
        base = __shfl(base, l, 64);
        if (shared && body == v) pos = adjStart[v] + base + (int)__popcll(same & ((1ull << lane) - 1ull));
        m &= ~same;
    }
    return pos;
}

__global__ void k_fill_adj(ManifoldTable M, const Counters* cnt, const GForce* gf, int ng, const int* adjStart, int* cursor,
                           int* adjRef, int* adjOther, uint64_t* adjKey, const int* __restrict__ solved)
{
    const int nm = cnt->numManifolds;
    const int total = nm + ng;
    const int stride = gridDim.x * blockDim.x;
    for (int base = blockIdx.x * blockDim.x; base < total; base += stride) {      /* (whole waves stay in the loop: adj_position uses wave-wide votes) */
        const int item = base + (int)threadIdx.x;
        int a = -1, b = -1, kind = 0, index = 0; uint64_t key = 0;
        if (item < total) {
            if (item < nm) { int2 p = M.pair[item]; a = p.x; b = p.y; kind = FK_MANIFOLD; index = item; key = M.key[item]; }
            else { const GForce& f = gf[item - nm]; a = f.bodyA; b = f.bodyB; kind = gforce_ref_kind(f.kind); index = item - nm; key = f.key; }
        }
        /* (both positions first: two returning atomics in flight instead of one behind the other's stores) */
        const int pa = adj_position(a, a >= 0, solved, adjStart, cursor);
        const int pb = adj_position(b, b >= 0, solved, adjStart, cursor);
        if (a >= 0) { adjRef[pa] = make_inc(1, kind, index); adjOther[pa] = b; adjKey[pa] = key; }
        if (b >= 0) { adjRef[pb] = make_inc(0, kind, index); adjOther[pb] = a; adjKey[pb] = key; }
    }
}

/* Per dynamic body: order its adjacency newest-first (= the reference's body->forces walk,
 * force.cpp:21-29) and expand it into the row-incidence list consumed by k_primal.
 * A body with at most kSortRegs forces (every cube of a pile) takes them into registers — all loads in flight at once —, ranks every entry against the
 * others (position = how many are newer, incidence offset = how many entries those bring) and stores it where it belongs: no chain of dependent loads
 * through an insertion sort in memory.  Longer lists (the ground slab is not swept; a body under a heap of small ones) keep the insertion sort. */
constexpr int kSortRegs = 16;
__global__ void k_sort_expand(BodyTable B, ManifoldTable M, const GForce* gf, const int* adjStart, int* adjRef, int* adjOther, uint64_t* adjKey,
                              const int* incStart, int* inc, int* level, int* cursor, Counters* cnt)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B.n) return;
    cursor[b] = 0;
    if (level) level[b] = 0;                                          /* (sweep by creation index: k_order_by_index is the only writer) */
    if (!B.solved[b]) return;
    const int s = adjStart[b], e = adjStart[b + 1];
    const int o0 = incStart[b];
    int o = o0;
    bool nonManifold = false;
    if (e - s <= kSortRegs) {
        const int d = e - s;
        uint64_t key[kSortRegs]; int ref[kSortRegs], oth[kSortRegs], ent[kSortRegs];
#pragma unroll
        for (int i = 0; i < kSortRegs; ++i) {
            key[i] = 0; ref[i] = 0; oth[i] = 0;
            if (i < d) { key[i] = adjKey[s + i]; ref[i] = adjRef[s + i]; oth[i] = adjOther[s + i]; }
        }
#pragma unroll
        for (int i = 0; i < kSortRegs; ++i) {
            ent[i] = 0;
            if (i < d) {
                const int kind = inc_kind(ref[i]), idx = inc_index(ref[i]);
                if (kind == FK_MANIFOLD) ent[i] = M.nc[idx];
                else if (kind == FK_JOINT) ent[i] = 2;
                else if (kind == FK_SPRING) { ent[i] = gf[idx].kind == FK_HOST ? gforce_entries(gf[idx]) : 1; nonManifold = true; }
            }
        }
#pragma unroll
        for (int i = 0; i < kSortRegs; ++i) {
            if (i < d) {
                int rank = 0, off = 0;
#pragma unroll
                for (int j = 0; j < kSortRegs; ++j) {
                    const bool newer = j < d && (key[j] > key[i] || (key[j] == key[i] && j < i));
                    rank += newer ? 1 : 0; off += newer ? ent[j] : 0;
                }
                adjKey[s + rank] = key[i]; adjRef[s + rank] = ref[i]; adjOther[s + rank] = oth[i];
                const int kind = inc_kind(ref[i]), idx = inc_index(ref[i]), isA = inc_isA(ref[i]);
                int* out = inc + o0 + off;
                if (kind == FK_MANIFOLD) { for (int c = 0; c < ent[i]; ++c) out[c] = make_inc(isA, FK_MANIFOLD, idx * 4 + c); }
                else if (kind == FK_JOINT) { out[0] = make_inc(isA, FK_JOINT, idx * 2); out[1] = make_inc(isA, FK_JOINT, idx * 2 + 1); }
                else if (kind == FK_SPRING) { for (int q = 0; q < ent[i]; ++q) out[q] = make_inc(isA, FK_SPRING, idx | (q << 24)); }
                o += ent[i];
            }
        }
    } else {
        for (int i = s + 1; i < e; ++i) {
            uint64_t k = adjKey[i]; int r = adjRef[i], oo = adjOther[i];
            int j = i - 1;
            while (j >= s && adjKey[j] < k) { adjKey[j + 1] = adjKey[j]; adjRef[j + 1] = adjRef[j]; adjOther[j + 1] = adjOther[j]; --j; }
            adjKey[j + 1] = k; adjRef[j + 1] = r; adjOther[j + 1] = oo;
        }
        for (int i = s; i < e; ++i) {
            int r = adjRef[i];
            int kind = inc_kind(r), idx = inc_index(r), isA = inc_isA(r);
            if (kind == FK_MANIFOLD) {
                int nc = M.nc[idx];
                for (int c = 0; c < nc; ++c) inc[o++] = make_inc(isA, FK_MANIFOLD, idx * 4 + c);
            } else if (kind == FK_JOINT) {
                inc[o++] = make_inc(isA, FK_JOINT, idx * 2);
                inc[o++] = make_inc(isA, FK_JOINT, idx * 2 + 1);
            } else if (kind == FK_SPRING) {
                const int parts = gf[idx].kind == FK_HOST ? gforce_entries(gf[idx]) : 1;    /* a host force: three rows per entry */
                for (int q = 0; q < parts; ++q) inc[o++] = make_inc(isA, FK_SPRING, idx | (q << 24));
            }
            if (kind == FK_SPRING) nonManifold = true;                     /* joints are fine for the resident schedule, springs are not */
        }
    }
    /* the resident schedule (avbd_resident.h) keeps one Manifold contact per lane of the body's wavefront */
    if (nonManifold || o - o0 > 64) atomicOr(&cnt->residentBad, 1);
    else if (o - o0 > kResLiteMaxEntriesDecl) atomicOr(&cnt->residentBad, 4);
}

/* Dependency levels of the Gauss-Seidel sweep: level(b) = 1 + max level of dynamic neighbours
 * created later than b (they precede b in the reference's body list).  Single workgroup. */
constexpr int kLevelsLdsBodies = 8192, kLevelsLdsNbrs = 49152;
constexpr int kLevelsLdsBytes = (kLevelsLdsBodies + kLevelsLdsNbrs) * 2;
__global__ __launch_bounds__(1024) void k_levels(BodyTable B, const int* adjStart, const int* adjRef, const int* adjOther, int* level,
                                                  int* lvlCount, int* lvlStart, int* order, Counters* cnt, int lvlCap, int residentWaves)
{
    __shared__ int sChanged, sMax;
    __shared__ int sScan[1024];
    extern __shared__ unsigned short sDyn[];      /* [kLevelsLdsBodies] levels, then [kLevelsLdsNbrs] compacted neighbour lists */
    const int n = B.n, t = threadIdx.x;
    if (n <= kLevelsLdsBodies && adjStart[n] <= kLevelsLdsNbrs) {
        /* small scenes: levels and the later-created swept neighbours of every body live in LDS (16-bit indices), compacted
         * at the start of the body's own CSR segment — a sweep is then LDS reads only, no dependent global loads */
        unsigned short* sLevel = sDyn;
        unsigned short* sNbr = sDyn + kLevelsLdsBodies;
        constexpr int kPer = kLevelsLdsBodies / 1024;
        int seg[kPer], cnt_[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int b = t + q * 1024;
            seg[q] = 0; cnt_[q] = -1;
            if (b < n) sLevel[b] = 0;
            if (b < n && B.solved[b]) {
                const int s0 = adjStart[b], s1 = adjStart[b + 1];
                int c = 0;
                for (int k = s0; k < s1; ++k) {
                    const int o = adjOther[k];
                    if (o > b && inc_kind(adjRef[k]) != FK_IGNORE && B.solved[o]) sNbr[s0 + c++] = (unsigned short)o;
                }
                seg[q] = s0; cnt_[q] = c;
            }
        }
        __syncthreads();
        for (int sweep = 0; sweep <= n; ++sweep) {
            if (t == 0) sChanged = 0;
            __syncthreads();
            int changed = 0;
#pragma unroll
            for (int q = 0; q < kPer; ++q) {
                if (cnt_[q] < 0) continue;
                const int b = t + q * 1024;
                int lv = 1;
                for (int z = 0; z < cnt_[q]; ++z) { int lo = (int)((volatile unsigned short*)sLevel)[sNbr[seg[q] + z]] + 1; if (lo > lv) lv = lo; }
                if (lv != (int)sLevel[b]) { sLevel[b] = (unsigned short)lv; changed = 1; }
            }
            if (changed) sChanged = 1;
            __syncthreads();
            int c = sChanged;
            __syncthreads();
            if (!c) break;
        }
        for (int b = t; b < n; b += 1024) level[b] = sLevel[b];
        __syncthreads();
    } else {
    for (int sweep = 0; sweep <= n; ++sweep) {
        if (t == 0) sChanged = 0;
        __syncthreads();
        int changed = 0;
        for (int b = t; b < n; b += 1024) {
            if (!B.solved[b]) continue;
            int lv = 1;
            for (int k = adjStart[b]; k < adjStart[b + 1]; ++k) {
                int o = adjOther[k];
                if (o > b && inc_kind(adjRef[k]) != FK_IGNORE && B.solved[o]) {
                    int lo = ((volatile int*)level)[o] + 1;
                    if (lo > lv) lv = lo;
                }
            }
            if (lv != level[b]) { level[b] = lv; changed = 1; }
        }
        if (changed) sChanged = 1;
        __syncthreads();
        int c = sChanged;
        __syncthreads();
        if (!c) break;
    }
    }
    /* histogram */
    if (t == 0) sMax = 0;
    for (int l = t; l < lvlCap; l += 1024) lvlCount[l] = 0;
    __syncthreads();
    int mx = 0;
    for (int b = t; b < n; b += 1024) {
        int lv = level[b];
        if (lv > 0) { if (lv < lvlCap) atomicAdd(&lvlCount[lv], 1); if (lv > mx) mx = lv; }
    }
    atomicMax(&sMax, mx);
    __syncthreads();
    const int L = sMax; /* levels are 1..L */
    /* exclusive scan of lvlCount[1..L] into lvlStart[0..L] (lvlStart[l-1] = start of level l) */
    const int chunk = (L + 1023) / 1024;
    const int lo = t * chunk, hi = min(L, lo + chunk);
    int a = 0;
    for (int i = lo; i < hi; ++i) a += lvlCount[i + 1];
    sScan[t] = a;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (t >= off) ? sScan[t - off] : 0;
        __syncthreads();
        sScan[t] += v;
        __syncthreads();
    }
    int base = sScan[t] - a;
    for (int i = lo; i < hi; ++i) { lvlStart[i] = base; base += lvlCount[i + 1]; lvlCount[i + 1] = 0; }
    if (t == 1023) lvlStart[L] = sScan[1023];
    __syncthreads();
    for (int b = t; b < n; b += 1024) {
        int lv = level[b];
        if (lv > 0 && lv < lvlCap) order[lvlStart[lv - 1] + atomicAdd(&lvlCount[lv], 1)] = b;
    }
    /* Counters / DiagAccum words are updated with atomics by other workgroups (possibly on other XCDs, whose L2s are not
     * coherent with this one): a plain store here could write a stale copy of its neighbours back over them. */
    if (t == 0) { atomicExch(&cnt->numLevels, L); if (L >= lvlCap) atomicOr(&cnt->overflow, 8); if (sScan[1023] > residentWaves) atomicOr(&cnt->residentBad, 2); }
}

/* ------------------------------------------------------------------------------------------
 * K4 predict — solver.cpp:299-337
 * ------------------------------------------------------------------------------------------ */
/* (warn_log, sanitize3, sanitizeq: avbd_util.h) */

/* (returns whether the body is dynamic: the caller counts them — one atomic per workgroup, not one per wavefront on the Counters line) */
__device__ __forceinline__ bool predict_body(const BodyTable& B, int i, const SolverParams& P, Counters* cnt)
{
    float4 pr = B.plane(BP_POSR)[i];
    f3 pos = mk3(pr.x, pr.y, pr.z);
    q4 rot = ldq(B.plane(BP_QUAT), i);
    f3 w = ld3(B.plane(BP_ANGV), i);
    {   /* clampAngularVelocity, solver.cpp:85-92 */
        const float maxSpeed = 80.0f;
        float len = len3(w);
        if (len > maxSpeed && len > kVecEps) w = w * (maxSpeed / len);
    }
    B.plane(BP_IPOS)[i] = pack3(pos, 0.f);
    B.plane(BP_IQUAT)[i] = packq(rot);
    f3 npos = pos; q4 nrot = rot;
    const float invMass = B.plane(BP_MASS)[i].y;
    if (invMass > 0.0f) {
        f3 v = ld3(B.plane(BP_LINV), i);
        f3 vprev = ld3(B.plane(BP_PLIN), i);
        sanitize3(v, cnt, warn_key(0, 0, i, 0), i, WK_LINVEL);
        sanitize3(w, cnt, warn_key(0, 0, i, 1), i, WK_ANGVEL);
        const float dt = P.dt;
        const f3 g = mk3(P.gx, P.gy, P.gz);
        npos = pos + v * dt + g * (dt * dt);
        q4 omega = mkq(w.x, w.y, w.z, 0.0f);
        nrot = qnormalize(rot + (omega * rot) * (0.5f * dt));
        float gLen = len3(g);
        float aw = 0.0f;
        if (gLen > 1e-5f) {
            f3 accel = (v - vprev) / dt;
            float proj = dot3(accel, g / gLen);
            aw = rclamp(proj / gLen, 0.0f, 1.0f);
            if (!isfinite(aw)) aw = 0.0f;
        }
        pos = pos + (v * dt + g * (aw * dt * dt));
        rot = nrot;
        sanitize3(pos, cnt, warn_key(0, 0, i, 2), i, WK_PRED_POS);
        sanitizeq(rot, cnt, warn_key(0, 0, i, 3), i, WK_PRED_ROT);
        B.plane(BP_LINV)[i] = pack3(v, 0.f);
    }
    B.plane(BP_ANGV)[i] = pack3(w, 0.f);
    B.plane(BP_NPOS)[i] = pack3(npos, 0.f);
    B.plane(BP_NQUAT)[i] = packq(nrot);
    B.plane(BP_POSR)[i] = pack3(pos, pr.w);
    B.plane(BP_QUAT)[i] = packq(rot);
    return invMass > 0.0f;
}

/* (+ the done-counters of the task schedule: primalDone[cap] | dualDone[cap], nullptr in the level mode — a fill launch less per step) */
constexpr int kPredictThreads = 512;
__global__ __launch_bounds__(kPredictThreads) void k_predict(BodyTable B, SolverParams P, Counters* cnt, int* doneFlags)
{
    __shared__ int sDyn[kPredictThreads / 64];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool dyn = false;
    if (i < B.n) {
        if (doneFlags) { doneFlags[i] = 0; doneFlags[B.cap + i] = 0; }
        dyn = predict_body(B, i, P, cnt);
    }
    /* the dynamic-body count of the diagnostics: 64 000 bodies are 1 000 wavefronts, and 1 000 atomics on the Counters line are 13 us of a 16 us launch */
    const int mine = (int)__popcll(__ballot(dyn));
    if ((threadIdx.x & 63) == 0) sDyn[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int k = 0; k < kPredictThreads / 64; ++k) total += sDyn[k];
        if (total) atomicAdd(&cnt->dynamicBodies, total);
    }
}

/* ------------------------------------------------------------------------------------------
 * Small scenes (<= 4096 bodies, <= 6144 adjacency entries): the whole per-step topology — what k_scan, k_fill_adj,
 * k_sort_expand and k_levels do in four launches — plus the predictor and the reset of the data-flow flags, as ONE
 * workgroup working in LDS.  Five dependent launches of a few microseconds each become phases separated by
 * workgroup barriers.  Same outputs (adjStart/adjRef/adjOther sorted newest-first, incStart/inc, level order).
 * ------------------------------------------------------------------------------------------ */
} // namespace avbd
#include "avbd_place.h"
namespace avbd {
constexpr int kTopoBodies = 4096, kTopoEntries = 6144;
constexpr int kTopoIntWords = (kTopoBodies + 1) * 2 + kTopoBodies;            /* adjStart | incStart | cursor */
constexpr int kTopoLdsBytes = kTopoIntWords * 4 + 8 /* align */ + kTopoEntries * 8 /* keys, later neighbour lists */ +
                              kTopoBodies * 2 /* levels */ + kTopoEntries * 2 * 3 /* entry body | rank | rows */ + kTopoBodies /* swept? */;

/* Large scenes, data-flow schedules: the sweep order only has to be a linear extension of "later-created swept neighbours
 * first", and descending creation index IS one — no level relaxation (a single workgroup needs one sweep per level: 15 ms
 * at 64 000 bodies / 128 levels).  All swept bodies form one "level"; the level schedule (AVBD_MODE=levels) keeps k_levels. */
__global__ __launch_bounds__(1024) void k_order_by_index(BodyTable B, int* level, int* lvlStart, int* order, Counters* cnt, int residentWaves)
{
    __shared__ unsigned long long sWaveTotals[16];
    __shared__ int sTotal;
    const int n = B.n, t = threadIdx.x;
    int running = 0;
    /* pass 1: how many swept bodies */
    int mine = 0;
    for (int b = t; b < n; b += 1024) mine += B.solved[b] ? 1 : 0;
    const int inclAll = (int)block_scan_inclusive_u64((unsigned long long)mine, sWaveTotals);
    if (t == 1023) sTotal = inclAll;
    __syncthreads();
    const int nd = sTotal;
    /* pass 2: ascending rank by tiles of 4096 bodies (four consecutive bodies per thread); slot = nd - 1 - rank (newest first) */
    for (int base = 0; base < n; base += 4096) {
        const int b0 = base + 4 * t;
        int f[4], mine4 = 0;
        for (int k = 0; k < 4; ++k) { f[k] = (b0 + k < n && B.solved[b0 + k]) ? 1 : 0; mine4 += f[k]; }
        const int incl = (int)block_scan_inclusive_u64((unsigned long long)mine4, sWaveTotals);
        int rank = running + incl - mine4;
        for (int k = 0; k < 4; ++k) {
            const int b = b0 + k;
            if (b < n) level[b] = f[k];
            if (f[k]) order[nd - 1 - rank++] = b;
        }
        if (t == 1023) sTotal = incl;
        __syncthreads();
        running += sTotal;
        __syncthreads();
    }
    if (t == 0) {
        lvlStart[0] = 0; lvlStart[1] = nd;
        atomicExch(&cnt->numLevels, nd > 0 ? 1 : 0);
        if (nd > residentWaves) atomicOr(&cnt->residentBad, 2);
    }
}

__global__ __launch_bounds__(1024) void k_topology_small(BodyTable B, ManifoldTable M, const GForce* gf, int ng, Counters* cnt, const int* nextCount,
                                                          int* degForce, int* degInc, int* adjStart, int* incStart, int* adjRef, int* adjOther,
                                                          uint64_t* adjKey, int2* adjTmp, int* inc, int* level, int* lvlCount, int* lvlStart, int* order, int lvlCap,
                                                          int residentWaves, SolverParams P, int* flags, int nFlags, int byIndex,
                                                          const int2* placeCrit, int* placeTail, int* placeSlotOf, int placeW, int* place, PlaceInfo* placeInfo, int placeKeep, int placeRounds, int sweptKept)
{
    extern __shared__ int sTopo[];
    __shared__ unsigned long long sWaveTotals[16];
    __shared__ int sMax, sBad, sTotal, sHist[1024];
    const int n = B.n, t = threadIdx.x;
    int* sAdj = sTopo;
    int* sInc = sTopo + (kTopoBodies + 1);
    int* sCur = sTopo + 2 * (kTopoBodies + 1);
    unsigned long long* sKey = reinterpret_cast<unsigned long long*>(sTopo + ((kTopoIntWords + 1) & ~1));
    unsigned short* sNbr = reinterpret_cast<unsigned short*>(sKey);                   /* reused once the keys are dead */
    unsigned short* sLevel = reinterpret_cast<unsigned short*>(sKey + kTopoEntries);
    unsigned short* sEntBody = sLevel + kTopoBodies;
    unsigned short* sEntRank = sEntBody + kTopoEntries;
    unsigned short* sEntRows = sEntRank + kTopoEntries;
    unsigned char* sSwept = reinterpret_cast<unsigned char*>(sEntRows + kTopoEntries);
#ifdef AVBD_DF_STATS
    unsigned long long* tstamp = reinterpret_cast<unsigned long long*>(flags + nFlags + 16 + 1024 * 32 + 20 * 1024 * 8);
    if (t == 0 && blockIdx.x == 0) tstamp[0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (blockIdx.x > 0) {
        const int nb = (int)gridDim.x - 1, b = ((int)blockIdx.x - 1) * 1024 + t;
        bool dyn = false;
        if (b < n) dyn = predict_body(B, b, P, cnt);
        for (int i = ((int)blockIdx.x - 1) * 1024 + t; i < nFlags; i += nb * 1024) flags[i] = 0;
        const int mine = (int)__popcll(__ballot(dyn));                 /* the dynamic-body count of the diagnostics: one atomic per workgroup */
        if ((t & 63) == 0) sWaveTotals[t >> 6] = (unsigned long long)mine;
        __syncthreads();
        if (t == 0) { int total = 0; for (int k = 0; k < 16; ++k) total += (int)sWaveTotals[k]; if (total) atomicAdd(&cnt->dynamicBodies, total); }
        return;
    }
    const int nm = min(*nextCount, M.cap);                           /* (the narrow phase counts the survivors it had no slot for as well: Counters::overflow bit 1 says so) */
    const int4 placedFor = (place && placeKeep) ? *reinterpret_cast<const int4*>(placeInfo) : make_int4(-1, 0, 0, 0);   /* {nd, groups, grouped, valid} of the table in memory */
    if (t == 0) { atomicExch(&cnt->numManifolds, nm); sBad = 0; }   /* atomics only on Counters words inside a kernel that also adds to them */
    /* what phases 1 and 2 read first is requested before phase 0 runs (every phase of this kernel starts with a memory round trip, and nothing
     * overlaps them but the order of the requests): this thread's share of the degree arrays and its first force */
    const int chunk = (n + 1023) / 1024;
    const int lo = t * chunk, hi = min(n, lo + chunk);
    unsigned long long mine = 0;
    int deg0F = 0, deg0I = 0;
    for (int i = lo; i < hi; ++i) {
        const int da = degForce[i], db = degInc[i];
        if (i == lo) { deg0F = da; deg0I = db; }
        mine += ((unsigned long long)(unsigned)db << 32) | (unsigned)da;
    }
    int2 pair0 = make_int2(-1, -1); uint64_t key0 = 0; int rows0 = 0;
    if (t < nm) { pair0 = M.pair[t]; key0 = M.key[t]; rows0 = M.nc[t]; }

    /* ---- phase 0: predictor (solver.cpp:299-337) and flag reset: independent of everything below — the work of the launch's OTHER workgroups,
     *      1024 bodies each, beside this one (3 of its 17 us when it did them itself) ---- */
    for (int b = t; b < n; b += 1024) { sSwept[b] = B.solved[b] ? 1 : 0; sLevel[b] = 0; }

#ifdef AVBD_DF_STATS
    __syncthreads(); if (t == 0) tstamp[1] = __builtin_amdgcn_s_memrealtime();
#endif
    /* ---- phase 1: exclusive scans of the two degree arrays (one 64-bit scan: high word rows, low word forces) ---- */
    {
        const unsigned long long incl = block_scan_inclusive_u64(mine, sWaveTotals);
        unsigned long long base = incl - mine;
        for (int i = lo; i < hi; ++i) {
            const int da = i == lo ? deg0F : degForce[i], db = i == lo ? deg0I : degInc[i];
            const int a0 = (int)(unsigned)base, b0 = (int)(base >> 32);
            sAdj[i] = a0; sInc[i] = b0; adjStart[i] = a0; incStart[i] = b0; sCur[i] = 0;
            base += ((unsigned long long)(unsigned)db << 32) | (unsigned)da;
            degForce[i] = 0; degInc[i] = 0;                          /* the next narrow phase counts from zero */
        }
        if (t == 1023) {
            const int ea = (int)(unsigned)incl, eb = (int)(incl >> 32);
            sAdj[n] = ea; sInc[n] = eb; adjStart[n] = ea; incStart[n] = eb;
        }
    }
    __syncthreads();
    const int E = sAdj[n];
    const bool inLds = E <= kTopoEntries;

#ifdef AVBD_DF_STATS
    __syncthreads(); if (t == 0) tstamp[2] = __builtin_amdgcn_s_memrealtime();
#endif
    /* ---- phase 2: scatter every force into the CSR segments of its two bodies ---- */
    for (int item = t; item < nm + ng; item += 1024) {
        int a, b, kind, index, rows; uint64_t key;
        if (item < nm) {
            const int2 p = item == t ? pair0 : M.pair[item];
            a = p.x; b = p.y; kind = FK_MANIFOLD; index = item; key = item == t ? key0 : M.key[item]; rows = item == t ? rows0 : M.nc[item];
        }
        else { const GForce& f = gf[item - nm]; a = f.bodyA; b = f.bodyB; kind = f.kind; index = item - nm; key = f.key; rows = gforce_entries(f); kind = gforce_ref_kind(kind); }
        if (a >= 0) {
            const int pos = sAdj[a] + atomicAdd(&sCur[a], 1);
            adjTmp[pos] = make_int2(make_inc(1, kind, index), b);
            if (inLds) { sKey[pos] = key; sEntBody[pos] = (unsigned short)a; sEntRows[pos] = (unsigned short)rows; } else adjKey[pos] = key;
        }
        if (b >= 0) {
            const int pos = sAdj[b] + atomicAdd(&sCur[b], 1);
            adjTmp[pos] = make_int2(make_inc(0, kind, index), a);
            if (inLds) { sKey[pos] = key; sEntBody[pos] = (unsigned short)b; sEntRows[pos] = (unsigned short)rows; } else adjKey[pos] = key;
        }
    }
    __syncthreads();

#ifdef AVBD_DF_STATS
    __syncthreads(); if (t == 0) tstamp[3] = __builtin_amdgcn_s_memrealtime();
#endif
    /* ---- phase 3: newest-first order inside every swept body's segment, the row-incidence list, and the later-created
     *      swept neighbours of every swept body (what the level relaxation needs) ---- */
    if (inLds && byIndex) {
        /* one thread per ENTRY, one pass (the sweep by creation index needs no neighbour lists, so the keys stay where they are): rank of an
         * entry = entries of the same body with a larger key, its rows start behind the rows of those entries */
        for (int pos = t; pos < E; pos += 1024) {
            const int2 ro = adjTmp[pos];
            const int b = sEntBody[pos];
            const int s0 = sAdj[b], s1 = sAdj[b + 1];
            int rank = pos - s0, off = 0;                              /* unswept bodies keep the scatter order (never visited in order) */
            const bool swept = sSwept[b] != 0;
            if (swept) {
                const unsigned long long k = sKey[pos];
                rank = 0;
                for (int j = s0; j < s1; ++j) { const bool newer = sKey[j] > k; rank += newer ? 1 : 0; off += newer ? (int)sEntRows[j] : 0; }
            }
            adjRef[s0 + rank] = ro.x; adjOther[s0 + rank] = ro.y;
            if (swept) {
                const int kind = inc_kind(ro.x), idx = inc_index(ro.x), isA = inc_isA(ro.x);
                int* out = inc + sInc[b] + off;
                if (kind == FK_MANIFOLD) { const int nc = sEntRows[pos]; for (int c = 0; c < nc; ++c) out[c] = make_inc(isA, FK_MANIFOLD, idx * 4 + c); }
                else if (kind == FK_JOINT) { out[0] = make_inc(isA, FK_JOINT, idx * 2); out[1] = make_inc(isA, FK_JOINT, idx * 2 + 1); }
                else if (kind == FK_SPRING) { const int parts = gf[idx].kind == FK_HOST ? gforce_entries(gf[idx]) : 1; for (int q = 0; q < parts; ++q) out[q] = make_inc(isA, FK_SPRING, idx | (q << 24)); sBad = 1; }
            }
        }
    } else if (inLds) {
        /* one thread per ENTRY: rank of an entry = entries of the same body with a larger key */
        for (int pos = t; pos < E; pos += 1024) {
            const int b = sEntBody[pos];
            int rank = pos - sAdj[b];                                 /* unswept bodies keep the scatter order (never visited in order) */
            if (sSwept[b]) {
                const unsigned long long k = sKey[pos];
                const int s0 = sAdj[b], s1 = sAdj[b + 1];
                rank = 0;
                for (int j = s0; j < s1; ++j) rank += sKey[j] > k ? 1 : 0;
            }
            sEntRank[pos] = (unsigned short)rank;
        }
        for (int b = t; b < n; b += 1024) sCur[b] = 0;                 /* becomes the neighbour count */
        __syncthreads();
        for (int pos = t; pos < E; pos += 1024) {
            const int b = sEntBody[pos];
            const int s0 = sAdj[b], s1 = sAdj[b + 1];
            const int rank = sEntRank[pos];
            const int2 ro = adjTmp[pos];
            adjRef[s0 + rank] = ro.x; adjOther[s0 + rank] = ro.y;
            if (sSwept[b]) {
                /* row-incidence entries of this force start after those of every force that precedes it in the list */
                int off = 0;
                for (int j = s0; j < s1; ++j) off += sEntRank[j] < rank ? sEntRows[j] : 0;
                const int kind = inc_kind(ro.x), idx = inc_index(ro.x), isA = inc_isA(ro.x);
                int* out = inc + sInc[b] + off;
                if (kind == FK_MANIFOLD) { const int nc = sEntRows[pos]; for (int c = 0; c < nc; ++c) out[c] = make_inc(isA, FK_MANIFOLD, idx * 4 + c); }
                else if (kind == FK_JOINT) { out[0] = make_inc(isA, FK_JOINT, idx * 2); out[1] = make_inc(isA, FK_JOINT, idx * 2 + 1); }
                else if (kind == FK_SPRING) { const int parts = gf[idx].kind == FK_HOST ? gforce_entries(gf[idx]) : 1; for (int q = 0; q < parts; ++q) out[q] = make_inc(isA, FK_SPRING, idx | (q << 24)); sBad = 1; }
                /* the key area is dead now (nothing below reads it): compact the neighbour lists into it, any order */
                if (ro.y > b && kind != FK_IGNORE && sSwept[ro.y]) sNbr[s0 + atomicAdd(&sCur[b], 1)] = (unsigned short)ro.y;
            }
        }
    } else {
        /* more entries than the LDS staging holds (dense packings): one thread per BODY, insertion sort in global memory */
        for (int b = t; b < n; b += 1024) {
            const int s0 = sAdj[b], s1 = sAdj[b + 1];
            if (sSwept[b]) {
                for (int i = s0 + 1; i < s1; ++i) {
                    const uint64_t k = adjKey[i]; const int2 ro = adjTmp[i];
                    int j = i - 1;
                    while (j >= s0 && adjKey[j] < k) { adjKey[j + 1] = adjKey[j]; adjTmp[j + 1] = adjTmp[j]; --j; }
                    adjKey[j + 1] = k; adjTmp[j + 1] = ro;
                }
            }
            int o = sInc[b];
            for (int i = s0; i < s1; ++i) {
                const int2 ro = adjTmp[i];
                adjRef[i] = ro.x; adjOther[i] = ro.y;
                if (!sSwept[b]) continue;
                const int kind = inc_kind(ro.x), idx = inc_index(ro.x), isA = inc_isA(ro.x);
                if (kind == FK_MANIFOLD) { const int nc = M.nc[idx]; for (int c = 0; c < nc; ++c) inc[o++] = make_inc(isA, FK_MANIFOLD, idx * 4 + c); }
                else if (kind == FK_JOINT) { inc[o++] = make_inc(isA, FK_JOINT, idx * 2); inc[o++] = make_inc(isA, FK_JOINT, idx * 2 + 1); }
                else if (kind == FK_SPRING) { const int parts = gf[idx].kind == FK_HOST ? gforce_entries(gf[idx]) : 1; for (int q = 0; q < parts; ++q) inc[o++] = make_inc(isA, FK_SPRING, idx | (q << 24)); sBad = 1; }
            }
            sCur[b] = s1 - s0;                                          /* the sweeps below filter the whole segment */
        }
    }
    for (int b = t; b < n; b += 1024)
        if (sSwept[b] && sInc[b + 1] - sInc[b] > 64) sBad = 1;         /* the resident schedule keeps one contact per lane */
        else if (sSwept[b] && sInc[b + 1] - sInc[b] > kResLiteMaxEntriesDecl) atomicOr(&cnt->residentBad, 4);
    __syncthreads();

#ifdef AVBD_DF_STATS
    __syncthreads(); if (t == 0) tstamp[4] = __builtin_amdgcn_s_memrealtime();
#endif
    if (byIndex) {
        /* data-flow schedules only need a linear extension of "later-created swept neighbours first": descending creation
         * index is one, and costs one scan instead of one relaxation sweep per level */
        /* (sweptKept >= 0: the host has seen no upload that changes the swept bodies since this branch last wrote level / order / lvlStart — they
         * are what they were, and that many bodies are swept) */
        int running = 0;
        for (int base = 0; sweptKept < 0 && base < n; base += 1024) {
            const int b = base + t;
            const int f = (b < n && sSwept[b]) ? 1 : 0;
            const int incl = (int)block_scan_inclusive_u64((unsigned long long)f, sWaveTotals);
            if (b < n) { level[b] = f; sLevel[b] = (unsigned short)(running + incl - 1); }   /* ascending rank, parked in sLevel */
            if (t == 1023) sTotal = incl;
            __syncthreads();
            running += sTotal;
            __syncthreads();
        }
        const int ndAll = sweptKept >= 0 ? sweptKept : running;
        if (sweptKept < 0) for (int b = t; b < n; b += 1024) if (sSwept[b]) order[ndAll - 1 - (int)sLevel[b]] = b;
        if (t == 0) {
            if (sweptKept < 0) { lvlStart[0] = 0; lvlStart[1] = ndAll; }
            atomicExch(&cnt->numLevels, ndAll > 0 ? 1 : 0);
            int bad = sBad ? 1 : 0;
            if (ndAll > residentWaves) bad |= 2;
            if (bad) atomicOr(&cnt->residentBad, bad);
        }
#ifdef AVBD_DF_STATS
        __syncthreads(); if (t == 0) tstamp[5] = __builtin_amdgcn_s_memrealtime();
#endif
        /* (kept: the host saw no upload that changes the swept bodies since the table was built, and the table is one — a step that
         * declined the one-body-per-wave launch leaves none — for this many bodies) */
        if (place && !(placedFor.w == 1 && placedFor.x == ndAll)) {
            /* one body per wave: which workgroup every body's wave sits in (avbd_place.h), on the sweep order just written and the
             * report of the last iteration launch; the staging area above is dead by now */
            __syncthreads();
            place_chains(order, ndAll, sBad != 0 || ndAll > residentWaves, placeCrit, placeTail, placeSlotOf, B.cap, placeW, place, placeInfo,
                         reinterpret_cast<char*>(sTopo), sWaveTotals, placeRounds);
        }
#ifdef AVBD_DF_STATS
        __syncthreads(); if (t == 0) tstamp[6] = __builtin_amdgcn_s_memrealtime();
#endif
        return;
    }
    /* ---- phase 4: dependency levels (level = 1 + max level of the later-created swept neighbours), relaxed in LDS ---- */
    constexpr int kPer = kTopoBodies / 1024, kNbReg = 8;
    int seg[kPer], cnt_[kPer];
    unsigned short nb[kPer][kNbReg];                                  /* the first neighbours of each body: registers, not LDS */
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int b = t + q * 1024;
        seg[q] = 0; cnt_[q] = -1;
        if (b < n && sSwept[b]) { seg[q] = sAdj[b]; cnt_[q] = sCur[b]; }
#pragma unroll
        for (int z = 0; z < kNbReg; ++z) nb[q][z] = (inLds && z < cnt_[q]) ? sNbr[seg[q] + z] : (unsigned short)0;
    }
    __shared__ int sChangedAlt[2];
    if (t == 0) { sChangedAlt[0] = 0; sChangedAlt[1] = 0; }
    __syncthreads();
    for (int sweep = 0; sweep <= n; ++sweep) {
        int changed = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            if (q * 1024 >= n) break;                                   /* wave-uniform: a 1000-body scene runs one of the four copies */
            if (cnt_[q] < 0) continue;
            const int b = t + q * 1024;
            int lv = 1;
            if (inLds) {
                int lvz[kNbReg];
#pragma unroll
                for (int z = 0; z < kNbReg; ++z) lvz[z] = (int)sLevel[nb[q][z]];             /* independent reads */
#pragma unroll
                for (int z = 0; z < kNbReg; ++z) if (z < cnt_[q] && lvz[z] + 1 > lv) lv = lvz[z] + 1;
                for (int z = kNbReg; z < cnt_[q]; ++z) { int lo2 = (int)sLevel[sNbr[seg[q] + z]] + 1; if (lo2 > lv) lv = lo2; }
            } else {
                for (int z = 0; z < cnt_[q]; ++z) {
                    const int2 ro = adjTmp[seg[q] + z];
                    if (ro.y > b && inc_kind(ro.x) != FK_IGNORE && sSwept[ro.y]) { int lo2 = (int)sLevel[ro.y] + 1; if (lo2 > lv) lv = lo2; }
                }
            }
            if (lv != (int)sLevel[b]) { sLevel[b] = (unsigned short)lv; changed = 1; }
        }
        if (changed) sChangedAlt[sweep & 1] = 1;
        if (t == 0) sChangedAlt[(sweep + 1) & 1] = 0;                  /* the flag of the next sweep; nobody reads or sets it now */
        __syncthreads();
        if (!sChangedAlt[sweep & 1]) break;
        __syncthreads();
    }

#ifdef AVBD_DF_STATS
    __syncthreads(); if (t == 0) tstamp[5] = __builtin_amdgcn_s_memrealtime();
#endif
    /* ---- phase 5: bodies grouped by level (counting sort; levels of a small scene fit one histogram word per thread) ---- */
    if (t == 0) sMax = 0;
    sHist[t] = 0;
    __syncthreads();
    int mx = 0;
    for (int b = t; b < n; b += 1024) {
        const int lv = sLevel[b];
        level[b] = lv;
        if (lv > 0) { if (lv < 1024) atomicAdd(&sHist[lv], 1); if (lv > mx) mx = lv; }
    }
    atomicMax(&sMax, mx);
    __syncthreads();
    const int L = sMax;                                               /* levels are 1..L */
    if (L >= 1024 || L >= lvlCap) { if (t == 0) { atomicExch(&cnt->numLevels, 0); atomicOr(&cnt->overflow, 8); } return; }
    /* exclusive scan of sHist[1..L]: thread t owns level t+1 */
    const int hv = (t < L) ? sHist[t + 1] : 0;
    const int incl = (int)block_scan_inclusive_u64((unsigned long long)hv, sWaveTotals);
    const int startL = incl - hv;
    if (t < L) { lvlStart[t] = startL; sHist[t + 1] = startL; }       /* sHist becomes the fill cursor of its level */
    if (t == 1023) { lvlStart[L] = incl; sTotal = incl; }
    __syncthreads();
    for (int b = t; b < n; b += 1024) {
        const int lv = sLevel[b];
        if (lv > 0) order[atomicAdd(&sHist[lv], 1)] = b;
    }
#ifdef AVBD_DF_STATS
    __syncthreads(); if (t == 0) tstamp[6] = __builtin_amdgcn_s_memrealtime();
#endif
    if (t == 0) {
        atomicExch(&cnt->numLevels, L);
        int bad = sBad ? 1 : 0;
        if (sTotal > residentWaves) bad |= 2;
        if (bad) atomicOr(&cnt->residentBad, bad);
    }
}

/* K5 primal, K6 dual and the fused data-flow iteration kernel live in avbd_primal.h / avbd_iterate.h */
} // namespace avbd
#include "avbd_primal.h"
#include "avbd_iterate.h"
#include "avbd_groups.h"
#include "avbd_cuts.h"
namespace avbd {
static_assert(kResLiteMaxEntriesDecl == kResLiteMaxEntries, "the topology kernels and the resident schedule must agree on the short product lines");

/* ------------------------------------------------------------------------------------------
 * K7 velocities — solver.cpp:434-469
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ void velocity_body(const BodyTable& B, int i, const SolverParams& P, Counters* cnt, float& speedLin, float& speedAng)
{
    if (!(B.plane(BP_MASS)[i].y > 0.0f)) return;
    const float linearDamping = 0.995f, angularDamping = 0.97f;
    f3 v = ld3(B.plane(BP_LINV), i), w = ld3(B.plane(BP_ANGV), i);
    B.plane(BP_PLIN)[i] = pack3(v, 0.f);
    B.plane(BP_PANG)[i] = pack3(w, 0.f);
    f3 pos = ld3(B.plane(BP_POSR), i), ipos = ld3(B.plane(BP_IPOS), i);
    q4 rot = ldq(B.plane(BP_QUAT), i), irot = ldq(B.plane(BP_IQUAT), i);
    v = (pos - ipos) / P.dt;
    q4 dq = rot * qconj(irot);
    f3 av = mk3(dq.x, dq.y, dq.z);
    av = av * (2.0f / P.dt);
    if (dq.w < 0.0f) av = -av;
    w = av;
    v = v * linearDamping;
    w = w * angularDamping;
    sanitize3(v, cnt, warn_key(2, 0, i, 0), i, WK_LINVEL);
    sanitize3(w, cnt, warn_key(2, 0, i, 1), i, WK_ANGVEL);
    B.plane(BP_LINV)[i] = pack3(v, 0.f);
    B.plane(BP_ANGV)[i] = pack3(w, 0.f);
    speedLin = len3(v);                                              /* solver.cpp:472-497: maxima over the dynamic bodies */
    speedAng = len3(w);
}

/* ------------------------------------------------------------------------------------------
 * K8 diagnostics — solver.cpp:472-497 (max and integer sums only: order-independent)
 * ------------------------------------------------------------------------------------------ */
/* per-thread partial results, reduced over the wavefront with shuffles: one set of atomics per wave instead of five per contact
 * (480 000 contact slots at 64 000 cubes) */
__device__ __forceinline__ void diag_contacts(const BodyTable& B, const ManifoldTable& M, const Counters* cnt, DiagAccum* diag, int first, int stride)
{
    const int nm = cnt->numManifolds;
    const int total = nm * 4;
    float maxPen = 0.0f, maxViol = 0.0f, maxImp = 0.0f;
    int manifolds = 0, contacts = 0;
    bool nan = false;
    for (int s = first; s < total; s += stride) {
        const int m = s >> 2, ci = s & 3;
        const int nc = M.nc[m];
        if (ci == 0) { ++manifolds; contacts += nc; }
        if (ci >= nc) continue;
        const int2 pr = M.pair[m];
        float4 ra = M.cRA[s], rb = M.cRB[s], nn = M.cN[s];
        f3 wrA = qrotate(ldq(B.plane(BP_QUAT), pr.x), mk3(ra.x, ra.y, ra.z));
        f3 wrB = qrotate(ldq(B.plane(BP_QUAT), pr.y), mk3(rb.x, rb.y, rb.z));
        f3 pA = ld3(B.plane(BP_POSR), pr.x) + wrA;
        f3 pB = ld3(B.plane(BP_POSR), pr.y) + wrB;
        float separation = dot3(pA - pB, mk3(nn.x, nn.y, nn.z));
        float pen = rmax(0.0f, -separation);
        float viol = rmax(0.0f, kPenetrationSlop - separation);
        float imp = fabsf(M.cLam[s].x);
        /* NaNs are skipped like in atomic_max_nonneg; all three quantities are >= 0 */
        if (pen == pen) maxPen = pen > maxPen ? pen : maxPen;
        if (viol == viol) maxViol = viol > maxViol ? viol : maxViol;
        if (imp == imp) maxImp = imp > maxImp ? imp : maxImp;
        (void)nan;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float p2 = __shfl_xor(maxPen, o), v2 = __shfl_xor(maxViol, o), i2 = __shfl_xor(maxImp, o);
        maxPen = p2 > maxPen ? p2 : maxPen; maxViol = v2 > maxViol ? v2 : maxViol; maxImp = i2 > maxImp ? i2 : maxImp;
        manifolds += __shfl_xor(manifolds, o); contacts += __shfl_xor(contacts, o);
    }
    /* one set of atomics per workgroup: the five words share a cache line, and an atomic on one line costs about 13 ns whoever
     * issues it (one set per wave was 45 000 atomics = 0.33 ms at 64 000 cubes) */
    __shared__ float sMax[3][4];
    __shared__ int sCount[2][4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sMax[0][wv] = maxPen; sMax[1][wv] = maxViol; sMax[2][wv] = maxImp; sCount[0][wv] = manifolds; sCount[1][wv] = contacts; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)(blockDim.x >> 6);
        for (int k = 1; k < nw; ++k) {
            maxPen = sMax[0][k] > maxPen ? sMax[0][k] : maxPen; maxViol = sMax[1][k] > maxViol ? sMax[1][k] : maxViol; maxImp = sMax[2][k] > maxImp ? sMax[2][k] : maxImp;
            manifolds += sCount[0][k]; contacts += sCount[1][k];
        }
        if (manifolds) { atomicAdd(&diag->activeManifolds, manifolds); atomicAdd(&diag->activeContacts, contacts); }
        if (maxPen > 0.0f) atomic_max_nonneg(&diag->maxPenetration, maxPen);
        if (maxViol > 0.0f) atomic_max_nonneg(&diag->maxViolation, maxViol);
        if (maxImp > 0.0f) atomic_max_nonneg(&diag->maxNormalImpulse, maxImp);
    }
}

/* velocities (first `velBlocks` workgroups), contact diagnostics (the rest) and the dynamic-body count in one launch */
__global__ void k_tail(BodyTable B, ManifoldTable M, SolverParams P, Counters* cnt, DiagAccum* diag, int velBlocks)
{
    if ((int)blockIdx.x < velBlocks) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        /* counted by the predictor, long finished; an atomic because the other workgroups of this launch update the same
         * cache line with atomics from other XCDs (a plain store's write-back could put stale neighbours over them) */
        if (i == 0) atomicExch(&diag->dynamicBodies, cnt->dynamicBodies);
        float sl = 0.0f, sa = 0.0f;
        if (i < B.n) velocity_body(B, i, P, cnt, sl, sa);
        if (!(sl == sl)) sl = 0.0f;                                   /* NaNs are skipped, like in atomic_max_nonneg */
        if (!(sa == sa)) sa = 0.0f;
        for (int o = 32; o > 0; o >>= 1) {
            const float l2 = __shfl_xor(sl, o), a2 = __shfl_xor(sa, o);
            sl = l2 > sl ? l2 : sl; sa = a2 > sa ? a2 : sa;
        }
        __shared__ float sSpeed[2][4];
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sSpeed[0][wv] = sl; sSpeed[1][wv] = sa; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < (int)(blockDim.x >> 6); ++k) { sl = sSpeed[0][k] > sl ? sSpeed[0][k] : sl; sa = sSpeed[1][k] > sa ? sSpeed[1][k] : sa; }
            if (sl > 0.0f) atomic_max_nonneg(&diag->maxLinearSpeed, sl);
            if (sa > 0.0f) atomic_max_nonneg(&diag->maxAngularSpeed, sa);
        }
    } else {
        diag_contacts(B, M, cnt, diag, (blockIdx.x - velBlocks) * blockDim.x + threadIdx.x, (gridDim.x - velBlocks) * blockDim.x);
    }
}

/* start-of-step reset of the per-step counters */
__global__ void k_begin_step(Counters* cnt, DiagAccum* diag, int* nextCount, int* cellHead, int buckets, int* narrowCntNext)
{
    /* grid scenes: the broadphase's bucket heads start empty (-1); one launch instead of a fill behind it */
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < buckets; i += gridDim.x * blockDim.x) cellHead[i] = -1;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    cnt->numCandidates = 0; cnt->numContacts = 0; cnt->numIncidences = 0; cnt->warnings = 0; cnt->dynamicBodies = 0;
    cnt->numLevels = 0; cnt->residentBad = 0;
    *nextCount = 0;
    if (narrowCntNext) for (int k = 0; k < kNarrowClasses; ++k) narrowCntNext[k * kNarrowCntStride] = 0;
    diag->maxPenetration = 0; diag->maxViolation = 0; diag->maxLinearSpeed = 0; diag->maxAngularSpeed = 0; diag->maxNormalImpulse = 0;
    diag->activeContacts = 0; diag->activeManifolds = 0; diag->dynamicBodies = 0;
}

/* ---- sharded runs: exchange of body poses between GPUs once per iteration --------------------------------
 * perm lists the solved bodies rank by rank (rank r owns perm[rankStart[r] .. rankStart[r+1])); every rank
 * contributes maxCount rows of {position+radius, orientation} (2 float4 = 32 B per body) to the all-gather. */
__global__ void k_pack_poses(BodyTable B, const int* perm, int first, int count, float4* send)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int b = perm[first + k];
    send[2 * k] = B.plane(BP_POSR)[b];
    send[2 * k + 1] = B.plane(BP_QUAT)[b];
}

__global__ void k_unpack_poses(BodyTable B, const int* perm, const int* rankStart, int nranks, int myRank, int maxCount, const float4* recv)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;
    if (r == myRank) return;
    const int first = rankStart[r], count = rankStart[r + 1] - first;
    if (k >= count) return;
    const int b = perm[first + k];
    B.plane(BP_POSR)[b] = recv[2 * ((size_t)r * maxCount + k)];
    B.plane(BP_QUAT)[b] = recv[2 * ((size_t)r * maxCount + k) + 1];
}

/* the velocity half of the 13-float exchange (once per step, after K7): linear + angular velocity, 2 float4 per body like the poses */
__global__ void k_pack_velocities(BodyTable B, const int* perm, int first, int count, float4* send)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const int b = perm[first + k];
    send[2 * k] = B.plane(BP_LINV)[b];
    send[2 * k + 1] = B.plane(BP_ANGV)[b];
}

/* every rank has re-derived these velocities from the gathered poses already: count the words that differ, then store */
__global__ void k_unpack_velocities(BodyTable B, const int* perm, const int* rankStart, int nranks, int myRank, int maxCount, const float4* recv, unsigned* mismatches)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;
    if (r == myRank) return;
    const int first = rankStart[r], count = rankStart[r + 1] - first;
    if (k >= count) return;
    const int b = perm[first + k];
    const float4 v = recv[2 * ((size_t)r * maxCount + k)], w = recv[2 * ((size_t)r * maxCount + k) + 1];
    const float4 v0 = B.plane(BP_LINV)[b], w0 = B.plane(BP_ANGV)[b];
    const unsigned bad = (__float_as_uint(v.x) != __float_as_uint(v0.x)) + (__float_as_uint(v.y) != __float_as_uint(v0.y)) + (__float_as_uint(v.z) != __float_as_uint(v0.z))
                       + (__float_as_uint(w.x) != __float_as_uint(w0.x)) + (__float_as_uint(w.y) != __float_as_uint(w0.y)) + (__float_as_uint(w.z) != __float_as_uint(w0.z));
    if (bad) atomicAdd(mismatches, bad);
    const unsigned long long act = __ballot(1);                      /* rows compared, one atomic per wavefront */
    if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(mismatches + 1, (unsigned)__popcll(act));
    B.plane(BP_LINV)[b] = v;
    B.plane(BP_ANGV)[b] = w;
}

/* ---- debug race checker (AVBD_RACE_CHECK=1) --------------------------------------------------------
 * The schedules are race-free by construction; this kernel checks the construction, every step, from the topology the step
 * will run on:
 *   level schedule (one launch per level): no two bodies of one launch share a force that has rows — a launch would read a
 *     pose / write a manifold that another wavefront of the same launch writes;
 *   data-flow schedules: the sweep order is a linear extension of "later-created swept neighbours first" (what makes the
 *     smallest unfinished task runnable, hence the persistent launches deadlock-free), lists a swept body exactly once, and
 *     every incidence entry of a body refers to a force that body is an endpoint of.
 * A violation sets bit 5 of Counters.overflow; the host turns it into an error at the next synchronising call. */
__global__ void k_race_check(BodyTable B, ManifoldTable M, const GForce* gf, const Counters* cnt, Counters* cntW, const int* __restrict__ adjStart,
                             const int* __restrict__ adjRef, const int* __restrict__ adjOther, const int* __restrict__ incStart, const int* __restrict__ inc,
                             const int* __restrict__ level, const int* __restrict__ order, const int* __restrict__ lvlStart, int* __restrict__ slotOf, int byLevel, int pass)
{
    const int nd = lvlStart[cnt->numLevels];
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nd; s += gridDim.x * blockDim.x) {
        const int b = order[s];
        bool bad = b < 0 || b >= B.n || !B.solved[b];
        if (pass == 0) { if (!bad) slotOf[b] = s; else atomicOr(&cntW->overflow, 32); continue; }
        if (bad || slotOf[b] != s) { atomicOr(&cntW->overflow, 32); continue; }            /* listed twice: the later slot won */
        for (int k = adjStart[b]; k < adjStart[b + 1]; ++k) {
            const int o = adjOther[k];
            if (o < 0 || inc_kind(adjRef[k]) == FK_IGNORE || !B.solved[o]) continue;
            if (byLevel) { if (level[o] == level[b]) bad = true; }
            else if (o > b && !(slotOf[o] < s)) bad = true;
        }
        for (int e = incStart[b]; e < incStart[b + 1]; ++e) {
            const int ent = inc[e], kind = inc_kind(ent), idx = inc_index(ent);
            int a = -2, c2 = -2;
            if (kind == FK_MANIFOLD) { const int2 pr = M.pair[idx >> 2]; a = pr.x; c2 = pr.y; if ((idx & 3) >= M.nc[idx >> 2]) bad = true; }
            else if (kind == FK_JOINT) { a = gf[idx >> 1].bodyA; c2 = gf[idx >> 1].bodyB; }
            else { a = gf[idx & 0xFFFFFF].bodyA; c2 = gf[idx & 0xFFFFFF].bodyB; }
            if ((inc_isA(ent) ? a : c2) != b) bad = true;
        }
        if (bad) atomicOr(&cntW->overflow, 32);
    }
}

/* ---- known-answer kernels --------------------------------------------------------------- */
struct KatContact { int feature; float rA[3], rB[3], normal[3]; float penetration, C0_n, C0_t[3]; int stick; };

__global__ __launch_bounds__(64) void k_kat_collide(int n, const float* pA, const float* qA, const float* sA, const float* pB, const float* qB, const float* sB,
                                                    KatContact* out, int* counts)
{
    /* the cooperative narrow phase the solver runs (avbd_collide_coop.h): four pairs per wavefront, sixteen lanes per pair */
    __shared__ CoopScratch sWs[kPairsPerWave];
    const int gl = threadIdx.x & 15, gBase = threadIdx.x & 48, gj = threadIdx.x >> 4;
    const int i = blockIdx.x * kPairsPerWave + gj;
    if (i >= n) return;                                               /* uniform over the group */
    BoxPose A, Bx;
    A.pos = mk3(pA[3 * i], pA[3 * i + 1], pA[3 * i + 2]); A.rot = mkq(qA[4 * i], qA[4 * i + 1], qA[4 * i + 2], qA[4 * i + 3]);
    A.size = mk3(sA[3 * i], sA[3 * i + 1], sA[3 * i + 2]);
    Bx.pos = mk3(pB[3 * i], pB[3 * i + 1], pB[3 * i + 2]); Bx.rot = mkq(qB[4 * i], qB[4 * i + 1], qB[4 * i + 2], qB[4 * i + 3]);
    Bx.size = mk3(sB[3 * i], sB[3 * i + 1], sB[3 * i + 2]);
    CoopScratch& ws = sWs[gj];
    const int nc = collide_boxes_coop(A, Bx, ws, gl, gBase);
    if (gl == 0) counts[i] = nc;
    if (gl < nc) {
        const RawContact& r = ws.raw[gl];
        KatContact& o = out[4 * i + gl];
        o.feature = r.feature;
        o.rA[0] = r.rA.x; o.rA[1] = r.rA.y; o.rA[2] = r.rA.z;
        o.rB[0] = r.rB.x; o.rB[1] = r.rB.y; o.rB[2] = r.rB.z;
        o.normal[0] = r.normal.x; o.normal[1] = r.normal.y; o.normal[2] = r.normal.z;
        o.penetration = r.penetration; o.C0_n = 0.f; o.C0_t[0] = o.C0_t[1] = o.C0_t[2] = 0.f; o.stick = 0;
    }
}

__global__ void k_kat_solve6(int n, const float* A, const float* b, float* x)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = A + 36 * i;
    m33 m[4];
    for (int k = 0; k < 4; ++k)
        m[k] = mkm(mk3(a[9 * k], a[9 * k + 1], a[9 * k + 2]), mk3(a[9 * k + 3], a[9 * k + 4], a[9 * k + 5]), mk3(a[9 * k + 6], a[9 * k + 7], a[9 * k + 8]));
    f3 xl, xa;
    solve_block6(m[0], m[1], m[2], m[3], mk3(b[6 * i], b[6 * i + 1], b[6 * i + 2]), mk3(b[6 * i + 3], b[6 * i + 4], b[6 * i + 5]), xl, xa);
    x[6 * i] = xl.x; x[6 * i + 1] = xl.y; x[6 * i + 2] = xl.z; x[6 * i + 3] = xa.x; x[6 * i + 4] = xa.y; x[6 * i + 5] = xa.z;
}

/* Solver::pick (solver.cpp:145-228): nearest dynamic box hit by a ray.  One workgroup per ray; each lane runs the reference's
 * slab test on every 256th body, the workgroup keeps the smallest (t, newest body) — the reference walks its list newest
 * first and replaces only on a strictly smaller t — and the lane that holds it writes the body-local hit point. */
__global__ void __launch_bounds__(256) k_pick(BodyTable B, int nrays, const float* origins, const float* dirs, int* bodyOut, float* localOut)
{
    const int ray = blockIdx.x;
    if (ray >= nrays) return;
    __shared__ unsigned long long sKey[256];
    const float epsilon = 1.0e-6f;
    const f3 origin = mk3(origins[3 * ray], origins[3 * ray + 1], origins[3 * ray + 2]);
    f3 rayDir = mk3(dirs[3 * ray], dirs[3 * ray + 1], dirs[3 * ray + 2]);
    const float dirLenSq = lensq3(rayDir);
    unsigned long long best = ~0ull;
    f3 bestLocal = zero3();
    if (!(dirLenSq < epsilon)) {
        rayDir = rayDir / sqrtf(dirLenSq);
        const float4* posr = B.plane(BP_POSR); const float4* quat = B.plane(BP_QUAT);
        const float4* size = B.plane(BP_SIZE); const float4* mass = B.plane(BP_MASS);
        for (int i = threadIdx.x; i < B.n; i += 256) {
            if (mass[i].y <= 0.0f) continue;
            const float4 p = posr[i], q = quat[i], s = size[i];
            const q4 invRot = qconj(mkq(q.x, q.y, q.z, q.w));
            const f3 lo = qrotate(invRot, origin - mk3(p.x, p.y, p.z));
            const f3 ld = qrotate(invRot, rayDir);
            const f3 half = mk3(s.x, s.y, s.z) * 0.5f;
            float tEnter = 0.0f, tExit = FLT_MAX;
            bool hit = true;
            for (int axis = 0; axis < 3 && hit; ++axis) {
                const float o = comp3(lo, axis), d = comp3(ld, axis);
                const float minB = -comp3(half, axis), maxB = comp3(half, axis);
                if (fabsf(d) < epsilon) {
                    if (o < minB || o > maxB) hit = false;
                    continue;
                }
                const float invD = 1.0f / d;
                float t0 = (minB - o) * invD, t1 = (maxB - o) * invD;
                if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
                tEnter = rmax(tEnter, t0);
                tExit = rmin(tExit, t1);
                if (tEnter > tExit) hit = false;
            }
            if (!hit) continue;
            const float tHit = (tEnter >= 0.0f) ? tEnter : tExit;
            if (tHit < 0.0f || !(tHit < FLT_MAX)) continue;
            const unsigned long long key = ((unsigned long long)__float_as_uint(tHit + 0.0f) << 32) | (unsigned)(0x7fffffff - i);
            if (key < best) { best = key; bestLocal = lo + ld * tHit; }
        }
    }
    sKey[threadIdx.x] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && sKey[threadIdx.x + s] < sKey[threadIdx.x]) sKey[threadIdx.x] = sKey[threadIdx.x + s];
        __syncthreads();
    }
    const unsigned long long win = sKey[0];
    if (win == ~0ull) {
        if (threadIdx.x == 0) bodyOut[ray] = -1;
    } else if (best == win) {
        bodyOut[ray] = 0x7fffffff - (int)(unsigned)(win & 0xffffffffu);
        localOut[3 * ray] = bestLocal.x; localOut[3 * ray + 1] = bestLocal.y; localOut[3 * ray + 2] = bestLocal.z;
    }
}

__global__ void k_kat_arith(int n, const float* a, const float* b, float* od, float* os, float* om, float* oa)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    od[i] = a[i] / b[i];
    os[i] = sqrtf(fabsf(a[i]));
    om[i] = a[i] * b[i] + a[i]; /* must round twice (no FMA contraction) */
    oa[i] = 1.0f / sqrtf(fabsf(b[i]));
}

} // namespace avbd
