/*
 * avbd_place.h — which workgroup a body's wavefront sits in (one body per wave, k_iterate_single).
 *
 * Placement never changes a result: every wave runs the same arithmetic on the same records whatever CU it is on.  It changes
 * the latency of the hand-offs: a pose that travels through LDS reaches its consumer in ~0.1 us, through memory in ~0.8 us,
 * and Stress1000's step is a chain of ~200 such hand-offs.  So the bodies that wait for each other are put into
 * the same workgroup, four at a time:
 *
 *   - every wave reports the later-created partner whose pose its last waiting task received LAST (ResidentExchange.crit):
 *     an edge "pred -> body" of the graph of critical hand-offs (acyclic: pred is created later, i.e. swept earlier);
 *   - tail[b] = length of the longest chain of such edges that starts at b, relaxed by one round per step (the graph changes
 *     slowly; the value persists in memory between steps);
 *   - a body that several others wait for keeps the successor with the longest tail (ties: the one that finished last): the
 *     edges that are left form disjoint paths, which are cut into groups of kIterWaves from their heads;
 *   - groups take whole workgroups; bodies without a group fill the slots that are left, spread over the workgroups in sweep
 *     order like the unplaced launch does.
 *
 * One workgroup of 1024 threads, thread t = slot t of the sweep order.  Runs between the topology kernel and the iteration
 * launch of a step, on the report of the previous step.
 */
#pragma once

namespace avbd {

struct PlaceInfo { int nd, groups, grouped, valid; };

constexpr int kPlaceWaves = 4;                                     /* = kIterWaves (avbd_iterate.h, included later) */
constexpr int kPlaceLdsBytes = 8192 + 6 * 4096 + 3 * 1024 + 256;   /* working storage of place_chains */

/* the work of k_place_chains for one workgroup of 1024 threads; `lds` is kPlaceLdsBytes of 8-byte aligned LDS nobody else uses */
__device__ __forceinline__ void place_chains(const int* __restrict__ order, int nd, bool declined, const int2* __restrict__ crit, int* __restrict__ tail,
                                             int* __restrict__ slotOf, int bodyCap, int W, int* __restrict__ place, PlaceInfo* info, char* lds,
                                             unsigned long long* sWaveTotals /* [16] */, int rounds = 1)
{
    unsigned long long* sBest = reinterpret_cast<unsigned long long*>(lds);
    int* sTail = reinterpret_cast<int*>(lds + 8192);
    int* sNext = sTail + 1024; int* sLead = sNext + 1024; int* sFree = sLead + 1024; int* sGroupWg = sFree + 1024; int* sMisc = sGroupWg + 1024;
    unsigned char* sAccepted = reinterpret_cast<unsigned char*>(sMisc + 1024);
    unsigned char* sRank = sAccepted + 1024; unsigned char* sSize = sRank + 1024; unsigned char* sWgSize = sSize + 1024;
    const int t = threadIdx.x;
    const int WG = W / kPlaceWaves;
    if (nd <= 0 || nd > W || W > 1024 || declined) {                    /* nothing to place (the launch that would use the table declines too) */
        for (int s = t; s < W; s += 1024) place[s] = -1;
        if (t == 0) { info->nd = nd; info->groups = 0; info->grouped = 0; info->valid = 0; }
        return;
    }
    const int body = t < nd ? order[t] : -1;
    if (body >= 0) slotOf[body] = t;
    sBest[t] = 0ull; sTail[t] = 1; sFree[t] = 1; sGroupWg[t] = 1; sAccepted[t] = 0; sLead[t] = t; sRank[t] = 0; sSize[t] = 1;
    for (int s = t; s < W; s += 1024) place[s] = -1;
    __syncthreads();

    /* the reported edge, if it still exists: the partner is swept this step and was created later */
    int ps = -1; unsigned fin = 0;
    if (body >= 0) {
        const int2 c = crit[body];
        fin = (unsigned)c.y;
        if (c.x > body && c.x < bodyCap) { const int s = slotOf[c.x]; if (s >= 0 && s < nd && order[s] == c.x) ps = s; }
    }
    /* `rounds` rounds of tail[b] = 1 + max over the bodies that wait for b (a table that is kept for several steps relaxes that many
     * rounds when it is rebuilt; three buffers in turn: the one after next is reset while this round's is filled) */
    int myTail = body >= 0 ? max(1, min(tail[body], 62)) : 1;
    for (int r = 0; r < rounds; ++r) {
        int* const dst = r % 3 == 0 ? sTail : (r % 3 == 1 ? sFree : sGroupWg);
        int* const afterNext = r % 3 == 0 ? sGroupWg : (r % 3 == 1 ? sTail : sFree);
        if (ps >= 0) atomicMax(&dst[ps], myTail + 1);
        afterNext[t] = 1;
        __syncthreads();
        myTail = r + 1 < rounds ? max(1, min(dst[t], 62)) : dst[t];
    }
    if (body >= 0) tail[body] = myTail;
    /* a body keeps the successor with the longest tail */
    if (ps >= 0) atomicMax(&sBest[ps], ((unsigned long long)myTail << 44) | ((unsigned long long)min(fin, 0xffffffu) << 12) | (unsigned long long)t);
    __syncthreads();
    const int nxt = sBest[t] ? (int)(sBest[t] & 0xfffull) : -1;
    sNext[t] = nxt;
    if (nxt >= 0) sAccepted[nxt] = 1;
    __syncthreads();
    /* heads walk their path and cut it into groups; the groups of ONE path go to workgroups of ONE XCD (workgroup g runs on XCD g % 8: a hand-off
     * between two groups of a path then stays inside that XCD's L2, 0.88 instead of 0.98 us), paths to the XCDs in turn */
    const bool isHead = body >= 0 && nxt >= 0 && !sAccepted[t];
    const int pathIdx = (int)block_scan_inclusive_u64(isHead ? 1ull : 0ull, sWaveTotals) - 1;
    if (isHead) {
        int leader = t, n = 0;
        for (int k = t; k >= 0; k = sNext[k]) {
            if (n == kPlaceWaves) { leader = k; n = 0; }
            sLead[k] = leader; sRank[k] = (unsigned char)n; sSize[leader] = (unsigned char)(n + 1);
            if (n == 0) sMisc[leader] = pathIdx & 7;
            ++n;
        }
    }
    __syncthreads();
    /* groups of two and more take whole workgroups (at most one group per workgroup), on their path's XCD in sweep order of their leaders */
    const bool leads = body >= 0 && sLead[t] == t && sSize[t] >= 2;
    int gidx;
    if (WG % 8 == 0 && WG <= 256) {
        const int xcd = leads ? sMisc[t] : 0;
        const unsigned long long one = leads ? 1ull << (16 * (xcd & 3)) : 0ull;
        const unsigned long long lo = block_scan_inclusive_u64(xcd < 4 ? one : 0ull, sWaveTotals);
        const unsigned long long hi = block_scan_inclusive_u64(xcd >= 4 ? one : 0ull, sWaveTotals);
        const int rank = (int)(((xcd < 4 ? lo : hi) >> (16 * (xcd & 3))) & 0xffffull) - 1;
        gidx = (leads && rank < WG / 8) ? xcd + 8 * rank : WG;          /* (an XCD that is full: the group's bodies are placed one by one below) */
    } else {
        gidx = (int)block_scan_inclusive_u64(leads ? 1ull : 0ull, sWaveTotals) - (leads ? 1 : 0);
    }
    sGroupWg[t] = (leads && gidx < WG && gidx < 256) ? gidx : -1;
    if (t < 256) sWgSize[t] = 0;
    __syncthreads();
    if (sGroupWg[t] >= 0) sWgSize[sGroupWg[t]] = sSize[t];
    const int myWg = body >= 0 ? sGroupWg[sLead[t]] : -1;
    const bool single = body >= 0 && myWg < 0;
    __syncthreads();
    if (myWg >= 0) place[myWg * kPlaceWaves + sRank[t]] = body;
    /* the slots that are left, wave-major (the waves of a workgroup take bodies a quarter of the sweep apart) */
    int slot = -1; bool isFree = false;
    if (t < W) { const int wv = t / WG, wg = t % WG; slot = wg * kPlaceWaves + wv; isFree = wg >= 256 || wv >= (int)sWgSize[wg]; }
    const unsigned long long mine = (isFree ? 1ull : 0ull) | (single ? (1ull << 32) : 0ull);
    const unsigned long long incl = block_scan_inclusive_u64(mine, sWaveTotals);
    const int freeRank = (int)(unsigned)(incl - mine), singleRank = (int)((incl - mine) >> 32);
    if (isFree) sFree[freeRank] = slot;
    __syncthreads();
    if (single) place[sFree[singleRank]] = body;
    const int placedGroups = (int)block_scan_inclusive_u64(sGroupWg[t] >= 0 && leads ? 1ull : 0ull, sWaveTotals);
    if (t == 1023) { info->nd = nd; info->groups = placedGroups; info->grouped = nd - (int)(incl >> 32); info->valid = 1; }
}

/* stand-alone launch (scenes whose topology is not built by the one-workgroup kernel, which otherwise runs place_chains itself) */
__global__ __launch_bounds__(1024) void k_place_chains(const int* __restrict__ order, const int* __restrict__ lvlStart, const Counters* __restrict__ cnt,
                                                       const int2* __restrict__ crit, int* __restrict__ tail, int* __restrict__ slotOf, int bodyCap,
                                                       int W, int* __restrict__ place, PlaceInfo* info)
{
    __shared__ unsigned long long sWaveTotals[16];
    __shared__ __attribute__((aligned(8))) char sLds[kPlaceLdsBytes];
    place_chains(order, lvlStart[cnt->numLevels], cnt->residentBad != 0, crit, tail, slotOf, bodyCap, W, place, info, sLds, sWaveTotals);
}

} // namespace avbd
