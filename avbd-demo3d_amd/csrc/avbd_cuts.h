/*
 * avbd_cuts.h — exact cuts between the ranks of a sharded scene (ShardExchange in avbd_groups.h).
 *
 * Every rank builds the manifolds that touch a body it sweeps, so a manifold between bodies of two ranks exists on both — from
 * the same poses, hence with the same contacts.  Its visit / dual records must sit where BOTH copies look for them.  The slot
 * needs no communication: both ranks hold the same set of manifolds of their cut, so "number of manifolds of this cut with a
 * smaller (bodyA, bodyB)" is the same number on both (k_cut_rank).  Records of all other contacts stay in the rank's own range.
 *
 *   record index space (contacts): [rank r: r * L, (r + 1) * L)  [cut of ranks p < q: N * L + (p * N + q) * C, ... + C)
 *
 * k_cut_fill then writes, per incidence entry of a swept body, the record index of its contact (bit 31: the contact's owner is
 * swept by another rank).  k_unpack_records: after ALL ranks' loops, the final poses of the bodies other ranks sweep, from their
 * pose records, into this rank's body table (one copy per step instead of one all-gather per iteration).
 */
#pragma once

namespace avbd {

struct CutCounters { int count, overflow, pad0, pad1; };

__global__ void k_cut_list(ManifoldTable M, const Counters* __restrict__ cnt, const int* __restrict__ owner, int nranks, unsigned long long* keys, int* man, int cap,
                           int* cutOf, CutCounters* cc)
{
    const int nm = cnt->numManifolds;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < nm; m += gridDim.x * blockDim.x) {
        const int2 pr = M.pair[m];
        const int ra = owner[pr.x], rb = owner[pr.y];
        cutOf[m] = -1;
        if (ra >= 0 && rb >= 0 && ra != rb) {
            const int pos = atomicAdd(&cc->count, 1);
            if (pos < cap) {
                const unsigned long long region = (unsigned long long)(min(ra, rb) * nranks + max(ra, rb));
                keys[pos] = (region << 48) | ((unsigned long long)(unsigned)pr.x << 24) | (unsigned long long)(unsigned)pr.y;
                man[pos] = m;
            } else atomicOr(&cc->overflow, 1);
        }
    }
}

__global__ void k_cut_rank(const unsigned long long* __restrict__ keys, const int* __restrict__ man, int cap, int* cutOf, CutCounters* cc, int slotsPerCut)
{
    const int n = min(cc->count, cap);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) { const unsigned long long o = keys[j]; rank += ((o >> 48) == (k >> 48) && o < k) ? 1 : 0; }
        if (rank >= slotsPerCut) { atomicOr(&cc->overflow, 2); rank = 0; }
        cutOf[man[i]] = rank;
    }
}

__global__ void k_cut_fill(BodyTable B, ManifoldTable M, const int* __restrict__ order, const int* __restrict__ lvlStart, const Counters* __restrict__ cnt,
                           const int* __restrict__ incStart, const int* __restrict__ inc, const int* __restrict__ owner, int nranks, int rank,
                           const int* __restrict__ cutOf, int localSlots, int slotsPerCut, int* recIdx)
{
    const int nd = lvlStart[cnt->numLevels];
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nd; s += gridDim.x * blockDim.x) {
        const int b = order[s];
        for (int e = incStart[b]; e < incStart[b + 1]; ++e) {
            const int ent = inc[e], idx = inc_index(ent), m = idx >> 2, k = idx & 3;
            const int2 pr = M.pair[m];
            const int other = inc_isA(ent) ? pr.y : pr.x;
            const int co = cutOf[m];
            int rec = rank * localSlots + idx;
            if (co >= 0) {
                const int ro = owner[other];
                rec = nranks * localSlots + (min(rank, ro) * nranks + max(rank, ro)) * slotsPerCut * 4 + co * 4 + k;
                if (other < b) rec |= 0x80000000;                      /* the earlier-created body owns the contact: swept by the other rank */
            }
            recIdx[e] = rec;
        }
    }
}

__global__ void k_unpack_records(BodyTable B, const int* __restrict__ owner, int rank, const float4* __restrict__ pose)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B.n; i += gridDim.x * blockDim.x) {
        const int o = owner[i];
        if (o < 0 || o == rank) continue;
        const float4 r0 = pose[3 * (size_t)i], r1 = pose[3 * (size_t)i + 1], r2 = pose[3 * (size_t)i + 2];
        const float radius = B.plane(BP_POSR)[i].w;
        B.plane(BP_POSR)[i] = make_float4(r0.x, r0.y, r0.z, radius);
        B.plane(BP_QUAT)[i] = make_float4(r1.x, r1.y, r1.z, r2.x);
    }
}

} // namespace avbd
