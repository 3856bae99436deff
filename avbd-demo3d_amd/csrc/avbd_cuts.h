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
                           const int* __restrict__ cutOf, int localBase, int cutBase, int slotsPerCut, int* recIdx, const GForce* __restrict__ gf, int jointBase)
{
    const int nd = lvlStart[cnt->numLevels];
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nd; s += gridDim.x * blockDim.x) {
        const int b = order[s];
        for (int e = incStart[b]; e < incStart[b + 1]; ++e) {
            const int ent = inc[e], idx = inc_index(ent), m = idx >> 2, k = idx & 3;
            if (inc_kind(ent) == FK_JOINT) {
                /* a Joint half: every rank holds every GForce at the same index, so its dual record has ONE slot in the space all ranks share
                 * (behind the contacts' slots); bit 31: the half's owner — the earlier-created dynamic endpoint — is swept by another rank */
                const GForce& f = gf[idx >> 1];
                const int other = inc_isA(ent) ? f.bodyB : f.bodyA;
                int rec = jointBase + idx;
                if (other >= 0 && other < b && owner[other] >= 0 && owner[other] != rank) rec |= 0x80000000;
                recIdx[e] = rec;
                continue;
            }
            const int2 pr = M.pair[m];
            const int other = inc_isA(ent) ? pr.y : pr.x;
            const int co = cutOf[m];
            int rec = localBase + idx;                                    /* (one allocation for all ranks: rank * L; a window per rank: 0) */
            if (co >= 0) {
                const int ro = owner[other];
                rec = cutBase + (min(rank, ro) * nranks + max(rank, ro)) * slotsPerCut * 4 + co * 4 + k;   /* the same slot in both ranks' index spaces */
                if (other < b) rec |= 0x80000000;                      /* the earlier-created body owns the contact: swept by the other rank */
            }
            recIdx[e] = rec;
        }
    }
}

struct PeerWindows { const float4* pose[8]; volatile unsigned* head[8]; int n; };   /* n == 1: one allocation for all ranks */

__global__ void k_unpack_records(BodyTable B, const int* __restrict__ owner, int rank, PeerWindows W)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B.n; i += gridDim.x * blockDim.x) {
        const int o = owner[i];
        if (o < 0 || o == rank) continue;
        /* (system-scope loads: the records were written by another process, perhaps another device, and this XCD's L2 may still hold the
         * lines it read a step ago) */
        const unsigned* pw = reinterpret_cast<const unsigned*>(W.pose[W.n > 1 ? o : 0] + 3 * (size_t)i);
        auto ldw = [&](int k) { return __uint_as_float(__hip_atomic_load(pw + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)); };
        const float4 r0 = make_float4(ldw(0), ldw(1), ldw(2), 0.0f), r1 = make_float4(ldw(4), ldw(5), ldw(6), 0.0f), r2 = make_float4(ldw(8), 0.0f, 0.0f, 0.0f);
        const float radius = B.plane(BP_POSR)[i].w;
        B.plane(BP_POSR)[i] = make_float4(r0.x, r0.y, r0.z, radius);
        B.plane(BP_QUAT)[i] = make_float4(r1.x, r1.y, r1.z, r2.x);
    }
}

/* Ranks that are processes: a rank's loop kernel is followed by k_step_done — "every record of step `serial` that this rank will ever write is
 * in its window" — and the kernel that copies the other ranks' final poses waits for that word of every peer first (k_wait_peers; bounded, the
 * error word of the context is raised on a time-out).  Besides the final poses this is what makes the windows reusable: no rank starts the
 * records of step s + 1 while a peer can still read those of step s (the last read is the owner's last dual, final_store). */
/* Ranks that are processes queue both forms of the loop kernel behind the plan and never read it back (a host synchronisation per step is a
 * pipeline bubble per step on every rank): a plan the device declined, or cut tables that overflowed, raise the context's error word instead —
 * errWord[0] = 2 / 3, decoded by the next call that looks (the peers' waves, which wait for records that will never come, time out). */
__global__ void k_plan_guard(const GroupPlan* __restrict__ plan, const CutCounters* __restrict__ cc, int* errWord)
{
    if (plan->bad) { __hip_atomic_store(errWord, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    if (cc->overflow) __hip_atomic_store(errWord, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_step_done(volatile unsigned* head, int word, unsigned serial)
{
    __hip_atomic_store(const_cast<unsigned*>(head) + word, serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait_peers(PeerWindows W, int rank, int word, unsigned serial, int* errWord, unsigned long long timeoutTicks)
{
    const int p = threadIdx.x;
    if (p >= W.n || p == rank) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned v = __hip_atomic_load(const_cast<unsigned*>(W.head[p]) + word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(v - serial) >= 0) return;
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeoutTicks || __hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(errWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

} // namespace avbd
