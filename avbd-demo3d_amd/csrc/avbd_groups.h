/*
 * avbd_groups.h — "group" schedule of the iteration loop (solver.cpp:340-431): several bodies in flight per wavefront.
 *
 * The resident schedule (avbd_resident.h) gives a body a whole wavefront: one lane per contact, 27 lanes for the ordered
 * accumulation, the rest idle — fine while there are at most as many swept bodies as resident waves, a serial resource
 * beyond (9.6 us of wave time per task once a wave has to park and reload bodies).  Here a wavefront is cut
 * into sub-groups of 16 / 32 / 64 lanes by the body's number of incidence entries, each sub-group owns ONE body for the
 * whole launch (state in registers, like the one-body-per-wave form) and the sub-groups of a wave advance independently:
 * a pass of the wave's loop polls the records every waiting sub-group still misses and then runs the task of every
 * sub-group whose inputs have arrived, under the lane mask of those sub-groups.  No sub-group ever waits for another one
 * of its wave, so the only dependencies are the reference's own (avbd_iterate.h) and any placement is deadlock-free.
 *
 *   - same hand-off records, stamps and arithmetic as avbd_resident.h (pose / visit / dual, 16-byte `sc1` records);
 *   - per-wave LDS is indexed by lane exactly as there (row = 3 * lane + k), so a sub-group simply uses the rows of its
 *     own lanes; the 27 accumulator entries of a body live on the first 16 lanes of its sub-group (two per lane);
 *   - a waiting lane polls ONE record (the last one its producer writes) until that arrives and only then fetches the
 *     rest: waves that cannot run stay off the memory system (round 1: 3.35x the algorithmic traffic was poll traffic).
 *
 * Plan (k_group_plan, every step, on the device): swept bodies are classed by entry count (<= 16, <= 32, <= 64), each
 * class gets its own waves (a wave's lane layout is static), consecutive bodies of a class go to different waves.
 */
#pragma once
#include "avbd_iterate.h"

#pragma clang fp contract(off)

namespace avbd {

constexpr int kMaxExactRanks = 8;

struct GroupPlan {
    int nClass[3];     /* swept bodies with <= 16 / <= 32 / <= 64 incidence entries */
    int waves[3];      /* resident waves of each class; layout in the launch: class 0 (16 lanes per body) | class 1 | class 2 */
    int rounds[3];     /* bodies per sub-group */
    int bad;           /* != 0: this step cannot run in the group schedule (the launch queued behind it runs instead) */
    int nd;            /* swept bodies */
    int pad[5];
};
static_assert(sizeof(GroupPlan) == 64, "GroupPlan layout");

/* Sharded runs in ONE persistent launch per step: this rank sweeps the bodies it owns; a body owned by another
 * rank ("remote") is a fixed pose during an iteration and changes between iterations through the all-gather.
 *   - the wave that publishes a pose also writes the body's row (position+radius, orientation) into this rank's block of
 *     gather[it & 1]; a wave counts its bodies in done[it] once all of them are past iteration it, and the wave that completes
 *     the rank's iteration raises `doneSignal` to base + it + 1, which is what a second stream waits for (hipStreamWaitValue32) before it all-gathers gather[it & 1] in
 *     place and writes base + it + 1 into `gatherWord` (hipStreamWriteValue32);
 *   - a task of iteration it >= 1 with a remote partner polls `gatherWord` until it reaches base + it (the all-gather behind
 *     iteration it - 1) and then reads the partner's row from gather[(it - 1) & 1].  Tasks without a remote partner never
 *     wait for the exchange: the interior of a slab runs ahead of it.
 * Two gather buffers are enough: gather(it + 1) needs every task of iteration it + 1, and those come after the last reader
 * of gather(it - 1). */
struct ShardExchange {
    float4* gather[2];        /* [nranks * maxCount * 2]: rank-major blocks of (position+radius, orientation) rows */
    unsigned gatherBytes;
    const int* row;           /* [bodies]: row of a body inside a gather buffer (its owner's block), -1 for replicated (static) bodies */
    unsigned* done;           /* [64]: tasks of iteration it this rank has finished (zeroed before the launch) */
    unsigned* doneSignal;     /* hipMallocSignalMemory: base + (iterations completed by this rank) */
    unsigned* gatherWord;     /* hipMallocSignalMemory: base + (all-gathers that have landed) */
    int nOwned;               /* swept bodies of this rank */
    int keepExchangeWaiting;  /* debug (AVBD_DEBUG_KEEP_EXCHANGE_WAITING): an abandoned launch does not release the exchange stream, so the other ranks stay as they are */
    /* EXACT cuts (SHARDK == 2): no exchange at all during the loop.  Every rank sweeps the bodies it owns with the
     * single-GPU protocol; the stamped records of ALL ranks live in one address space (one allocation shared by the contexts of a
     * process; peer-to-peer windows between GPUs), so a body whose partner another rank sweeps polls that partner's pose / visit /
     * dual records exactly as it polls a local one and the result is the single-GPU sweep, bit for bit.  Pose records are indexed
     * by body; the visit / dual records of a contact by recIdx: this rank's own range, or — contacts across a cut — a slot both
     * ranks of the cut compute alike (k_cut_rank: rank of the manifold among the cut's manifolds, ordered by body pair). */
    float4 *xPose, *xVisit, *xDual; unsigned xPoseBytes, xVisitBytes, xDualBytes;
    /* Ranks that are PROCESSES (one per GPU, or several on one device): every rank WRITES its records into a window of its own memory and READS
     * the records of a partner body — pose, and the visit / dual records of the contacts it shares with it — from the window of the rank that
     * sweeps that body, opened with hipIpcOpenMemHandle (avbd_exact_window_export / _attach).  All windows have the same layout, so a record has
     * the same offset in each of them.  nPeers == 1: the in-process emulation, one allocation for all ranks (xPose / xVisit / xDual). */
    float4 *peerPose[kMaxExactRanks], *peerVisit[kMaxExactRanks], *peerDual[kMaxExactRanks];
    int nPeers;
    const int* owner;         /* [bodies]: rank that sweeps a body (-1: static) — which window a partner's records are read from */
    unsigned long long* span; /* SHARD, profiling: [0] s_memrealtime (10 ns ticks) when the launch's first wave started, [1] when the rank's last iteration was complete; nullptr: off */
    const int* recIdx;        /* [incidence entries]: record index of the entry's contact | bit 31: the contact's owner is swept by another rank
                               * (this rank's copy of the contact takes the owner's last dual from its record when the loop ends) */
};

constexpr int kGrpWavesPerBlock = 4;
constexpr int kGrpSSlots = 7;             /* f | penalty | gyro.x | gyro.y | gyro.z | +0.0f | -0.0f */
constexpr int kGrpAccFloats = 64;         /* per 16 lanes: 27 accumulator entries, 21 floats of solve / inertia exchange */
constexpr int kGrpConstFloats = 40;       /* per 16 lanes, ten float4: inertial position | mass, inertial orientation, three rows of the body-frame inertia (w of the
                                            * first: inverse mass), three of its inverse, position | radius, orientation — the layout of a body's parked record (parkB) */
constexpr int kGrpWaveFloats = kResJFloats + kGrpSSlots * kResRows + 4 * (kGrpAccFloats + kGrpConstFloats);
constexpr int kGrpLdsBytes = kGrpWavesPerBlock * kGrpWaveFloats * 4;    /* 43 008 B: two workgroups per CU */
/* several bodies per sub-group: who the entries of a sub-group's bodies are (incidence word, partner | moving << 24 | owned << 25) stays in LDS
 * for the whole launch, hdr[round][lane]: examining a body costs no memory round trip for its header */
constexpr int kGrpMaxRounds = 16;
constexpr int kGrpHdrBytesPerWave = kGrpMaxRounds * 64 * 8;
/* The pool of a workgroup (iterate_groups, pool loop) has four words per slot — body, first incidence entry, arrivals the next task waits for,
 * and next iteration | claimed << 8 | entries << 16 | arrivals per iteration << 24 — in the tails of the four waves' header tables, which a plan
 * of at most kGrpPoolMaxRounds bodies per sub-group leaves free: no LDS of its own. */
constexpr int kGrpPoolMaxRounds = 12;
constexpr int kGrpLdsBytesMulti = kGrpLdsBytes + kGrpWavesPerBlock * kGrpHdrBytesPerWave;   /* 79 360 B */
static_assert((kGrpMaxRounds - kGrpPoolMaxRounds) * 64 * 8 >= 16 * kGrpPoolMaxRounds * 4 && 16 * kGrpPoolMaxRounds <= 3 * 64, "a pool's slot array fits the tail of a header table; three slots per lane");
static_assert(kGrpWaveFloats % 4 == 0 && 2 * kGrpLdsBytesMulti <= 160 * 1024, "two workgroups of k_iterate_groups share a CU's LDS");

/* ---- plan: classes, waves per class, the bodies of each class in sweep order -----------------------------------------------
 * byClass[c * cap + r] = r-th body of class c in sweep order.  Sub-group j of wave w (the wl-th wave of its class, which has Wc
 * waves of G = 4 >> c sub-groups) serves, in round k, rank r = (k * G + j) * Wc + wl — bodies that follow each other in the
 * sweep go to different waves.  One tile of 4096 slots of the sweep order per workgroup (tile_exclusive_prefix); the last
 * tile, which learns the totals from its look-back, writes the plan header. */
/* Which body a sub-group serves.  Layout of the launch's waves: class 0 (16 lanes per body) | class 1 | class 2, so that the class-0 waves
 * fill whole workgroups from wave 0 on.  Classes 1 and 2, and the class-0 waves behind the last FULL class-0 workgroup: sub-group j of the wl-th
 * wave serves, in round k, rank r = (k * G + j) * Wc + wl of its class — bodies that follow each other in the sweep go to different waves
 * (other static assignments were measured in round 3 and removed: neighbours of the sweep are ready together, and a wave runs its ready
 * sub-groups one pass at a time).  `pooled`: the class runs as POOLS — every workgroup (a plan with several bodies per sub-group gives each class
 * whole workgroups) serves a contiguous run of its class's ranks, and any of its sub-groups runs any of those bodies (iterate_groups, pool loop). */
__device__ __forceinline__ int group_unit_rank(const GroupPlan* plan, int cls, int wl, int j, int round, bool pooled)
{
    const int Wc = plan->waves[cls], G = 4 >> cls;
    if (pooled) {                                                      /* slot (wave-in-workgroup, sub-group, round) of pool wl / 4: a contiguous run of the class's ranks */
        const int R = plan->rounds[cls];
        return (wl >> 2) * (4 * G * R) + ((wl & 3) * G + j) * R + round;
    }
    return (round * G + j) * Wc + wl;
}
__device__ __forceinline__ int group_unit_body(const GroupPlan* plan, const int* __restrict__ byClass, int cap, int cls, int wl, int j, int round, bool pooled = false)
{
    const int r = group_unit_rank(plan, cls, wl, j, round, pooled);
    return r < plan->nClass[cls] ? byClass[cls * cap + r] : -1;
}

__global__ __launch_bounds__(1024) void k_group_plan(BodyTable B, const int* __restrict__ order, const int* __restrict__ lvlStart, Counters* cnt,
                                                     const int* __restrict__ incStart, GroupPlan* plan, int* __restrict__ byClass, int cap, int W, int maxRounds,
                                                     tile_v4u* agg, unsigned stamp, unsigned* __restrict__ depCnt, int* __restrict__ rankOf)
{
    __shared__ unsigned long long sWaveTotals[16];
    __shared__ unsigned long long sBcast, sTileSum;
    const int t = threadIdx.x, tile = blockIdx.x;
    const int nd = lvlStart[cnt->numLevels];
    const int s0 = tile * kTileBodies + 4 * t;
    int cl[4], body[4]; unsigned long long mine4 = 0;                 /* class counts, 21 bits each */
    for (int k = 0; k < 4; ++k) { body[k] = s0 + k < nd ? order[s0 + k] : -1; cl[k] = -1; }
    for (int k = 0; k < 4; ++k)
        if (body[k] >= 0) {
            const int ne = incStart[body[k] + 1] - incStart[body[k]];
            /* fewer bodies than waves (small ranks of a sharded run): a whole wave per body — no other sub-group ever holds its
             * wave while its inputs arrive */
            cl[k] = (nd <= W || ne > 32) ? 2 : (ne <= 16 ? 0 : 1);
            mine4 += 1ull << (21 * cl[k]);
        }
    const unsigned long long incl = block_scan_inclusive_u64(mine4, sWaveTotals);
    if (t == 1023) sTileSum = incl;
    __syncthreads();
    const unsigned long long before = tile_exclusive_prefix(agg, tile, sTileSum, stamp, &sBcast);
    unsigned long long off = before + incl - mine4;
    for (int k = 0; k < 4; ++k) {
        if (cl[k] < 0) continue;
        const int c = cl[k];
        const int r = (int)((off >> (21 * c)) & 0x1fffff);
        off += 1ull << (21 * c);
        byClass[c * cap + r] = body[k];
        /* arrival counters of the several-bodies loops (iterate_groups) live at the body's class-major rank: the bodies of a pool are
         * neighbours there, one poll of a pool's counters touches a few cache lines */
        if (depCnt) { depCnt[c * cap + r] = 0u; rankOf[body[k]] = c * cap + r; }
    }
    if (t == 1023 && tile == (int)gridDim.x - 1) {
        const unsigned long long all = before + incl;
        int n[3] = {(int)(all & 0x1fffff), (int)((all >> 21) & 0x1fffff), (int)((all >> 42) & 0x1fffff)};
        int wv[3], rounds[3] = {1, 1, 1};
        const int need1 = (n[1] + 1) / 2, need0 = (n[0] + 3) / 4;
        int bad = (cnt->residentBad & 1) || nd == 0 || nd >= (1 << 21) || nd > (int)gridDim.x * kTileBodies;
        if (n[2] + need1 + need0 <= W) {
            /* every body keeps its sub-group for the whole launch; class 0 takes all the waves that are left */
            wv[2] = n[2]; wv[1] = need1; wv[0] = n[0] ? W - wv[2] - wv[1] : 0;
        } else {
            /* several bodies per sub-group: waves in proportion to the lane-quarters a class needs */
            const long long U = (long long)n[0] + 2ll * n[1] + 4ll * n[2];
            wv[2] = n[2] ? (int)max(1ll, (4ll * n[2] * W) / U) : 0;
            wv[1] = n[1] ? (int)max(1ll, (2ll * n[1] * W) / U) : 0;
            /* a class of a few bodies must not end up with more rounds than a sub-group can hold (19 bodies of 17..32 contacts on the
             * one wave their share of the lanes gives them would be ten rounds): at least the waves that keep it within maxRounds */
            wv[2] = max(wv[2], (n[2] + maxRounds - 1) / maxRounds);
            wv[1] = max(wv[1], (n[1] + 2 * maxRounds - 1) / (2 * maxRounds));
            /* whole workgroups per class: a workgroup whose four waves serve one class can share its bodies between them (pool loop) */
            if ((W & 3) == 0 && W - ((wv[2] + 3) & ~3) - ((wv[1] + 3) & ~3) >= (n[0] ? 4 : 0)) { wv[2] = (wv[2] + 3) & ~3; wv[1] = (wv[1] + 3) & ~3; }
            wv[0] = W - wv[2] - wv[1];
            if (wv[0] < (n[0] ? 1 : 0)) bad = 1;
            for (int c = 0; c < 3; ++c) if (n[c]) { const int per = wv[c] * (4 >> c); rounds[c] = (n[c] + per - 1) / per; }
        }
        const int kmax = max(rounds[0], max(rounds[1], rounds[2]));
        if (kmax > maxRounds) bad = 1;
        plan->nClass[0] = n[0]; plan->nClass[1] = n[1]; plan->nClass[2] = n[2];
        plan->waves[0] = wv[0]; plan->waves[1] = wv[1]; plan->waves[2] = wv[2];
        plan->rounds[0] = rounds[0]; plan->rounds[1] = rounds[1]; plan->rounds[2] = rounds[2];
        plan->bad = bad; plan->nd = nd;
    }
}

/* World-space inertia and inverse inertia (rigid.cpp:51-59) by the first nine lanes of a sub-group, one entry (i, j) of both
 * matrices per lane; same expression trees as world_inertia_parallel. */
__device__ __forceinline__ void world_inertia_group(int gl, q4 rot, m33 Ib, m33 Jb, float* xch, m33& Iw, m33& invIw)
{
    const m33 R = quat_to_mat(rot);
    const m33 Rt = mtranspose(R);
    const int e = gl < 9 ? gl : 0;
    const int i = e % 3, j = e / 3;
    const f3 ri = i == 0 ? Rt.c0 : (i == 1 ? Rt.c1 : Rt.c2);
    const f3 rj = j == 0 ? Rt.c0 : (j == 1 ? Rt.c1 : Rt.c2);
    const f3 mi = mk3(dot3(ri, Ib.c0), dot3(ri, Ib.c1), dot3(ri, Ib.c2));
    const f3 mj = mk3(dot3(ri, Jb.c0), dot3(ri, Jb.c1), dot3(ri, Jb.c2));
    const float v = dot3(mi, rj), v2 = dot3(mj, rj);
    if (gl < 9) { xch[gl] = v; xch[9 + gl] = v2; }
    wave_sync();
    Iw = mkm(mk3(xch[0], xch[1], xch[2]), mk3(xch[3], xch[4], xch[5]), mk3(xch[6], xch[7], xch[8]));
    invIw = mkm(mk3(xch[9], xch[10], xch[11]), mk3(xch[12], xch[13], xch[14]), mk3(xch[15], xch[16], xch[17]));
    wave_sync();
}

/* one accumulator entry: operand offsets into a J row, kind, gyro slot, running sum */
struct GAcc { int opA, opB, mode, gy; float acc; };
__device__ __forceinline__ GAcc make_gacc(int entry, const BodyLocal& L)
{
    const AccLane a = init_acc_lane(entry, L);
    GAcc g; g.opA = a.opA; g.opB = a.opB; g.mode = a.mode; g.gy = a.gy; g.acc = a.acc;
    return g;
}

/* operands and kind of accumulator entry `entry` (0-5 ll lower, 6-14 la, 15-20 aa lower, 21-26 rhs): the part of init_acc_lane
 * that does not depend on the body */
__device__ __forceinline__ void grp_entry_ops(int entry, GAcc& a)
{
    a.opA = 0; a.opB = 0; a.mode = 3; a.gy = -1; a.acc = 0.0f;
    const int k6 = entry < 6 ? entry : (entry >= 15 && entry < 21 ? entry - 15 : 0);
    const int ti = k6 < 3 ? k6 : (k6 < 5 ? k6 - 2 : 2);
    const int tj = k6 < 3 ? 0 : (k6 < 5 ? 1 : 2);
    if (entry < 6) { a.mode = 0; a.opA = ti; a.opB = tj; }
    else if (entry < 15) { const int k = entry - 6; a.mode = 0; a.opA = k % 3; a.opB = 3 + k / 3; }
    else if (entry < 21) { a.mode = 1; a.opA = 3 + ti; a.opB = 3 + tj; a.gy = (ti == tj) ? (8 + ti) : -1; }
    else if (entry < 24) { a.mode = 2; a.opA = entry - 21; a.opB = 6; }
    else if (entry < 27) { a.mode = 2; a.opA = 3 + (entry - 24); a.opB = 6; }
}

/* ordered accumulation (solver.cpp:383-397): acc += (Ja*Jb)*penalty [+ gyro] / acc += J*f, row by row in list order.  The
 * Jacobians lie TRANSPOSED in LDS (Jt[column][row]), so one 16-byte read brings an operand of four consecutive rows; the
 * products are formed here, next to the additions they feed (the multiplications are off the chain of additions).  TWO: the
 * two entries of a lane at once (two independent chains).  ALLON: every row has a usable penalty (solver.cpp:387), the normal
 * case; otherwise a lhs entry skips the rows whose penalty is not positive and finite.  `rowBase`: first row of the sub-group
 * (a multiple of 48). */
struct GAccView { const float4 *a4, *b4, *s4, *g4; bool isAA, rhs; int bMask; };
constexpr int kGrpOnesFloat = 56;         /* accs[56..59] of a wave's first sub-group: four 1.0f — the second factor of the rhs entries (x * 1.0f == x for every x), so that the
                                            * products of a row are two multiplications on every lane, no select in between, and pair up into packed instructions */
__device__ __forceinline__ GAccView grp_view(const GAcc& a, const float* Jt, const float* S, int rowBase)
{
    GAccView v;
    v.isAA = a.mode == 1; v.rhs = a.mode == 2;
    v.a4 = reinterpret_cast<const float4*>(Jt + a.opA * kResRows + rowBase);
    v.b4 = v.rhs ? reinterpret_cast<const float4*>(S + kGrpSSlots * kResRows + kGrpOnesFloat) : reinterpret_cast<const float4*>(Jt + a.opB * kResRows + rowBase);
    v.bMask = v.rhs ? 0 : -1;                                        /* rhs entries: the same four 1.0f for every block */
    v.s4 = reinterpret_cast<const float4*>(S + (v.rhs ? 0 : kResRows) + rowBase);
    /* behind the row term: the gyroscopic component (diagonal aa entries), +0.0f (off-diagonal aa entries, like the reference's
     * diagonal() matrix) or -0.0f (every other entry: x + (-0.0f) == x for every x, so no lane needs a select around the addition) */
    v.g4 = reinterpret_cast<const float4*>(S + kResRows * (a.gy >= 0 ? a.gy - 6 : (v.isAA ? 5 : 6)) + rowBase);
    return v;
}
template <bool ALLON>
__device__ __forceinline__ float grp_add_row(float acc, float va, float vb, float sv, float gv, const GAccView& v)
{
    const float pr = va * vb;                                        /* (rhs entries: vb = 1.0f) */
    const float nv = acc + pr * sv;
    const float nvg = nv + gv;
    if (ALLON) return nvg;
    const float nv2 = v.isAA ? nvg : nv;
    const bool on = v.rhs || ((sv > 0.0f) & (fabsf(sv) <= FLT_MAX));   /* lhs entries scale by the penalty itself */
    return on ? nv2 : acc;
}
template <bool ALLON, bool TWO>
__device__ __forceinline__ void grp_accumulate(GAcc& a0, GAcc& a1, const float* Jt, const float* S, int rowBase, int nRows)
{
    const GAccView v0 = grp_view(a0, Jt, S, rowBase), v1 = grp_view(a1, Jt, S, rowBase);
    float acc0 = a0.acc, acc1 = a1.acc;
    /* whole blocks of four rows without predicates (the next block's operands are in flight while this one is added; the block
     * behind the last full one exists in LDS, whatever it holds), then the last one to three rows */
    const int nFull = nRows & ~3;
    float4 ja0 = v0.a4[0], jb0 = v0.b4[0], s0 = v0.s4[0], g0 = v0.g4[0], ja1, jb1, s1, g1;
    if (TWO) { ja1 = v1.a4[0]; jb1 = v1.b4[0]; s1 = v1.s4[0]; g1 = v1.g4[0]; }
    for (int r4 = 0; r4 < nFull; r4 += 4) {
        const float4 ca0 = ja0, cb0 = jb0, cs0 = s0, cg0 = g0, ca1 = ja1, cb1 = jb1, cs1 = s1, cg1 = g1;
        const int q = (r4 >> 2) + 1;
        ja0 = v0.a4[q]; jb0 = v0.b4[q & v0.bMask]; s0 = v0.s4[q]; g0 = v0.g4[q];
        if (TWO) { ja1 = v1.a4[q]; jb1 = v1.b4[q & v1.bMask]; s1 = v1.s4[q]; g1 = v1.g4[q]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc0 = grp_add_row<ALLON>(acc0, comp4(ca0, k), comp4(cb0, k), comp4(cs0, k), comp4(cg0, k), v0);
            if (TWO) acc1 = grp_add_row<ALLON>(acc1, comp4(ca1, k), comp4(cb1, k), comp4(cs1, k), comp4(cg1, k), v1);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (nFull + k < nRows) {
            acc0 = grp_add_row<ALLON>(acc0, comp4(ja0, k), comp4(jb0, k), comp4(s0, k), comp4(g0, k), v0);
            if (TWO) acc1 = grp_add_row<ALLON>(acc1, comp4(ja1, k), comp4(jb1, k), comp4(s1, k), comp4(g1, k), v1);
        }
    }
    a0.acc = acc0; a1.acc = acc1;
}

__device__ __forceinline__ float grp_bcast(float v, int srcLane) { return __shfl(v, srcLane, 64); }
/* minimum over the 64 lanes of a wavefront (all of them active), wave-uniform: six DPP steps on the vector ALU — quads, half rows, rows, then the rows' last
 * lanes broadcast into the rows behind them — instead of six ds_bpermute round trips through the LDS crossbar (what __shfl_xor compiles to: ~100 cycles each,
 * and the claim loop of the pools runs this reduction up to five times per pass, one behind the other) */
__device__ __forceinline__ int wave_min_i32(int v)
{
    int t;
    t = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false); v = t < v ? t : v;      /* quad_perm:[1,0,3,2] */
    t = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false); v = t < v ? t : v;      /* quad_perm:[2,3,0,1] */
    t = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false); v = t < v ? t : v;     /* row_half_mirror */
    t = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false); v = t < v ? t : v;     /* row_mirror: every lane holds its row's minimum */
    t = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); v = t < v ? t : v;     /* row_bcast:15 into rows 1 and 3 */
    t = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false); v = t < v ? t : v;     /* row_bcast:31 into rows 2 and 3 */
    return __builtin_amdgcn_readlane(v, 63);
}
/* minimum over a sub-group of gw = 16 / 32 / 64 lanes (aligned; every lane of it active), in all its lanes: the four steps inside a 16-lane row on the vector ALU,
 * the one or two across rows through the crossbar */
__device__ __forceinline__ int group_min_i32(int v, int gw)
{
    int t;
    t = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false); v = t < v ? t : v;
    t = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false); v = t < v ? t : v;
    if (gw > 16) { t = __shfl_xor(v, 16, 64); v = t < v ? t : v; }       /* (wave-uniform: gw is the class's) */
    if (gw > 32) { t = __shfl_xor(v, 32, 64); v = t < v ? t : v; }
    return v;
}
/* the stamp word of a record alone (aux 16 = sc1, like xld) */
__device__ __forceinline__ unsigned xldw(__amdgpu_buffer_rsrc_t r, int byteOff) { return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, byteOff, 0, 16); }

/* The whole iteration loop, contacts only.  A sub-group keeps ONE body for the launch when the plan has one round (state in
 * registers from the first iteration to the last); with several rounds it runs whichever of its bodies has its inputs
 * published (oldest iteration first) and parks a body's state between its tasks exactly like the resident schedule does
 * (owned contacts in the Manifold tables, which only their owner writes; a first visitor's copy of normal + tangent memo
 * and the pose in arrays private to the sub-group).  The sub-groups of a wave pick independently and the wave runs their
 * picks together, so a wave that serves 4 x K bodies does several tasks per pass. */
/* JOINTS: a lane holds one Manifold contact or one half (three rows) of a Joint, as in the resident schedule (avbd_resident.h); the joint-aware code is
 * compiled only into the kernels that scenes with Joint forces launch (it costs registers on every path).  A joint lane keeps its constants in the
 * fields a contact lane uses for things a joint does not have — no extra registers live across the loop:
 *   c.rA, c.rB anchors | c.normal.xyz, c.u1.x q_rel0 | c.u1.yz, c.u2.x stiffness of its three rows | c.u2.yz, c.C0n motor | pOther, (wOther, c.C0t1) the partner's
 *   position and orientation | c.lambda, c.penalty, c.C the rows. */
template <bool MULTI, int SHARDK, bool JOINTS = false>      /* SHARDK 0: one GPU; 1: sharded, all-gather per iteration; 2: sharded, exact cuts */
__device__ __forceinline__ void iterate_groups(BodyTable B, ManifoldTable M, const int* __restrict__ incStart, const int* __restrict__ inc, SolverParams P, int* errWord,
                      Counters* cntW, ResidentExchange X, const GroupPlan* __restrict__ plan, const int* __restrict__ byClass, int cap, int sleepSel, ShardExchange Z,
                      int blockIndex /* workgroup of this rank's share of the launch */)
{
    extern __shared__ float4 sGroups[];                              /* kGrpLdsBytes: per wave Jt | S | accs | constants */
    constexpr bool SHARD = SHARDK == 1, EXACT = SHARDK == 2;
    static_assert(!JOINTS || SHARDK == 0 || SHARDK == 2, "joint entries: one GPU, or exact cuts (the all-gather forms keep the resident schedule)");
    GForce* const gf = X.gf;
    /* the host queues k_iterate_groups<false>, <true> and the fallback launch one behind the other; the plan says which one runs */
    if (plan->bad || (plan->rounds[0] > 1 || plan->rounds[1] > 1 || plan->rounds[2] > 1) != MULTI) return;   /* (SHARD: the host has read the plan) */
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIndex * kGrpWavesPerBlock + wv;
    const int W2 = plan->waves[2], W1 = plan->waves[1], W0 = plan->waves[0];
    if (w >= W2 + W1 + W0) return;
    const int cls = w < W0 ? 0 : (w < W0 + W1 ? 1 : 2);              /* wave-uniform; class 0 first: its waves fill whole workgroups (pools) */
    const int wl = w - (cls == 0 ? 0 : (cls == 1 ? W0 : W0 + W1));    /* which wave of its class */
    const int rounds = __builtin_amdgcn_readfirstlane(plan->rounds[cls]);
    /* does this wave's class run as pools (several bodies per sub-group, one GPU, arrival counters, whole workgroups per class)? */
    const bool pooled = MULTI && SHARDK == 0 && X.depCnt != nullptr && X.poolMode != 0 && rounds >= X.poolMinRounds && rounds <= kGrpPoolMaxRounds &&
                        (W0 & 3) == 0 && (W1 & 3) == 0 && (W2 & 3) == 0;
    const int gw = 16 << cls, gl = lane & (gw - 1), gBase = lane & ~(gw - 1);
    const unsigned long long gMask = (cls == 2 ? ~0ull : ((1ull << gw) - 1ull)) << gBase;
    float* const J = reinterpret_cast<float*>(sGroups) + wv * kGrpWaveFloats;
    float* const S = J + kResJFloats;
    const int rowBase = gBase * 3;                                   /* the sub-group's first row in Jt and S */
    float* const accs = S + kGrpSSlots * kResRows + (gBase >> 4) * kGrpAccFloats;
    float* const xch = accs + 32;
    float* const cst = S + kGrpSSlots * kResRows + 4 * kGrpAccFloats + (gBase >> 4) * kGrpConstFloats;

    /* EXACT: the records cross process (and device) boundaries — system scope (sc0 sc1) instead of device scope (sc1); the same price on one
     * device (tools/ubench/ipc_pingpong.hip: 0.53 us per hop either way, plain or uncached memory) */
    constexpr int AUX = EXACT ? 17 : 16;
    auto xldA = [](__amdgpu_buffer_rsrc_t r, int byteOff) __attribute__((always_inline)) -> v4u { return __builtin_amdgcn_raw_buffer_load_b128(r, byteOff, 0, AUX); };
    auto xstA = [](__amdgpu_buffer_rsrc_t r, int byteOff, v4u v) __attribute__((always_inline)) { __builtin_amdgcn_raw_buffer_store_b128(v, r, byteOff, 0, AUX); };
    auto xldwA = [](__amdgpu_buffer_rsrc_t r, int byteOff) __attribute__((always_inline)) -> unsigned { return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, byteOff, 0, AUX); };
    int peer = 0;                                                     /* EXACT: whose window this lane's partner publishes in */
    int ctrO = 0;                                                     /* arrival counters: where this lane's partner counts (its class-major rank, k_group_plan's rankOf) */
    const __amdgpu_buffer_rsrc_t rP = EXACT ? xrsrc(Z.xPose, Z.xPoseBytes) : xrsrc(X.pose, X.poseBytes),
                                 rV = EXACT ? xrsrc(Z.xVisit, Z.xVisitBytes) : xrsrc(X.visit, X.visitBytes),
                                 rD = EXACT ? xrsrc(Z.xDual, Z.xDualBytes) : xrsrc(X.dual, X.dualBytes);
    const __amdgpu_buffer_rsrc_t rG0 = xrsrc(SHARD ? (void*)Z.gather[0] : (void*)X.pose, SHARD ? Z.gatherBytes : 16u),
                                 rG1 = xrsrc(SHARD ? (void*)Z.gather[1] : (void*)X.pose, SHARD ? Z.gatherBytes : 16u),
                                 rGW = xrsrc(SHARD ? (void*)Z.gatherWord : (void*)X.pose, 4u);
    const int itStop = P.iterations + (P.postStabilize ? 1 : 0);
    const float4* massPlane = B.plane(BP_MASS);
    const float invDt2 = 1.0f / (P.dt * P.dt);

    /* ---- the task in hand: body and iteration, uniform over the lanes of the sub-group ---- */
    int it = 0;
    int body = -1, eBeg = 0, nRows = 0;
    bool valid = false, has = false;
    int idx = 0, other = 0; bool isA = false, owned = false, moving = false, fixedN = false;
    bool isJoint = false, hasA = true;                                /* JOINTS: this lane holds a joint half (rows 3 * (idx & 1) ..); the joint has a bodyA */
    bool remote = false; int offR = 0; int ownRow = 0;                /* SHARD: the partner is swept by another rank; its row / this body's row in a gather buffer */
    bool finalD = false;                                              /* EXACT: the contact's owner is swept by another rank */
    float oinv = 0.0f, fric = 0.0f;
    ContactRows c;
    c.rA = c.rB = c.normal = c.u1 = c.u2 = zero3(); c.C0n = c.C0t1 = c.C0t2 = 0.0f; c.penetration = 0.0f; c.stick = false;
    for (int k = 0; k < 3; ++k) { c.lambda[k] = 0.0f; c.penalty[k] = 0.0f; c.C[k] = 0.0f; c.fmin[k] = 0.0f; c.fmax[k] = 0.0f; }
    f3 wOther = zero3(), pOther = zero3();
    int offP = 0, offV = 0, offD = 0;
    f3 pos = zero3(); q4 rot = identq(); float invMass = 0.0f, radius = 0.0f;

    /* who the entries of a body are and whom they wait for: enough to tell whether a task can start.  `seen`: the body has been
     * examined before in this launch, its partners are in X.probe (partner | moving << 24 | owned << 25). */
    typedef typename std::conditional<EXACT, int4, int2>::type HdrT;   /* EXACT: + the record index (half as many rounds fit) */
    HdrT* const hdr = reinterpret_cast<HdrT*>(reinterpret_cast<char*>(sGroups) + kGrpLdsBytes + wv * kGrpHdrBytesPerWave);   /* MULTI: [round][lane] */
    const HdrT* hdrRow = nullptr;                                     /* pool loop: the header row of the slot in hand (any wave's table) */
    auto load_header = [&](int b, int e0, int ne, int round) __attribute__((always_inline)) {
        body = b; valid = b >= 0; eBeg = e0;
        has = valid && gl < ne;
        nRows = ne * 3;
        idx = 0; other = 0; isA = false; owned = false; moving = false;
        if (JOINTS) { isJoint = false; hasA = true; }
        int rec = 0;
        if (MULTI) {
            const HdrT h = hdrRow ? hdrRow[gl] : hdr[round * 64 + lane];   /* filled once, before the loop */
            if (has) { idx = inc_index(h.x); isA = inc_isA(h.x) != 0; other = h.y & 0xffffff; moving = (h.y >> 24) & 1; owned = (h.y >> 25) & 1; }
            if (JOINTS && has) { isJoint = inc_kind(h.x) == FK_JOINT; hasA = (h.y >> 26) & 1; }
            if constexpr (EXACT) rec = has ? h.z : 0; else rec = (JOINTS && isJoint) ? X.jointRecBase + idx : idx;
        } else if (has) {
            const int ent = inc[e0 + gl];
            idx = inc_index(ent); isA = inc_isA(ent) != 0;
            if (JOINTS && inc_kind(ent) == FK_JOINT) {
                const GForce& f = gf[idx >> 1];
                isJoint = true; hasA = f.bodyA >= 0;
                other = isA ? f.bodyB : f.bodyA;
                moving = other >= 0 && (EXACT ? massPlane[other].y > 0.0f : B.solved[other] != 0);
                owned = !moving || other > b;
                if (other < 0) other = 0;                                  /* world anchor: a fixed partner, never polled */
                rec = EXACT ? Z.recIdx[e0 + gl] : X.jointRecBase + idx;    /* (EXACT: k_cut_fill — the same slot on every rank, bit 31 when another rank sweeps the half's owner) */
            } else {
            const int2 pr = M.pair[idx >> 2];
            other = isA ? pr.y : pr.x;
            moving = EXACT ? massPlane[other].y > 0.0f : B.solved[other] != 0;   /* EXACT: whoever sweeps it, a dynamic partner moves */
            owned = !moving || other > b;
            rec = EXACT ? Z.recIdx[e0 + gl] : idx;
            }
        }
        finalD = EXACT && has && (rec < 0);
        rec &= 0x7fffffff;
        if constexpr (EXACT) { peer = 0; if (Z.nPeers > 1 && has && moving) { if constexpr (MULTI) peer = hdr[round * 64 + lane].w; else peer = max(Z.owner[other], 0); } }
        offP = other * 48; offV = rec * 16; offD = rec * 32;
    };
    /* JOINTS: one half of a Joint (joint.cpp:68-139) — its constants and rows from the GForce record, which only the half's owner writes during the
     * launch (and where it parks lambda / penalty between its tasks), and the partner's pose from the planes: the predicted pose of a partner that has
     * not run yet, the constant pose of a partner nobody sweeps; a moving partner's later poses come from its records (finish_visit) */
    auto load_joint = [&]() __attribute__((always_inline)) {
        const GForce& f = gf[idx >> 1];
        const int jp = (idx & 1) * 3;
        c.rA = mk3(f.rA[0], f.rA[1], f.rA[2]); c.rB = mk3(f.rB[0], f.rB[1], f.rB[2]);
        c.normal = mk3(f.q0[0], f.q0[1], f.q0[2]); c.u1.x = f.q0[3];
        c.u1.y = f.stiffness[jp]; c.u1.z = f.stiffness[jp + 1]; c.u2.x = f.stiffness[jp + 2];
        c.u2.y = f.motor[jp]; c.u2.z = f.motor[jp + 1]; c.C0n = f.motor[jp + 2];
        c.C0t2 = (f.pad & 1) ? 1.0f : 0.0f;                            /* avbd_set_joint_angular_sign */
        for (int k = 0; k < 3; ++k) { c.lambda[k] = f.lambda[jp + k]; c.penalty[k] = f.penalty[jp + k]; c.C[k] = f.C[jp + k]; }
        pOther = zero3(); wOther = zero3(); c.C0t1 = 1.0f;             /* world anchor: identity */
        if (isA || hasA) {
            const float4 op = B.plane(BP_POSR)[other], oq = B.plane(BP_QUAT)[other];
            pOther = mk3(op.x, op.y, op.z); wOther = mk3(oq.x, oq.y, oq.z); c.C0t1 = oq.w;
        }
        oinv = 0.0f; fric = 0.0f; c.penetration = 0.0f; c.stick = false;
    };
    /* everything else a task needs that does not come from another wave of this launch: the tables the earlier kernels of this step
     * wrote (first iteration), or what this sub-group parked after the body's previous task (several rounds, later iterations) */
    auto load_rest = [&]() __attribute__((always_inline)) {
        AVBD_MARK(g_load_rest);
        fixedN = false;
        if (SHARDK == 0 && MULTI && X.depCnt != nullptr) ctrO = (has && moving) ? X.rankOf[other] : 0;   /* (early: the signal behind the publish must not wait for it) */
        if (MULTI && it > 0) {
            /* a later task of a body that has been parked: ONE contiguous record per entry for what its tasks change (64 B: normal +
             * tangent memo, lambda, penalty, stick), one for what they do not (80 B: anchors, bias terms, friction, the partner's inverse
             * mass and — partners nobody sweeps — its contact point), one per body (160 B).  The planes of the Manifold tables cost a
             * cache line per field and contact: the pass was bound by those transactions, not by bytes. */
            if (JOINTS && has && isJoint) load_joint();
            else if (has) {
                const size_t e = (size_t)(eBeg + gl);
                const float4 k0 = X.parkC[e * 5], k1 = X.parkC[e * 5 + 1], k2 = X.parkC[e * 5 + 2], k3 = X.parkC[e * 5 + 3], k4 = X.parkC[e * 5 + 4];
                const float4 s0 = X.parkS[e * 4], s1 = X.parkS[e * 4 + 1], s2 = X.parkS[e * 4 + 2], s3 = X.parkS[e * 4 + 3];
                c.rA = mk3(k0.x, k0.y, k0.z); c.C0n = k0.w; c.rB = mk3(k1.x, k1.y, k1.z); c.C0t1 = k1.w; c.C0t2 = k2.x; fric = k2.y; oinv = k2.z;
                pOther = mk3(k3.x, k3.y, k3.z); wOther = mk3(k4.x, k4.y, k4.z);
                /* (w: stick | normal is a fixed point of its normalisation << 1 — what one body per sub-group keeps in a register for the whole launch:
                 * without it every task replays one or two normalisations per contact, square root and division included) */
                const int fl = (int)s0.w;
                c.normal = mk3(s0.x, s0.y, s0.z); c.stick = (fl & 1) != 0; fixedN = (fl & 2) != 0; c.u1 = mk3(s1.x, s1.y, s1.z); c.u2 = mk3(s2.x, s2.y, s2.z);
                c.lambda[0] = s1.w; c.lambda[1] = s2.w; c.lambda[2] = s3.x; c.penalty[0] = s3.y; c.penalty[1] = s3.z; c.penalty[2] = s3.w;
                c.penetration = 0.0f;                                      /* (an output of every computeConstraint call; the last task stores it) */
                if (SHARD) { remote = !moving && oinv > 0.0f; offR = remote ? Z.row[other] * 32 : 0; }
            }
            if (SHARD && valid) ownRow = Z.row[body] * 32;
            if (valid && gl < 10) reinterpret_cast<float4*>(cst)[gl] = X.parkB[(size_t)body * 10 + gl];
            wave_sync();
            if (valid) {
                pos = mk3(cst[32], cst[33], cst[34]); radius = cst[35]; rot = mkq(cst[36], cst[37], cst[38], cst[39]);
                invMass = cst[11];
            }
            return;
        }
        if (JOINTS && has && isJoint) load_joint();
        else if (has) {
            oinv = massPlane[other].y;
            fric = M.fric[idx >> 2];
            load_contact<false>(M, idx, c);                           /* the tables the earlier kernels of this step wrote */
            const float4 op = B.plane(BP_POSR)[other], oq = B.plane(BP_QUAT)[other];   /* predicted pose; constant for unswept partners */
            wOther = qrotate(mkq(oq.x, oq.y, oq.z, oq.w), isA ? c.rB : c.rA);
            pOther = mk3(op.x, op.y, op.z) + wOther;
            if (SHARD) { remote = !moving && oinv > 0.0f; offR = remote ? Z.row[other] * 32 : 0; }
            if (MULTI) {                                               /* what the later tasks of this entry read instead of the tables */
                const size_t e = (size_t)(eBeg + gl);
                X.parkC[e * 5] = make_float4(c.rA.x, c.rA.y, c.rA.z, c.C0n); X.parkC[e * 5 + 1] = make_float4(c.rB.x, c.rB.y, c.rB.z, c.C0t1);
                X.parkC[e * 5 + 2] = make_float4(c.C0t2, fric, oinv, 0.0f);
                X.parkC[e * 5 + 3] = make_float4(pOther.x, pOther.y, pOther.z, 0.0f); X.parkC[e * 5 + 4] = make_float4(wOther.x, wOther.y, wOther.z, 0.0f);
            }
        }
        if (SHARD && valid) ownRow = Z.row[body] * 32;
        if (valid) {
            const float4 pp = B.plane(BP_POSR)[body], pq = B.plane(BP_QUAT)[body];
            radius = pp.w;
            pos = mk3(pp.x, pp.y, pp.z); rot = mkq(pq.x, pq.y, pq.z, pq.w);
            invMass = massPlane[body].y;
        }
        if (valid && gl == 0) {                                      /* constants of the body that every task needs again */
            const float4 snp = B.plane(BP_NPOS)[body], snq = B.plane(BP_NQUAT)[body], smp = massPlane[body];
            float4* c4 = reinterpret_cast<float4*>(cst);
            c4[0] = make_float4(snp.x, snp.y, snp.z, smp.x); c4[1] = snq;
            for (int k = 0; k < 3; ++k) {
                const float4 ik = B.plane(BP_I0 + k)[body], jk = B.plane(BP_J0 + k)[body];
                c4[2 + k] = make_float4(ik.x, ik.y, ik.z, k == 0 ? smp.y : 0.0f); c4[5 + k] = make_float4(jk.x, jk.y, jk.z, 0.0f);
            }
            if (MULTI) for (int k = 0; k < 8; ++k) X.parkB[(size_t)body * 10 + k] = c4[k];
        }
        wave_sync();
    };
    /* several rounds: park the body's state until its next task (its last task has written the step's final state already) */
    auto park_body = [&]() __attribute__((always_inline)) {
        AVBD_MARK(g_park);
        if (JOINTS && has && isJoint) {
            if (owned) store_joint_half(gf[idx >> 1], idx & 1, c, false);      /* the other endpoint takes lambda / penalty from the dual records */
        } else if (has) {
            int slotF = eBeg + gl;
            asm volatile("" : "+v"(slotF));
            float4* ps = X.parkS + (size_t)slotF * 4;
            ps[0] = make_float4(c.normal.x, c.normal.y, c.normal.z, (float)((c.stick ? 1 : 0) | (fixedN ? 2 : 0)));
            ps[1] = make_float4(c.u1.x, c.u1.y, c.u1.z, c.lambda[0]);
            ps[2] = make_float4(c.u2.x, c.u2.y, c.u2.z, c.lambda[1]);
            ps[3] = make_float4(c.lambda[2], c.penalty[0], c.penalty[1], c.penalty[2]);
        }
        if (gl == 0) { X.parkB[(size_t)body * 10 + 8] = pack3(pos, radius); X.parkB[(size_t)body * 10 + 9] = packq(rot); }
    };

    for (int r = lane; r < kResRows; r += 64) { S[5 * kResRows + r] = 0.0f; S[6 * kResRows + r] = -0.0f; }
    if (lane < 4) S[kGrpSSlots * kResRows + kGrpOnesFloat + lane] = 1.0f;

    /* ---- what the first half of a task leaves for the second ---- */
    GAcc a0, a1;
    f3 wOwn = zero3();
    bool needP = false, needV = false, needD = false, okP = true, okV = true, okD = true;
    bool needR = false, okR = true; unsigned sR = 0;                 /* SHARD: the all-gather behind the previous iteration */
    unsigned sP = 0, sV = 0, sD = 0;
    ContactPre pre; f3 G[3];
    v4u p0 = {0, 0, 0, 0}, p1 = p0, p2 = p0, vd = p0, d1 = p0;   /* vd: the visit record of an owned contact, or the first dual record of another one */
    pre.n = pre.t1 = pre.t2 = zero3(); pre.biasN = pre.biasT1 = pre.biasT2 = pre.cap = 0.0f;
    G[0] = G[1] = G[2] = zero3();
    a0.opA = a0.opB = 0; a0.mode = 3; a0.gy = -1; a0.acc = 0.0f; a1 = a0;

#ifdef AVBD_DF_STATS
    unsigned long long stat[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned dbgTasks = 0, dbgOldest = 0;                            /* several bodies per sub-group: tasks run | of which the oldest of their sub-group */
    unsigned long long tprev = __builtin_amdgcn_s_memrealtime();
#define GS_STAMP(i) do { __builtin_amdgcn_s_waitcnt(0); unsigned long long _n = __builtin_amdgcn_s_memrealtime(); stat[i] += _n - tprev; tprev = _n; } while (0)
#define GS_STAMP_ND(i) do { unsigned long long _n = __builtin_amdgcn_s_memrealtime(); stat[i] += _n - tprev; tprev = _n; } while (0)   /* without draining memory operations */
#else
#define GS_STAMP(i) do { } while (0)
#define GS_STAMP_ND(i) do { } while (0)
#endif

    /* which records of other tasks this task reads, and the stamps they must carry (from the header and `it` alone) */
    auto set_needs = [&]() __attribute__((always_inline)) {
        const unsigned stampNow = X.base + (unsigned)it + 1u;
        needP = false; needV = false; needD = false; sP = 0; sV = 0; sD = 0;
        if (EXACT && it >= itStop) {                                   /* the task behind the last iteration: the owner's last dual */
            if (finalD) { needD = true; sD = X.base + (unsigned)itStop; }
        } else if (has && moving) {
            if (owned) { needV = !(JOINTS && isJoint); sV = stampNow; needP = true; sP = stampNow; }   /* (a joint's first visit mutates nothing: no visit record) */
            else if (it > 0) { needD = true; sD = X.base + (unsigned)it; needP = true; sP = sD; }
        }
        okP = !needP; okV = !needV; okD = !needD;
    };
    auto prepare_task = [&]() __attribute__((always_inline)) {
        AVBD_MARK(g_prepare);
        const float alpha = P.postStabilize ? (it < P.iterations ? 1.0f : 0.0f) : P.alpha;
        BodyLocal L;
        L.pos = pos; L.rot = rot; L.invMass = invMass; L.invDt2 = invDt2;
        {
            L.npos = mk3(cst[0], cst[1], cst[2]); L.mass = cst[3]; L.nrot = mkq(cst[4], cst[5], cst[6], cst[7]);
            const m33 Ib = mkm(mk3(cst[8], cst[9], cst[10]), mk3(cst[12], cst[13], cst[14]), mk3(cst[16], cst[17], cst[18]));
            const m33 Jb = mkm(mk3(cst[20], cst[21], cst[22]), mk3(cst[24], cst[25], cst[26]), mk3(cst[28], cst[29], cst[30]));
            world_inertia_group(gl, rot, Ib, Jb, xch, L.Iw, L.invIw);     /* rigid.cpp:51-59 */
        }
        GS_STAMP(10);
        AVBD_MARK(g_prep_gacc);
        a0 = make_gacc(gl < 27 ? gl : 63, L);
        a1 = make_gacc((cls == 0 && gl + 16 < 27) ? gl + 16 : 63, L);
        wOwn = zero3();
        if (!MULTI) set_needs();                                      /* (several bodies per sub-group: set, and polled, when the body was examined) */
        AVBD_MARK(g_prep_replay);
        if (JOINTS && has && isJoint) {
            /* Jacobians of this body's three rows (joint.cpp:108-139): functions of its own orientation only */
            JointRows jr0; jr0.rA = c.rA; jr0.rB = c.rB; jr0.q0 = mkq(c.normal.x, c.normal.y, c.normal.z, c.u1.x); jr0.fixAngular = c.C0t2 != 0.0f;
            for (int k = 0; k < 3; ++k) {
                f3 Jl, Ja;
                joint_compute_derivatives(jr0, isA, false, rot, (idx & 1) * 3 + k, Jl, Ja);
                G[k] = zero3();
                float* jr = J + lane * 3 + k;                          /* Jt[column][row] */
                jr[0] = Jl.x; jr[kResRows] = Jl.y; jr[2 * kResRows] = Jl.z; jr[3 * kResRows] = Ja.x; jr[4 * kResRows] = Ja.y; jr[5 * kResRows] = Ja.z;
            }
        } else if (has) {
            wOwn = qrotate(rot, isA ? c.rA : c.rB);                    /* manifold.cpp:27 and :265 use the same product */
            if (moving) {
                if (owned) replay_normal<MULTI>(c, fixedN);          /* (<MULTI>: the unit-length test of the basis memo on the bits, see contact_basis_memo) */
                else if (it > 0) { replay_normal<MULTI>(c, fixedN); replay_normal<MULTI>(c, fixedN); }
            }
            AVBD_MARK(g_prep_pre);
            ContactBasis db;
            contact_constraint_pre<MULTI>(c, alpha, isA ? invMass : oinv, isA ? oinv : invMass, pre, db);
            AVBD_MARK(g_prep_deriv);
            f3 Jl[3], Ja[3];
            contact_compute_derivatives_w(db, isA, wOwn, Jl, Ja);
            for (int k = 0; k < 3; ++k) {
                G[k] = abs3(cross3(Ja[k], L.invIw * Ja[k]));          /* solver.cpp:393-397, to be scaled by |f| */
                float* jr = J + lane * 3 + k;                          /* Jt[column][row] */
                jr[0] = Jl[k].x; jr[kResRows] = Jl[k].y; jr[2 * kResRows] = Jl[k].z; jr[3 * kResRows] = Ja[k].x; jr[4 * kResRows] = Ja[k].y; jr[5 * kResRows] = Ja[k].z;
            }
        }
        if (SHARD && !MULTI) { needR = has && remote && it > 0; sR = X.base + (unsigned)it; okR = !needR; }
        if (!MULTI && cls == 0) {                                     /* the seeds of the 27 sums, for the wave-wide accumulation */
            if (gl < 27) accs[gl] = a0.acc;
            if (gl + 16 < 27) accs[gl + 16] = a1.acc;
        }
        wave_sync();
        GS_STAMP(11);
    };

    /* EXACT, the task behind a body's last iteration: a contact across a cut lives on both ranks of the cut; the rank that sweeps its
     * owner has run the last dual (solver.cpp:411-430) — lambda, penalty and the stick flag go into this rank's copy from the
     * owner's record, and the stored normal takes the two re-normalisations the owner's last visit and dual made (replay_normal),
     * so that both copies enter the next step's Manifold::initialize with the same bits (manifold.cpp:76-84). */
    auto final_store = [&]() __attribute__((always_inline)) {
        if (JOINTS && finalD && isJoint) {
            /* a Joint across a cut: every rank holds the GForce record; the half's rows on this rank's copy are the owner's last dual */
            c.lambda[0] = bitsf(vd.x); c.lambda[1] = bitsf(vd.y); c.lambda[2] = bitsf(vd.z);
            c.penalty[0] = bitsf(d1.x); c.penalty[1] = bitsf(d1.y); c.penalty[2] = bitsf(d1.z);
            store_joint_half(gf[idx >> 1], idx & 1, c, false);
        } else
        if (finalD) {
            c.lambda[0] = bitsf(vd.x); c.lambda[1] = bitsf(vd.y); c.lambda[2] = bitsf(vd.z);
            c.penalty[0] = bitsf(d1.x); c.penalty[1] = bitsf(d1.y); c.penalty[2] = bitsf(d1.z);
            c.stick = (d1.w & 1u) != 0;
            replay_normal<MULTI>(c, fixedN); replay_normal<MULTI>(c, fixedN);
            int idxF = idx;
            asm volatile("" : "+v"(idxF));
            store_contact_dual<false>(M, idxF, c, false);
        }
    };

    bool allOnG = true;                                              /* what finish_visit leaves for the accumulation */
    auto finish_visit = [&]() __attribute__((always_inline)) {
        AVBD_MARK(g_visit);
        const float alpha = P.postStabilize ? (it < P.iterations ? 1.0f : 0.0f) : P.alpha;
        const bool doDual = it < P.iterations;
        const unsigned stampNow = X.base + (unsigned)it + 1u;
#ifdef AVBD_DF_STATS
        if (gl == 0) race_note_consumed(errWord - 2 * B.cap, body, it);
#endif
        if (needP) {
            const f3 opos = mk3(bitsf(p0.x), bitsf(p0.y), bitsf(p0.z)); const q4 orot = mkq(bitsf(p1.x), bitsf(p1.y), bitsf(p1.z), bitsf(p2.x));
            if (JOINTS && isJoint) { pOther = opos; wOther = mk3(orot.x, orot.y, orot.z); c.C0t1 = orot.w; }
            else {
            wOther = qrotate(orot, isA ? c.rB : c.rA);
            pOther = opos + wOther;
            }
        }
        if (SHARD && needR) {
            /* the partner's pose after iteration it - 1, as every rank received it (Jacobi across the cut) */
            const v4u g0 = (it & 1) ? xld(rG0, offR) : xld(rG1, offR), g1 = (it & 1) ? xld(rG0, offR + 16) : xld(rG1, offR + 16);
            const f3 opos = mk3(bitsf(g0.x), bitsf(g0.y), bitsf(g0.z)); const q4 orot = mkq(bitsf(g1.x), bitsf(g1.y), bitsf(g1.z), bitsf(g1.w));
            wOther = qrotate(orot, isA ? c.rB : c.rA);
            pOther = opos + wOther;
        }
        if (needV) { c.lambda[1] = bitsf(vd.x); c.lambda[2] = bitsf(vd.y); c.stick = vd.z != 0; }
        if (needD) {
            c.lambda[0] = bitsf(vd.x); c.lambda[1] = bitsf(vd.y); c.lambda[2] = bitsf(vd.z);
            c.penalty[0] = bitsf(d1.x); c.penalty[1] = bitsf(d1.y); c.penalty[2] = bitsf(d1.z);
            c.stick = (d1.w & 1u) != 0;
        }
        /* ---- primal visit (solver.cpp:344-409) ---- */
        bool rowsOn = true;
        if (JOINTS && has && isJoint) {
            /* Joint::computeConstraint (joint.cpp:68-106) + the row forces of solver.cpp:372-385 for this half */
            JointRows jr1; jr1.rA = c.rA; jr1.rB = c.rB; jr1.q0 = mkq(c.normal.x, c.normal.y, c.normal.z, c.u1.x);
            const q4 orot = mkq(wOther.x, wOther.y, wOther.z, c.C0t1);
            if (isA) joint_compute_constraint(jr1, hasA, pos, rot, pOther, orot);
            else     joint_compute_constraint(jr1, hasA, pOther, orot, pos, rot);
            const int jp = (idx & 1) * 3;
            for (int k = 0; k < 3; ++k) {
                const float stiff = k == 0 ? c.u1.y : (k == 1 ? c.u1.z : c.u2.x), motor = k == 0 ? c.u2.y : (k == 1 ? c.u2.z : c.C0n);
                c.C[k] = jr1.C[jp + k];
                const float lw = (stiff == FLT_MAX) ? c.lambda[k] : 0.0f;
                const float desired = c.penalty[k] * c.C[k] + lw + motor;
                const float f = rclamp(desired, -FLT_MAX, FLT_MAX);
                const bool penOk = (c.penalty[k] > 0.0f) & (fabsf(c.penalty[k]) <= FLT_MAX);
                const int r = lane * 3 + k;
                /* not a Manifold row: no geometric-stiffness term (solver.cpp:393) — every aa entry adds -0.0f, the identity of IEEE addition, where a
                 * Manifold row adds its gyroscopic component (diagonal) or the +0.0f of the reference's diagonal() matrix (off the diagonal) */
                S[r] = f; S[kResRows + r] = c.penalty[k]; S[2 * kResRows + r] = -0.0f; S[3 * kResRows + r] = -0.0f; S[4 * kResRows + r] = -0.0f; S[5 * kResRows + r] = -0.0f;
                rowsOn &= penOk;
            }
        } else if (has) {
            const f3 pOwn = pos + wOwn;
            contact_constraint_post(c, pre, isA ? pOwn : pOther, isA ? pOther : pOwn, fric);
            if (!owned) { v4u r = {fbits(c.lambda[1]), fbits(c.lambda[2]), c.stick ? 1u : 0u, stampNow}; xstA(rV, offV, r); }
            for (int k = 0; k < 3; ++k) {
                const float desired = c.penalty[k] * c.C[k] + c.lambda[k] + 0.0f;     /* solver.cpp:379-381 with stiffness FLT_MAX, motor 0 */
                const float f = rclamp(desired, c.fmin[k], c.fmax[k]);
                const f3 gyro = G[k] * fabsf(f);
                const bool penOk = (c.penalty[k] > 0.0f) & (fabsf(c.penalty[k]) <= FLT_MAX);   /* solver.cpp:387 */
                const int r = lane * 3 + k;
                S[r] = f; S[kResRows + r] = c.penalty[k]; S[2 * kResRows + r] = gyro.x; S[3 * kResRows + r] = gyro.y; S[4 * kResRows + r] = gyro.z;
                if (JOINTS) S[5 * kResRows + r] = 0.0f;                /* (the row of the lane may have been a joint's in an earlier task) */
                rowsOn &= penOk;
            }
        }
        allOnG = (__ballot(rowsOn) & gMask) == gMask;
        wave_sync();
        GS_STAMP(7);
        (void)alpha; (void)doDual;
    };
    /* the ordered sums by the lanes of the sub-group itself: two entries per lane in a 16-lane sub-group */
    auto finish_accumulate = [&]() __attribute__((always_inline)) {
        AVBD_MARK(g_accumulate);
        if (cls == 0) { if (allOnG) grp_accumulate<true, true>(a0, a1, J, S, rowBase, nRows); else grp_accumulate<false, true>(a0, a1, J, S, rowBase, nRows); }
        else          { if (allOnG) grp_accumulate<true, false>(a0, a1, J, S, rowBase, nRows); else grp_accumulate<false, false>(a0, a1, J, S, rowBase, nRows); }
        if (gl < 27) accs[gl] = a0.acc;
        if (cls == 0 && gl + 16 < 27) accs[gl + 16] = a1.acc;
        wave_sync();
    };
    bool depSig = false;                                             /* several bodies per sub-group, arrival counters (see there): this task signals its partners */
    auto finish_solve = [&]() __attribute__((always_inline)) {
        const float alpha = P.postStabilize ? (it < P.iterations ? 1.0f : 0.0f) : P.alpha;
        const bool doDual = it < P.iterations;
        const unsigned stampNow = X.base + (unsigned)it + 1u;
        AVBD_MARK(g_solve);
        GS_STAMP(8);
        BodyLocal L; L.pos = pos; L.rot = rot;
        f3 p; q4 q;
        solve_pose(accs, xch, gl, L, p, q);
#ifdef AVBD_DF_STATS
        race_check_publish(errWord - 2 * B.cap, cntW, body, other, has && moving, it);
#endif
        if (gl == 0) {
            sanitize3(p, cntW, warn_key(1, it, body, 0), body, WK_POS);
            sanitizeq(q, cntW, warn_key(1, it, body, 1), body, WK_ROT);
            const int o = body * 48;
            v4u r0 = {fbits(p.x), fbits(p.y), fbits(p.z), stampNow}, r1 = {fbits(q.x), fbits(q.y), fbits(q.z), stampNow}, r2 = {fbits(q.w), 0u, 0u, stampNow};
            xstA(rP, o, r0); xstA(rP, o + 16, r1); xstA(rP, o + 32, r2);
            if (SHARD) {                                              /* this rank's row of the all-gather behind iteration `it` */
                v4u s0 = {fbits(p.x), fbits(p.y), fbits(p.z), fbits(radius)}, s1 = {fbits(q.x), fbits(q.y), fbits(q.z), fbits(q.w)};
                if (it & 1) { xst(rG1, ownRow, s0); xst(rG1, ownRow + 16, s1); } else { xst(rG0, ownRow, s0); xst(rG0, ownRow + 16, s1); }
            }
        }
        /* arrivals at the earlier-created partners (their task `it` reads this pose and the visit record, both on their way): as early as possible,
         * the hop from a publish to the consumer's examination is what the launch is a chain of */
        const bool firstOfForce = (JOINTS && isJoint) ? (idx & 1) == 0 : (idx & 3) == 0;   /* one arrival per manifold / joint */
        if (SHARDK == 0 && MULTI && depSig && has && moving && !owned && firstOfForce)
            __hip_atomic_fetch_add(X.depCnt + ctrO, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pos = mk3(grp_bcast(p.x, gBase), grp_bcast(p.y, gBase), grp_bcast(p.z, gBase));
        rot = mkq(grp_bcast(q.x, gBase), grp_bcast(q.y, gBase), grp_bcast(q.z, gBase), grp_bcast(q.w, gBase));
#ifdef AVBD_DF_STATS
        if (gl == 0 && X.privN) (reinterpret_cast<unsigned*>(X.privN) + ((size_t)it * B.n + (size_t)body) * 4)[1] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
        AVBD_MARK(g_dual);
        GS_STAMP(9);
        /* ---- dual update of the joint halves this body owns (solver.cpp:411-430 with Joint rows) ---- */
        if (JOINTS && doDual && has && owned && isJoint) {
            JointRows jr2; jr2.rA = c.rA; jr2.rB = c.rB; jr2.q0 = mkq(c.normal.x, c.normal.y, c.normal.z, c.u1.x); jr2.fixAngular = c.C0t2 != 0.0f;
            const q4 orot = mkq(wOther.x, wOther.y, wOther.z, c.C0t1);
            const q4 rotA = isA ? rot : orot, rotB = isA ? orot : rot;
            if (isA) joint_compute_constraint(jr2, hasA, pos, rot, pOther, orot);
            else     joint_compute_constraint(jr2, hasA, pOther, orot, pos, rot);
            const int jp = (idx & 1) * 3;
            for (int k = 0; k < 3; ++k) {
                const int row = jp + k;
                c.C[k] = jr2.C[row];
                const float stiff = k == 0 ? c.u1.y : (k == 1 ? c.u1.z : c.u2.x);
                if (stiff != FLT_MAX) continue;
                const float lu = rclamp(c.penalty[k] * c.C[k] + c.lambda[k], -FLT_MAX, FLT_MAX);
                const bool active = lu > -FLT_MAX && lu < FLT_MAX;
                c.lambda[k] = lu;
                if (active) {
                    float lw = 0.0f, aw = 0.0f;
                    f3 Jl, Ja;
                    if (hasA) { joint_compute_derivatives(jr2, true, false, rotA, row, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja); }
                    joint_compute_derivatives(jr2, false, false, rotB, row, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja);
                    const float betaRow = row_penalty_gain(lw, aw, P.beta);
                    c.penalty[k] = rmin(c.penalty[k] + betaRow * fabsf(c.C[k]), kPenaltyMax);
                }
            }
            if (moving) {
                v4u r0 = {fbits(c.lambda[0]), fbits(c.lambda[1]), fbits(c.lambda[2]), stampNow};
                v4u r1 = {fbits(c.penalty[0]), fbits(c.penalty[1]), fbits(c.penalty[2]), stampNow << 1};
                xstA(rD, offD, r0); xstA(rD, offD + 16, r1);
                if (SHARDK == 0 && MULTI && depSig && firstOfForce && it + 1 < itStop)
                    __hip_atomic_fetch_add(X.depCnt + ctrO, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        /* ---- dual update of the contacts this body owns (solver.cpp:411-430) ---- */
        if (doDual && has && owned && !(JOINTS && isJoint)) {
            const f3 wNew = qrotate(rot, isA ? c.rA : c.rB);
            const f3 pNew = pos + wNew;
            dual_contact_pts<MULTI>(c, isA ? pNew : pOther, isA ? wNew : wOther, isA ? invMass : oinv, isA ? pOther : pNew, isA ? wOther : wNew,
                             isA ? oinv : invMass, fric, P, alpha);
            if (moving) {
                v4u r0 = {fbits(c.lambda[0]), fbits(c.lambda[1]), fbits(c.lambda[2]), stampNow};
                v4u r1 = {fbits(c.penalty[0]), fbits(c.penalty[1]), fbits(c.penalty[2]), (stampNow << 1) | (c.stick ? 1u : 0u)};
                xstA(rD, offD, r0); xstA(rD, offD + 16, r1);
                /* ... and at the later-created partner, whose task it + 1 reads this dual */
                if (SHARDK == 0 && MULTI && depSig && (idx & 3) == 0 && it + 1 < itStop)
                    __hip_atomic_fetch_add(X.depCnt + ctrO, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        /* ---- the body's last task: final state back to the tables (plain stores; the next kernel of the stream reads them) ---- */
        AVBD_MARK(g_final_store);
        if (it == itStop - 1) {
            /* (opaque copies of the indices: the nine table addresses of a contact are formed here, once, instead of living in
             * registers through the whole loop) */
            int bodyF = body, idxF = idx;
            asm volatile("" : "+v"(bodyF), "+v"(idxF));
            if (gl == 0) { B.plane(BP_POSR)[bodyF] = pack3(pos, radius); B.plane(BP_QUAT)[bodyF] = packq(rot); }
            if (JOINTS && has && owned && isJoint) store_joint_half(gf[idxF >> 1], idxF & 1, c, true);
            else if (has && owned) store_contact_dual<false>(M, idxF, c, true);
        }
    };

    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (SHARDK != 0 && Z.span && w == 0 && lane == 0) Z.span[0] = t0;    /* profiling of a sharded launch: its span from the device's own clock (see shard_launch_persistent) */
    unsigned long long tIdle = t0;                                   /* when this wave last made progress (bounds the spin, not the launch) */
    unsigned spins = 0;
    /* the poll of a prepared task: the records ARE the payload (avbd_resident.h); only lanes that still miss something load */
    auto poll_issue = [&]() __attribute__((always_inline)) {
        const bool lp = !okP, lv = !okV, ld = !okD;
        if constexpr (EXACT) {
            if (Z.nPeers > 1) {
                /* one round of loads per window some lane reads from (a slab has two neighbours: one to three rounds) */
                unsigned long long todo = __ballot(lp | lv | ld);
                while (todo) {
                    const int pr = __builtin_amdgcn_readlane(peer, __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1));
                    const __amdgpu_buffer_rsrc_t qP = xrsrc(Z.peerPose[pr], Z.xPoseBytes), qV = xrsrc(Z.peerVisit[pr], Z.xVisitBytes), qD = xrsrc(Z.peerDual[pr], Z.xDualBytes);
                    const bool m = peer == pr;
                    if (lp && m) { p0 = xldA(qP, offP); p1 = xldA(qP, offP + 16); p2 = xldA(qP, offP + 32); }
                    if (lv && m) vd = xldA(qV, offV);
                    if (ld && m) { vd = xldA(qD, offD); d1 = xldA(qD, offD + 16); }
                    todo &= ~__ballot(m);
                }
                return;
            }
        }
        if (lp) { p0 = xldA(rP, offP); p1 = xldA(rP, offP + 16); p2 = xldA(rP, offP + 32); }
        if (lv) vd = xldA(rV, offV);
        if (ld) { vd = xldA(rD, offD); d1 = xldA(rD, offD + 16); }
    };
    /* the stamp word of a pose record some lanes watch (several bodies per sub-group), from the window of the rank that writes it */
    auto watch_pose = [&](bool want, int off, int pk, unsigned& val) __attribute__((always_inline)) {
        if constexpr (EXACT) {
            if (Z.nPeers > 1) {
                unsigned long long todo = __ballot(want);
                while (todo) {
                    const int pr = __builtin_amdgcn_readlane(pk, __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1));
                    const __amdgpu_buffer_rsrc_t qP = xrsrc(Z.peerPose[pr], Z.xPoseBytes);
                    const bool m = want && pk == pr;
                    if (m) val = xldwA(qP, off);
                    todo &= ~__ballot(m);
                }
                return;
            }
        }
        if (want) val = xldwA(rP, off);
    };
    auto poll_check = [&]() __attribute__((always_inline)) {
        const bool lp = !okP, lv = !okV, ld = !okD;
        asm volatile("" ::: "memory");
        if (lp) okP = stamp_reached(p0.w, sP) & stamp_reached(p1.w, sP) & stamp_reached(p2.w, sP);
        if (lv) okV = stamp_reached(vd.w, sV);
        if (ld) okD = stamp_reached(vd.w, sD) & stamp_reached(d1.w >> 1, sD);
        if (SHARD && !okR) okR = stamp_reached(xldw(rGW, 0), sR);
    };
    auto poll_records = [&]() __attribute__((always_inline)) { poll_issue(); poll_check(); };
    int reported = 0;                                                 /* SHARD, one body per sub-group: iterations this wave has counted in Z.done */
    int dbgBody = -1, dbgIt = 0, dbgOff = 0, dbgStamp = 0, dbgSeen = 0;   /* several bodies per sub-group: what lane k's body waits for (time-out report) */
    int itEndOne = itStop;                                            /* one body per sub-group: tasks of the body (EXACT: + the one behind the last iteration) */
    bool grpWaiting = true;                                           /* several bodies per sub-group: the sub-group still has a body with tasks left */
    auto back_off = [&]() __attribute__((always_inline)) -> bool {     /* false: the launch is being abandoned */
        if (sleepSel == 0) __builtin_amdgcn_s_sleep(1); else if (sleepSel == 1) __builtin_amdgcn_s_sleep(2);
        else if (sleepSel == 2) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(8);
        /* (`spins` and `tIdle` are per sub-group — a task resets them for its lanes only —, so the lanes of a wave come here at
         * different counts: the check itself is made by the whole wave, and the wave leaves as one) */
        const bool chk = (++spins & 1023u) == 0;
        if (__any(chk)) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (chk && spins == 1024u) tIdle = now - 1024ull * 30ull;  /* first check since the last task: the wait began about here */
            const bool waiting = MULTI ? grpWaiting : (valid && it < itEndOne);   /* (a sub-group that has finished its last task waits for nothing) */
            const bool dead = (chk && waiting && now - tIdle > kSpinTimeoutTicks) ||
                              __hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (__any(dead)) {
                {   /* what the first wave to give up was waiting for (errWord[1..15]; read by the host for its message) */
                    const bool mine = chk && waiting && now - tIdle > kSpinTimeoutTicks;
                    const unsigned long long mT = __ballot(mine), mP = __ballot(mine && !okP), mV = __ballot(mine && !okV), mD = __ballot(mine && !okD), mR = __ballot(mine && !okR);
                    const int stuck = mT ? __ffsll((long long)((mP | mV | mD | mR) ? (mP | mV | mD | mR) : mT)) - 1 : 0;
                    const int sb = __shfl(body, stuck, 64), so = __shfl(other, stuck, 64), si = __shfl(it, stuck, 64);
                    int mPick = 0, mB = 0, mO = 0, mI = 0, mS = 0, mW = 0;
                    if (MULTI) {
                        /* the first body of the first timed-out sub-group that still has tasks: body, iteration | last << 8, watched offset, stamps minus base */
                        const int g0 = mT ? ((__ffsll((long long)mT) - 1) & ~15) : 0;
                        const unsigned long long act = __ballot(dbgBody >= 0 && (dbgIt & 255) < (dbgIt >> 8)) & (0xffffull << g0);
                        mPick = act ? __ffsll((long long)act) - 1 : g0;
                        mB = __shfl(dbgBody, mPick, 64); mO = __shfl(dbgOff, mPick, 64); mI = __shfl(dbgIt, mPick, 64); mS = __shfl(dbgStamp, mPick, 64); mW = __shfl(dbgSeen, mPick, 64);
                    }
                    if (MULTI && lane == 0 && mT && __hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && atomicCAS(errWord + 1, 0, w + 1) == 0) {
                        errWord[2] = mB; errWord[3] = mO; errWord[4] = mI & 255; errWord[5] = mS; errWord[6] = mW; errWord[7] = mI >> 8; errWord[8] = mPick;
                        errWord[9] = 0; errWord[10] = 0; errWord[11] = 0; errWord[12] = 0; errWord[13] = (int)mT; errWord[14] = (int)(mT >> 32); errWord[15] = cls + 100;
                    } else
                    if (lane == 0 && mT && __hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && atomicCAS(errWord + 1, 0, w + 1) == 0) {
                        errWord[2] = sb; errWord[3] = so; errWord[4] = si; errWord[5] = (int)mP; errWord[6] = (int)(mP >> 32); errWord[7] = (int)mV; errWord[8] = (int)(mV >> 32);
                        errWord[9] = (int)mD; errWord[10] = (int)(mD >> 32); errWord[11] = (int)mR; errWord[12] = (int)(mR >> 32); errWord[13] = (int)mT; errWord[14] = (int)(mT >> 32); errWord[15] = cls;
                    }
                }
                if (!MULTI && X.privPose && valid) {
                    /* wait graph of the abandoned launch (debug: avbd_debug_wait_graph): per body {iteration, first partner an entry
                     * still waits for or -1, which records (1 pose, 2 visit, 4 dual, 8 gather), wave} */
                    const unsigned long long miss = __ballot(!(okP & okV & okD & okR) && it < itStop) & gMask;
                    const int f = miss ? __ffsll((long long)miss) - 1 : gBase;
                    const int fo = __shfl(other, f, 64);
                    const int fk = __shfl((okP ? 0 : 1) | (okV ? 0 : 2) | (okD ? 0 : 4) | (okR ? 0 : 8), f, 64);
                    if (gl == 0) X.privPose[(size_t)body * 2] = make_float4(__int_as_float(it), __int_as_float(miss ? fo : -1), __int_as_float(miss ? fk : 0), __int_as_float(w | (reported << 16)));
                    if (SHARD && miss && lane == f && !okR) {
                        /* the gather word as four kinds of access see it */
                        const unsigned a = xldw(rGW, 0);
                        const unsigned b2 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rGW, 0, 0, 17);
                        const unsigned c2 = __hip_atomic_load(Z.gatherWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const unsigned d2 = __hip_atomic_fetch_add(Z.gatherWord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        X.privPose[(size_t)body * 2 + 1] = make_float4(__int_as_float((int)(a - X.base)), __int_as_float((int)(b2 - X.base)), __int_as_float((int)(c2 - X.base)), __int_as_float((int)(d2 - X.base) | ((int)(sR - X.base) << 16)));
                    } else if (miss && lane == f) {
                        /* the stamp of the partner's last pose record as three kinds of load see it, and what the lane waits for */
                        const unsigned a = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rP, offP + 44, 0, 16);      /* sc1 (what the polls use) */
                        const unsigned b2 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rP, offP + 44, 0, 17);     /* sc0 sc1 */
                        const unsigned c2 = __hip_atomic_load(reinterpret_cast<const unsigned*>(X.pose) + (offP + 44) / 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        X.privPose[(size_t)body * 2 + 1] = make_float4(__int_as_float((int)(a - X.base)), __int_as_float((int)(b2 - X.base)), __int_as_float((int)(c2 - X.base)), __int_as_float((int)(sP - X.base)));
                    }
                }
                if (lane == 0) {
                    __hip_atomic_store(errWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    /* the exchange stream must not wait for iterations that will never complete */
                    if (SHARD && !Z.keepExchangeWaiting) __hip_atomic_fetch_max(Z.doneSignal, X.base + 63u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return false;
            }
        }
        return true;
    };

    if constexpr (!MULTI) {
        /* ---- one body per sub-group for the whole launch ---- */
        {
            const int b = group_unit_body(plan, byClass, cap, cls, wl, gBase >> (4 + cls), 0);
            int e0 = 0, ne = 0;
            if (b >= 0) { e0 = incStart[b]; ne = incStart[b + 1] - e0; }
            load_header(b, e0, ne, 0);
            load_rest();
        }
        if (valid) prepare_task();
        /* EXACT: a body with a contact whose owner another rank sweeps ends with one more task (final_store) */
        const int itEnd = itStop + ((EXACT && (__ballot(finalD) & gMask) != 0) ? 1 : 0);
        itEndOne = itEnd;
        GS_STAMP(3);
        /* The first half of a sub-group's next task (0.9 us of wave time) is deferred by one pass: the
         * wave first looks whether another of its sub-groups has become ready meanwhile (its own next task cannot be: it waits for
         * partners that wait for the pose just published). */
        bool needPrep = false;
        for (;;) {
            const bool active = valid && it < itEnd && !needPrep;
            if (__ballot(valid && it < itEnd) == 0) break;
            if (active) poll_records();
            const bool laneOk = !active || (okP & okV & okD & okR);
            const bool goAny = active && ((__ballot(laneOk) & gMask) == gMask);
            const bool goF = EXACT && goAny && it >= itStop;
            const bool go = goAny && !goF;
            GS_STAMP(0);
#ifdef AVBD_DF_STATS
            stat[14] += 1;
            if (__ballot(go)) { stat[5] += 1; stat[4] += (unsigned long long)__popcll(__ballot(go && gl == 0)); }
#endif
#ifdef AVBD_DF_STATS
            unsigned trecNone[4];
            unsigned* trec = X.privN ? reinterpret_cast<unsigned*>(X.privN) + ((size_t)it * B.n + (size_t)(body < 0 ? 0 : body)) * 4 : trecNone;   /* tools/debug_groups_critical.py */
            if (go && gl == 0) trec[0] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
            /* When ONE sub-group of the wave runs (the usual case: 1.1 tasks per pass) its 27 ordered sums go to 27 lanes of the WHOLE
             * wave, one chain of additions per lane instead of two per lane of the sub-group: everything an entry needs is in LDS
             * (seed, product line, S slots). */
            const unsigned long long goMask = __ballot(go);
            if (go) finish_visit();
            const unsigned long long leaders = goMask & 0x0001000100010001ull;
            if (cls == 0 && leaders != 0 && (leaders & (leaders - 1)) == 0) {
                const int gb = __builtin_amdgcn_readfirstlane(__ffsll((long long)leaders) - 1);        /* first lane of the running sub-group */
                const int nRowsG = __builtin_amdgcn_readlane(nRows, gb);
                const bool allOnW = __builtin_amdgcn_readlane(allOnG ? 1 : 0, gb) != 0;
                float* const accsW = S + kGrpSSlots * kResRows + (gb >> 4) * kGrpAccFloats;
                GAcc aw, none;
                grp_entry_ops(lane < 27 ? lane : 63, aw);
                aw.acc = accsW[lane < 27 ? lane : 0];
                none = aw;
                if (allOnW) grp_accumulate<true, false>(aw, none, J, S, gb * 3, nRowsG);
                else grp_accumulate<false, false>(aw, none, J, S, gb * 3, nRowsG);
                if (lane < 27) accsW[lane] = aw.acc;
                wave_sync();
            } else if (go) finish_accumulate();
            if (go) {
                finish_solve();
                ++it;
            }
            if (goF) { final_store(); ++it; spins = 0; }
#ifdef AVBD_DF_STATS
            if (go && gl == 0) { trec[2] = (unsigned)__builtin_amdgcn_s_memrealtime(); trec[3] = (unsigned)w; }
#endif
            GS_STAMP(1);
            bool prepNow = needPrep;                                   /* (one call site: the first half of a task is the largest piece of the kernel) */
            needPrep = false;
            if (go) {
                if (it < itStop) needPrep = true;                     /* (measured: 8 000 cubes 1.21 -> 1.15 ms per launch) */
                else if (EXACT && it < itEnd) set_needs();            /* what the task behind the last iteration waits for */
                spins = 0;
            }
            if (prepNow) prepare_task();
            if (SHARD && __ballot(go)) {
                /* Count finished iterations per WAVE: iteration k of this wave is complete when all its sub-groups are past it; one
                 * atomic per wave and iteration (not per task: 8 000 returning atomics on one word per iteration cost 20 % of the
                 * launch), issued once the rows have reached memory.  The wave that completes the rank's iteration raises the
                 * signal the exchange stream waits for. */
                int low = valid ? it : itStop;
                low = min(low, __shfl_xor(low, 16, 64)); low = min(low, __shfl_xor(low, 32, 64));
                low = __builtin_amdgcn_readfirstlane(low);
                if (reported < low) {
                    const unsigned mine = (unsigned)__popcll(__ballot(valid && gl == 0));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0)
                        for (int k = reported; k < low; ++k) {
                            const unsigned before = __hip_atomic_fetch_add(Z.done + k, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (before + mine == (unsigned)Z.nOwned) {
                                __hip_atomic_fetch_max(Z.doneSignal, X.base + (unsigned)k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                if (Z.span && k == itStop - 1) Z.span[1] = __builtin_amdgcn_s_memrealtime();   /* profiling: the rank's last iteration is complete */
                            }
                        }
                    reported = low;
                }
            }
            GS_STAMP(2);
            if (__ballot(goAny || needPrep) == 0 && !back_off()) return;
        }
    } else {
        /* ---- several bodies per sub-group, out of order: a sub-group runs whichever of its bodies has its inputs published —
         * the dependencies between tasks are carried by the stamped records alone, so any order that respects them gives the
         * same bits — oldest iteration first, so the head of the in-order list (which is what the rest of the launch may be
         * waiting for) is never passed over once it is ready.  Lane k of a sub-group keeps the bookkeeping of its k-th body: the
         * next iteration and ONE record to watch (the input that was missing when the body was last examined); a body is
         * examined in full only when that record has arrived.  Every pass of the wave runs the picks of all its sub-groups
         * together. ---- */
        int bodyK = -1, eBegK = 0, nEntK = 0, itK = itStop, blkOff = -1;
        int blkPeer = 0;                                              /* EXACT: the window the watched pose record lives in */
        int ctrK = 0;                                                 /* arrival counters: where lane k's body counts (its class-major rank) */
        /* ARRIVAL COUNTERS (one GPU): instead of watching ONE record of a body that was missing when it was last examined — and finding, after the
         * round trip that fetches records and parked state, that another one is missing too (two examinations out of five ended that way: 1.6 GB of
         * the 4.2 GB a 64 000-cube launch read, and a whole round trip of the wave's pass each) — lane k watches the number of partners that have
         * published what its body's next task reads.  A task that ends adds one to the counter of every moving partner whose NEXT task reads its
         * records: partner o > body (the body owns the contact, o is swept first): o's task it + 1 reads this dual; partner o < body: o's task `it`
         * reads this pose and visit record.  So task `it` of a body with nOwn later-created and nOth earlier-created moving partners (manifolds,
         * not contacts: the first contact of a manifold signals) is ready exactly when nOwn * (it + 1) + nOth * it arrivals have been counted —
         * none of the arrivals for task it + 1 can come before task `it` has run (each of them needs its records).  The counter is a hint that is
         * almost always right, not a proof: the adds are not ordered behind the record stores, so the stamps of the records still decide, and an
         * examination that finds one missing is simply repeated in the next pass. */
        constexpr bool DEPC = SHARDK == 0;
        const bool dep = DEPC && X.depCnt != nullptr && (pooled || X.depKeep || !(X.specMode != 2 && rounds <= X.specRounds));   /* (depKeep 0: classes that take a task in hand keep the watched records) */
        const __amdgpu_buffer_rsrc_t rC = xrsrc(dep ? (void*)X.depCnt : (void*)X.pose, dep ? (unsigned)cap * 12u : 16u);   /* three classes of `cap` ranks */
        unsigned depPerIt = 0, depNeed = 0;                           /* lane k: arrivals per iteration, arrivals its body's next task waits for */
        int itEndK = itStop;                                          /* EXACT: + the task that takes the owner's last dual (see the header table) */
        int reportedM = 0;                                            /* SHARD: iterations this wave has counted in Z.done */
        unsigned blkStamp = 0;
        if (gl < rounds) {
            bodyK = group_unit_body(plan, byClass, cap, cls, wl, gBase >> (4 + cls), gl, pooled);
            ctrK = cls * cap + group_unit_rank(plan, cls, wl, gBase >> (4 + cls), gl, pooled);
            if (bodyK >= 0) { eBegK = incStart[bodyK]; nEntK = incStart[bodyK + 1] - eBegK; itK = 0; }
        }
        /* the header table: three dependent loads per entry (incidence word -> pair -> is the partner swept?), four rounds in flight */
        for (int k0 = 0; k0 < rounds; k0 += 4) {
            int ent[4], oth[4], bk[4], e0s[4]; bool hk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + q < rounds ? k0 + q : rounds - 1;
                bk[q] = __shfl(bodyK, gBase + k, 64);
                const int e0 = __shfl(eBegK, gBase + k, 64), ne = __shfl(nEntK, gBase + k, 64);
                e0s[q] = e0;
                hk[q] = bk[q] >= 0 && gl < ne;
                ent[q] = hk[q] ? inc[e0 + gl] : 0;
            }
            bool hA[4];                                               /* JOINTS: the joint has a bodyA */
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                oth[q] = 0; hA[q] = true;
                if (JOINTS && hk[q] && inc_kind(ent[q]) == FK_JOINT) {
                    const GForce& f = gf[inc_index(ent[q]) >> 1];
                    hA[q] = f.bodyA >= 0;
                    oth[q] = inc_isA(ent[q]) ? f.bodyB : f.bodyA;          /* -1: the world */
                } else
                if (hk[q]) { const int2 pr = M.pair[inc_index(ent[q]) >> 2]; oth[q] = inc_isA(ent[q]) ? pr.y : pr.x; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int w1 = 0, rc = 0;
                if (hk[q]) {
                    const bool mv = (JOINTS && oth[q] < 0) ? false : (EXACT ? massPlane[oth[q]].y > 0.0f : B.solved[oth[q]] != 0);
                    w1 = (oth[q] < 0 ? 0 : oth[q]) | ((mv ? 1 : 0) << 24) | (((!mv || oth[q] > bk[q]) ? 1 : 0) << 25) | ((hA[q] ? 1 : 0) << 26);
                    if (EXACT) rc = Z.recIdx[e0s[q] + gl];
                }
                if (EXACT) {
                    /* a body with a contact whose owner another rank sweeps ends with one more task: the owner's last dual into this rank's copy */
                    const bool anyFinal = (__ballot(hk[q] && rc < 0) & gMask) != 0;
                    if (anyFinal && k0 + q < rounds && gl == k0 + q) itEndK = itStop + 1;
                }
                if (DEPC) {
                    const bool first = hk[q] && ((w1 >> 24) & 1) &&
                                       (inc_index(ent[q]) & ((JOINTS && inc_kind(ent[q]) == FK_JOINT) ? 1 : 3)) == 0;     /* one arrival per manifold / joint with a moving partner */
                    const unsigned nOwn = (unsigned)__popcll(__ballot(first && ((w1 >> 25) & 1)) & gMask), nOth = (unsigned)__popcll(__ballot(first && !((w1 >> 25) & 1)) & gMask);
                    if (k0 + q < rounds && gl == k0 + q) { depPerIt = nOwn + nOth; depNeed = nOwn; }
                }
                if (k0 + q < rounds) {
                    if constexpr (EXACT) hdr[(k0 + q) * 64 + lane] = make_int4(ent[q], w1, rc, (Z.nPeers > 1 && hk[q] && oth[q] >= 0) ? max(Z.owner[oth[q]], 0) : 0);
                    else hdr[(k0 + q) * 64 + lane] = make_int2(ent[q], w1);
                }
            }
        }
        wave_sync();
        GS_STAMP(3);
        /* (polling the watched words one pass ahead — issued before the pass's task, looked at when the next pass starts — was
         * measured and not kept: 64 000 cubes 6.9 -> 7.6 ms per step; a body that has just become ready then waits a whole pass
         * more, and the step is a chain of 372 such hops)
         *
         * A TASK IN HAND.  The step is that chain of hops, and a hop used to cost a whole pass: watched word (one round trip), the
         * records and the parked state (a second), the first half of the task, then the second.  Now a sub-group that has nothing
         * ready takes its OLDEST task in hand all the same — parked state loaded, first half done — and the next passes poll just the
         * records it still misses, like the one-body form: when they arrive only the second half of the task is left before the pose
         * goes out.  Nothing of a task in hand has been stored, so it is simply dropped when another body of the sub-group turns out
         * to be ready first (the records alone carry the dependencies: any order that respects them gives the same bits). */
        /* With many bodies per sub-group some body is ready soon and a task in hand is dropped before its records arrive (64 000 cubes,
         * 8 per sub-group: 6.3 -> 6.7 ms per step; 13 800 and 22 000 cubes: 2.75 -> 2.28 and 3.29 -> 3.08): classes with more than
         * X.specRounds bodies per sub-group run the loop without it, compiled separately (the poll state that lives across passes costs the
         * other form registers it has no use for). */
        const bool keepMode = X.specMode != 2 && rounds <= X.specRounds;
        auto multi_loop = [&](auto keepTag) __attribute__((always_inline)) -> bool {
        constexpr bool KEEP = decltype(keepTag)::value;
        unsigned watchVal = 0;
        bool prep = false;                                            /* uniform over the sub-group: a task is in hand (loaded, first half done) */
        int kPrep = 0;                                                /* which of the sub-group's bodies it belongs to */
        for (;;) {
            AVBD_MARK(g_examine);
            const bool activeK = itK < itEndK;
            if (__ballot(activeK) == 0) break;
            grpWaiting = (__ballot(activeK) & gMask) != 0;
            if (!KEEP) prep = false;
            /* one round trip: what the task in hand still misses, and the watched words of the other bodies */
            if (prep) poll_issue();
            if (dep) { if (activeK && depNeed != 0u) watchVal = xldw(rC, ctrK * 4); }
            else {
            watch_pose(activeK && blkOff >= 0, blkOff, blkPeer, watchVal);
            if (SHARD && activeK && blkOff == -2) watchVal = xldw(rGW, 0);   /* waiting for an all-gather */
            }
            if (prep) poll_check();
            bool candK = activeK;
            if (dep) { if (activeK && depNeed != 0u) candK = watchVal >= depNeed; }
            else if (activeK && blkOff != -1) candK = stamp_reached(watchVal, blkStamp);
            if (prep && gl == kPrep) candK = false;
            dbgBody = bodyK; dbgIt = itK | (itEndK << 8); dbgOff = blkOff; dbgStamp = (int)(blkStamp - X.base); dbgSeen = (int)(watchVal - X.base);
            GS_STAMP(12);                                              /* the look at the watched words (and whatever was still in flight) */
            bool ready = prep && ((__ballot(okP & okV & okD & okR) & gMask) == gMask);
            /* another body can run first: the task in hand is given up (keeping it while a YOUNGER task of the sub-group was ready, for up
             * to 5 / 20 / 100 us, was measured: 22 000 cubes 3.19 -> 3.25 / 3.35 / 5.2 ms — what a sub-group can run, it should run) */
            if (prep && !ready && (__ballot(candK) & gMask) != 0) prep = false;
            if (prep) candK = false;
            /* (only a wave with nothing else to do in this pass takes a task in hand early: the load and the first half of a task that
             * is not ready would hold up the ready tasks of the wave's other sub-groups — 64 000 cubes 6.3 -> 6.7 ms when every
             * sub-group did it in every pass) */
            const bool waveIdle = KEEP && (X.specMode == 1 || __ballot(ready || candK) == 0);
            bool inHand = false, triedK = false, specTried = false;
            int kHand = 0;
            for (;;) {
                int key = (candK && !triedK) ? ((itK << 6) | gl) : 0x7fffffff;
                key = group_min_i32(key, gw);
                if (key == 0x7fffffff && !prep && !inHand && !specTried && waveIdle) {
                    /* nothing is known to be ready: the oldest task of the sub-group (not the task behind the last iteration, which prepares nothing) */
                    specTried = true;
                    key = (activeK && itK < itStop) ? ((itK << 6) | gl) : 0x7fffffff;
                    /* arrival counters: the body that misses the fewest arrivals is the likeliest to run next (the oldest one was the next
                     * to run in one case out of five at eight bodies per sub-group) */
                    if (dep && activeK && itK < itStop) key = (int)(((depNeed - min(watchVal, depNeed)) << 12) | ((unsigned)itK << 6) | (unsigned)gl);
                    key = group_min_i32(key, gw);
                }
                const bool examine = !prep && !ready && key != 0x7fffffff;   /* uniform over the sub-group */
                if (__ballot(examine) == 0) break;
                if (examine) {
                    const int kk = key & 63, k = gBase + kk, t = __shfl(itK, k, 64);
                    load_header(__shfl(bodyK, k, 64), __shfl(eBegK, k, 64), __shfl(nEntK, k, 64), kk);
                    /* the records this task reads and the state the body parked after its last task travel together: one memory
                     * round trip tells whether the task can run and, if it can, has brought what it needs */
                    it = t;
                    set_needs();
                    poll_issue();
                    load_rest();
                    if (SHARD) { needR = has && remote && it > 0; sR = X.base + (unsigned)it; okR = !needR; }   /* the all-gather behind the previous iteration */
                    poll_check();
                    AVBD_MARK(g_examine2);
                    inHand = true; kHand = kk;
                    const unsigned long long noPose = __ballot(!okP) & gMask, noRest = __ballot(!(okV & okD)) & gMask;
                    const unsigned long long noGather = SHARD ? (__ballot(!okR) & gMask) : 0ull;
                    if ((noPose | noRest | noGather) == 0) {
                        ready = true;
                    } else if (dep) {
                        if (lane == k) { triedK = true; candK = false; }     /* in hand; the next passes poll the records it misses */
                    } else if (noPose) {
                        /* watch the pose of the first partner that is late (its last record's stamp word) */
                        const int f = __ffsll((long long)noPose) - 1;
                        const int bo = __shfl(offP + 44, f, 64);
                        const unsigned bs = (unsigned)__shfl((int)sP, f, 64);
                        const int bp = __shfl(peer, f, 64);
                        if (lane == k) { blkOff = bo; blkStamp = bs; blkPeer = bp; candK = false; }
                    } else if (noGather) {
                        if (lane == k) { blkOff = -2; blkStamp = X.base + (unsigned)t; candK = false; }
                    } else if (lane == k) triedK = true;                     /* visit / dual records follow their producer's pose within its task: next pass */
                }
            }
            GS_STAMP(0);
            /* the last task examined stays in hand, ready or not: its first half */
            if (inHand && !(ready || KEEP)) inHand = false;
            if (inHand) { prep = true; kPrep = kHand; }
            if (inHand && it < itStop) prepare_task();
            GS_STAMP(2);
#ifdef AVBD_DF_STATS
            stat[14] += 1;
            if (__ballot(ready)) { stat[5] += 1; stat[4] += (unsigned long long)__popcll(__ballot(ready && gl == 0)); }
#endif
            const bool runF = EXACT && ready && it >= itStop;           /* the owner's last dual into this rank's copy of the contact */
            const bool run = ready && !runF;
            if (runF) final_store();
            if (run) {
                finish_visit();
                finish_accumulate();
                finish_solve();
                GS_STAMP_ND(13);                                       /* dual: arithmetic and issue of its stores */
                GS_STAMP(3);                                           /* ... until they are acknowledged */
                if (it < itStop - 1 || (EXACT && (__ballot(finalD) & gMask) != 0)) park_body();   /* (EXACT: the task behind the last iteration reads the parked normal) */
                GS_STAMP_ND(15);                                       /* park: issue */
            }
            if (ready) {
                if (gl == kPrep) { ++itK; ++blkStamp; depNeed += depPerIt; if (EXACT && itK >= itStop) blkOff = -1; }   /* the same partner is the likeliest to be late again (the task behind the last iteration waits for no pose) */
                prep = false;
                spins = 0;
            }
            const bool sel = ready;
            AVBD_MARK(g_pass_end);
            if (SHARD && __ballot(sel)) {
                /* iteration k of this wave is complete when every body it serves is past it (see the one-body form above) */
                int low = bodyK >= 0 ? itK : itStop;
                for (int o = 1; o < 64; o <<= 1) low = min(low, __shfl_xor(low, o, 64));
                low = __builtin_amdgcn_readfirstlane(low);
                if (reportedM < low) {
                    const unsigned mine = (unsigned)__popcll(__ballot(bodyK >= 0));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0)
                        for (int k = reportedM; k < low; ++k) {
                            const unsigned before = __hip_atomic_fetch_add(Z.done + k, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (before + mine == (unsigned)Z.nOwned) {
                                __hip_atomic_fetch_max(Z.doneSignal, X.base + (unsigned)k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                if (Z.span && k == itStop - 1) Z.span[1] = __builtin_amdgcn_s_memrealtime();   /* profiling: the rank's last iteration is complete */
                            }
                        }
                    reportedM = low;
                }
            }
            GS_STAMP(1);
            if (__ballot(sel || inHand) == 0 && !back_off()) return false;
        }
        return true;
        };
        /* the loop without a task in hand (a sub-group examines a body when its watched word has arrived, and runs it at once) */
        auto multi_loop_plain = [&]() __attribute__((always_inline)) -> bool {
        unsigned watchVal = 0;
        for (;;) {
            AVBD_MARK(g_examine);
            const bool activeK = itK < itEndK;
            if (__ballot(activeK) == 0) break;
            grpWaiting = (__ballot(activeK) & gMask) != 0;
            bool candK = activeK;
            if (dep) {
                if (activeK && depNeed != 0u) { watchVal = xldw(rC, ctrK * 4); candK = watchVal >= depNeed; }
            } else {
            watch_pose(activeK && blkOff >= 0, blkOff, blkPeer, watchVal);
            if (SHARD && activeK && blkOff == -2) watchVal = xldw(rGW, 0);   /* waiting for an all-gather */
            if (activeK && blkOff != -1) candK = stamp_reached(watchVal, blkStamp);
            }
            dbgBody = bodyK; dbgIt = itK | (itEndK << 8); dbgOff = dep ? -3 : blkOff; dbgStamp = dep ? (int)depNeed : (int)(blkStamp - X.base); dbgSeen = dep ? (int)watchVal : (int)(watchVal - X.base);
            GS_STAMP(12);                                              /* the look at the watched words (and whatever was still in flight) */
            bool sel = false, triedK = false;
            for (;;) {
                int key = (candK && !triedK) ? ((itK << 6) | gl) : 0x7fffffff;
                key = group_min_i32(key, gw);
                const bool examine = !sel && key != 0x7fffffff;        /* uniform over the sub-group */
                if (__ballot(examine) == 0) break;
                if (examine) {
                    const int kk = key & 63, k = gBase + kk, t = key >> 6;
                    load_header(__shfl(bodyK, k, 64), __shfl(eBegK, k, 64), __shfl(nEntK, k, 64), kk);
                    /* the records this task reads and the state the body parked after its last task travel together: one memory
                     * round trip tells whether the task can run and, if it can, has brought what it needs */
                    it = t;
                    set_needs();
                    poll_issue();
                    load_rest();
                    if (SHARD) { needR = has && remote && it > 0; sR = X.base + (unsigned)it; okR = !needR; }   /* the all-gather behind the previous iteration */
                    poll_check();
                    AVBD_MARK(g_examine2);
                    const unsigned long long noPose = __ballot(!okP) & gMask, noRest = __ballot(!(okV & okD)) & gMask;
                    const unsigned long long noGather = SHARD ? (__ballot(!okR) & gMask) : 0ull;
                    if ((noPose | noRest | noGather) == 0) {
                        sel = true;
#ifdef AVBD_DF_STATS
                        {   /* was the task that runs the OLDEST task of its sub-group (what a speculative load would have fetched)? */
                            int keyO = activeK ? ((itK << 6) | gl) : 0x7fffffff;
                            keyO = group_min_i32(keyO, gw);
                            if (lane == k) { dbgTasks += 1; dbgOldest += (keyO == key) ? 1 : 0; }
                        }
#endif
                        if (lane == k) { ++itK; ++blkStamp; depNeed += depPerIt; if (EXACT && itK >= itStop) blkOff = -1; }   /* the same partner is the likeliest to be late again (the task behind the last iteration waits for no pose) */
                    } else if (dep) {
                        if (lane == k) triedK = true;                       /* counted, but a record's store is still on its way: next pass */
                    } else if (noPose) {
                        /* watch the pose of the first partner that is late (its last record's stamp word) */
                        const int f = __ffsll((long long)noPose) - 1;
                        const int bo = __shfl(offP + 44, f, 64);
                        const unsigned bs = (unsigned)__shfl((int)sP, f, 64);
                        const int bp = __shfl(peer, f, 64);
                        if (lane == k) { blkOff = bo; blkStamp = bs; blkPeer = bp; candK = false; }
                    } else if (noGather) {
                        if (lane == k) { blkOff = -2; blkStamp = X.base + (unsigned)t; candK = false; }
                    } else if (lane == k) triedK = true;                     /* visit / dual records follow their producer's pose within its task: next pass */
                }
            }
            GS_STAMP(0);
#ifdef AVBD_DF_STATS
            stat[14] += 1;
            if (__ballot(sel)) { stat[5] += 1; stat[4] += (unsigned long long)__popcll(__ballot(sel && gl == 0)); }
#endif
            const bool selF = EXACT && sel && it >= itStop;             /* the owner's last dual into this rank's copy of the contact */
            if (selF) { final_store(); spins = 0; }
            if (EXACT) sel = sel && it < itStop;
#ifdef AVBD_DF_STATS
            unsigned* trecM = reinterpret_cast<unsigned*>(X.privN) + ((size_t)(it < itStop ? it : 0) * B.n + (size_t)(body < 0 ? 0 : body)) * 4;   /* tools/debug_groups_critical.py */
            if (sel && gl == 0 && X.privN) trecM[0] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
            if (sel) prepare_task();
#ifdef AVBD_DF_STATS_PREP
            if (sel && gl == 0 && X.privN) trecM[3] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
            GS_STAMP(2);
            if (sel) {
                finish_visit();
                finish_accumulate();
                finish_solve();
                GS_STAMP_ND(13);                                       /* dual: arithmetic and issue of its stores */
                GS_STAMP(3);                                           /* ... until they are acknowledged */
                if (it < itStop - 1 || (EXACT && (__ballot(finalD) & gMask) != 0)) park_body();   /* (EXACT: the task behind the last iteration reads the parked normal) */
                GS_STAMP_ND(15);                                       /* park: issue */
                spins = 0;
#ifdef AVBD_DF_STATS
#ifdef AVBD_DF_STATS_PREP
                if (gl == 0 && X.privN) trecM[2] = (unsigned)__builtin_amdgcn_s_memrealtime();
#else
                if (gl == 0 && X.privN) { trecM[2] = (unsigned)__builtin_amdgcn_s_memrealtime(); trecM[3] = (unsigned)w; }
#endif
#endif
            }
            if (EXACT) sel = sel || selF;
            AVBD_MARK(g_pass_end);
            if (SHARD && __ballot(sel)) {
                /* iteration k of this wave is complete when every body it serves is past it (see the one-body form above) */
                int low = bodyK >= 0 ? itK : itStop;
                for (int o = 1; o < 64; o <<= 1) low = min(low, __shfl_xor(low, o, 64));
                low = __builtin_amdgcn_readfirstlane(low);
                if (reportedM < low) {
                    const unsigned mine = (unsigned)__popcll(__ballot(bodyK >= 0));
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0)
                        for (int k = reportedM; k < low; ++k) {
                            const unsigned before = __hip_atomic_fetch_add(Z.done + k, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (before + mine == (unsigned)Z.nOwned) {
                                __hip_atomic_fetch_max(Z.doneSignal, X.base + (unsigned)k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                if (Z.span && k == itStop - 1) Z.span[1] = __builtin_amdgcn_s_memrealtime();   /* profiling: the rank's last iteration is complete */
                            }
                        }
                    reportedM = low;
                }
            }
            GS_STAMP(1);
            if (__ballot(sel) == 0 && !back_off()) return false;
        }
        return true;
        };
        depSig = DEPC && X.depCnt != nullptr;                        /* every task signals, whichever loop its class runs: the partners may be counting */
        /* ---- POOL LOOP (round 4).  The launch is a chain of ~350 hops, and a third of a hop was a ready task waiting for ITS wave to come out
         * of the pass it was in for other sub-groups.  The sub-groups of a workgroup (16, 8 or 4 by class) therefore SHARE their bodies:
         * a slot table in LDS holds, per body, its next iteration and the arrivals that task waits for; every wave that starts a pass polls
         * the arrival counters of ALL the pool's slots (they are neighbours in memory: a few cache lines), takes up to four ready slots —
         * oldest iteration first, claimed with an LDS compare-and-swap — and its four sub-groups run them.  A task no longer waits for one
         * particular wave but for the first of four, and a pass fills up: the pool offers its ready tasks to whoever comes.
         * Nothing else changes: the records carry the dependencies, any order that respects them gives the same bits.
         * A slot is given back (next iteration, not claimed) one pass later, behind the poll's wait: the loads of that poll return in issue
         * order behind the stores that parked the body, so whoever claims the slot next finds the parked state in memory. */
        const bool poolWave = DEPC && pooled;
        auto pool_loop = [&]() __attribute__((always_inline)) -> bool {
            auto tail = [&](int k) { return reinterpret_cast<int*>(reinterpret_cast<char*>(sGroups) + kGrpLdsBytes + k * kGrpHdrBytesPerWave + kGrpPoolMaxRounds * 64 * 8); };
            int* const sBody = tail(0); int* const sE0 = tail(1); int* const sNeed = tail(2); int* const sState = tail(3);
            const HdrT* const hdr0 = reinterpret_cast<const HdrT*>(reinterpret_cast<char*>(sGroups) + kGrpLdsBytes);
            const int G = 4 >> cls, sg = gBase >> (4 + cls);            /* sub-groups per wave, this lane's */
            const int S = 4 * G * rounds, qBase = cls * cap + (wl >> 2) * S;   /* slots of this pool; where slot 0 counts its arrivals (its class-major rank) */
            /* slot (wave-in-workgroup, sub-group, round) <-> this lane's body: exactly the header table's layout */
            if (gl < rounds) {
                const int sl = (wv * G + sg) * rounds + gl;
                sBody[sl] = bodyK; sE0[sl] = eBegK; sNeed[sl] = (int)depNeed;
                sState[sl] = (bodyK >= 0 ? 0 : itStop) | (nEntK << 16) | ((int)depPerIt << 24);
            }
            __syncthreads();                                           /* (all four waves of a pool workgroup are here: they serve one class) */
            constexpr int kInf = 0x7fffffff;
            int relSlot = -1, relIt = 0, relNeed = 0;                 /* uniform over a sub-group: the slot it gives back behind the next poll */
            for (;;) {
                AVBD_MARK(g_examine);
                /* ---- poll: every lane looks at up to three slots ---- */
                int key[3]; unsigned cnt[3]; int st[3], nd[3];              /* 16 * kGrpPoolMaxRounds = 192 slots at most */
                bool anyActive = false;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int sl = lane + 64 * i;
                    st[i] = itStop; nd[i] = 0; cnt[i] = 0u;
                    if (sl < S) { st[i] = sState[sl]; nd[i] = sNeed[sl]; }
                    const bool act = (st[i] & 255) < itStop;
                    anyActive |= act;
                    if (act && !(st[i] & 256) && nd[i] != 0) cnt[i] = xldw(rC, (qBase + sl) * 4);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const bool act = (st[i] & 255) < itStop && !(st[i] & 256);
                    key[i] = (act && (nd[i] == 0 || cnt[i] >= (unsigned)nd[i])) ? (((st[i] & 255) << 8) | (lane + 64 * i)) : kInf;
                }
                if (__ballot(anyActive) == 0) break;
                grpWaiting = true;
                dbgBody = lane < S ? sBody[lane] : -1; dbgIt = (st[0] & 255) | (itStop << 8); dbgOff = -4; dbgStamp = nd[0]; dbgSeen = (int)cnt[0];
                GS_STAMP(12);
                /* ---- the slots of the last pass go back to the pool (its stores are in memory: the poll's loads returned behind them) ---- */
                if (__ballot(relSlot >= 0) != 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* (what the poll's wait amounts to: memory operations complete in issue order) */
                    if (relSlot >= 0 && gl == 0) { sNeed[relSlot] = relNeed; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); sState[relSlot] = relIt; }
                }
                relSlot = -1;
                /* ---- take up to four ready slots, oldest iteration first ---- */
                int mine = kInf;                                      /* the key this lane's sub-group runs */
                int taken = 0;
                for (int tries = 0; tries < 8 && taken < G; ++tries) {
                    int best = key[0] < key[1] ? key[0] : key[1]; best = key[2] < best ? key[2] : best;
                    const int m = wave_min_i32(best);
                    if (m == kInf) break;
                    bool got = false;
                    if (best == m) {                                   /* exactly one lane holds a given slot */
                        const int sl = m & 255;
                        int was = 0;
#pragma unroll
                        for (int i = 0; i < 3; ++i) if (key[i] == m) { was = st[i]; key[i] = kInf; }
                        got = atomicCAS(&sState[sl], was, was | 256) == was;
                    }
                    if (__ballot(got) != 0) { if (sg == taken) mine = m; ++taken; }
                }
                bool sel = mine != kInf;
                int slot = 0;
                if (sel) {
                    slot = mine & 255; it = mine >> 8;
                    hdrRow = hdr0 + (slot / (G * rounds)) * (kGrpHdrBytesPerWave / (int)sizeof(HdrT)) + (slot % rounds) * 64 + ((slot / rounds) % G) * gw;
                    load_header(sBody[slot], sE0[slot], (sState[slot] >> 16) & 255, 0);
                    set_needs();
                    poll_issue();
                    load_rest();
                    poll_check();
                    AVBD_MARK(g_examine2);
                    if ((__ballot(okP & okV & okD) & gMask) != gMask) {  /* counted, but a record's store is still on its way: back into the pool as it was */
                        if (gl == 0) sState[slot] = sState[slot] & ~256;
                        sel = false;
                    }
                }
                GS_STAMP(0);
#ifdef AVBD_DF_STATS
                stat[14] += 1;
                if (__ballot(sel)) { stat[5] += 1; stat[4] += (unsigned long long)__popcll(__ballot(sel && gl == 0)); }
                unsigned* trecM = reinterpret_cast<unsigned*>(X.privN) + ((size_t)(it < itStop ? it : 0) * B.n + (size_t)(body < 0 ? 0 : body)) * 4;
                if (sel && gl == 0 && X.privN) trecM[0] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
                if (sel) prepare_task();
                GS_STAMP(2);
                if (sel) {
                    finish_visit();
                    finish_accumulate();
                    finish_solve();
                    GS_STAMP_ND(13);
                    GS_STAMP(3);
                    if (it < itStop - 1) park_body();
                    GS_STAMP_ND(15);
                    spins = 0;
                    { const int wd = sState[slot]; relSlot = slot; relIt = (wd & ~511) | (it + 1); relNeed = sNeed[slot] + ((wd >> 24) & 255); }
#ifdef AVBD_DF_STATS
                    if (gl == 0 && X.privN) { trecM[2] = (unsigned)__builtin_amdgcn_s_memrealtime(); trecM[3] = (unsigned)w; }
#endif
                }
                AVBD_MARK(g_pass_end);
                GS_STAMP(1);
                if (__ballot(sel) == 0 && !back_off()) return false;
            }
            return true;
        };
        if (poolWave) { if (!pool_loop()) return; }
        else
        if (keepMode) { if (!multi_loop(std::true_type{})) return; }
        else if (!multi_loop_plain()) return;
    }
#ifdef AVBD_DF_STATS
    if (lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(errWord + 16) + (size_t)w * 16;
        stat[6] = __builtin_amdgcn_s_memrealtime() - t0;
        for (int i = 0; i < 16; ++i) o[i] = stat[i];
        if (MULTI) { unsigned a = dbgTasks, b = dbgOldest; for (int q = 1; q < 64; ++q) { a += (unsigned)__builtin_amdgcn_readlane((int)dbgTasks, q); b += (unsigned)__builtin_amdgcn_readlane((int)dbgOldest, q); } o[10] = a; o[11] = b; }
    }
#endif
}

template <bool MULTI, int SHARDK, bool JOINTS = false>
__global__ __launch_bounds__(64 * kGrpWavesPerBlock) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_iterate_groups(BodyTable B, ManifoldTable M, const int* __restrict__ incStart, const int* __restrict__ inc, SolverParams P, int* errWord,
                      Counters* cntW, ResidentExchange X, const GroupPlan* __restrict__ plan, const int* __restrict__ byClass, int cap, int sleepSel, ShardExchange Z)
{
    iterate_groups<MULTI, SHARDK, JOINTS>(B, M, incStart, inc, P, errWord, cntW, X, plan, byClass, cap, sleepSel, Z, (int)blockIdx.x);
    /* exact cuts between processes, profiling: the end of the span is the last wave out (the all-gather form stamps "last iteration complete" itself) */
    if (SHARDK == 2 && Z.span && (threadIdx.x & 63) == 0) atomicMax(Z.span + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

/* Exact cuts with several ranks on ONE device (the emulation of avbd_group_step): all ranks in one launch, `wgPerRank` workgroups
 * each.  The ranks wait for each other's records, so their workgroups must all be resident at the same time; separate launches
 * on separate streams are only that while every stream has a hardware queue of its own — a process with more streams than
 * queues runs two of them one after the other, and the first then waits for the second forever.  (Between GPUs every rank is a
 * launch of its own device.) */
struct ExactRankArgs {
    BodyTable B; ManifoldTable M; const int* incStart; const int* inc; SolverParams P; int* errWord; Counters* cntW; ResidentExchange X;
    const GroupPlan* plan; const int* byClass; int cap; int sleepSel; ShardExchange Z; int multi;
};
template <bool JOINTS>
__global__ __launch_bounds__(64 * kGrpWavesPerBlock) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_iterate_groups_exact(const ExactRankArgs* __restrict__ args, int wgPerRank)
{
    const int rank = (int)blockIdx.x / wgPerRank, local = (int)blockIdx.x % wgPerRank;
    const ExactRankArgs& A = args[rank];
    if (A.multi) iterate_groups<true, 2, JOINTS>(A.B, A.M, A.incStart, A.inc, A.P, A.errWord, A.cntW, A.X, A.plan, A.byClass, A.cap, A.sleepSel, A.Z, local);
    else iterate_groups<false, 2, JOINTS>(A.B, A.M, A.incStart, A.inc, A.P, A.errWord, A.cntW, A.X, A.plan, A.byClass, A.cap, A.sleepSel, A.Z, local);
}

} // namespace avbd
