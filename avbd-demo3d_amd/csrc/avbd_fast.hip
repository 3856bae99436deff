/*
 * avbd_fast.hip — the throughput sweep of the AVBD iteration loop (solver.cpp:340-431) for gfx950: "fast mode".
 *
 * See avbd_fast_api.h for what differs from the exact schedules (order of bodies, order of sums, who runs a dual) and
 * DESIGN.md section 2b for the stated tolerance.  Two kernels per step:
 *
 *   k_fast_plan   colours the swept bodies (Jones-Plassmann rounds on the CSR adjacency the topology kernels built; the
 *                 result is the sequential greedy colouring in priority order, whatever the timing), classes them by
 *                 incidence count (a body gets 1 / 4 / 8 / 16 / 32 / 64 lanes), lays the (colour, class) groups out as TILES
 *                 of 64 lanes and writes, per lane, the constants of its incidence entry (SoA planes indexed by
 *                 tile * 64 + lane: a wave reads them with coalesced 16-byte loads) — anchors, the contact basis, bias
 *                 terms, friction, force cap, partner, contact slot — and per body the constants of its visit.
 *   k_fast_sweep  one persistent launch for the whole loop.  A wave owns the tiles t = w, w + W, ...; a tile visit is
 *                 the primal solve of its 64 / G bodies: every lane evaluates its contact's three rows (computeConstraint,
 *                 computeDerivatives, row force, gyroscopic term: the functions of avbd_rows.h), the 27 entries of the
 *                 body's system are summed over the sub-group's lanes with DPP butterflies, the Schur solve and the pose
 *                 update follow.  Dependencies between colours: stamped 16-byte sc1 records (pose records carry the body's
 *                 epoch, contact records the visit count) that the consuming lane polls — or, stamps = 0, a grid barrier
 *                 per colour (the cross-check form: both give the same bits).
 *
 * HBM layout and algorithmic bytes: DESIGN.md section 4b.  No MFMA: 6x6 blocks per body, three rows per contact.
 */
#include "avbd_fast_api.h"
#include "avbd_rows.h"
#include "avbd_util.h"
#include <algorithm>

#pragma clang fp contract(off)

namespace avbd {

typedef unsigned fv4u __attribute__((ext_vector_type(4)));

constexpr int kPlanThreads = 1024;
constexpr int kSweepThreads = 512, kSweepThreadsLat = 256, kSweepThreadsOne = 768;   /* two-pass form: two waves per SIMD; latency form: one; one-pass form: three (145 registers; four waves at 128 registers spill 64 B: 64 000 cubes 893 -> 1 068 us) */
constexpr unsigned long long kFastTimeoutTicks = 400000000ull;   /* 4 s of s_memrealtime (100 MHz) */

enum : int { FL_VALID = 1, FL_ISA = 2, FL_FIRST = 4, FL_SOLE = 8,
             FL_JOINT = 16 /* the lane holds one half (three rows) of a Joint */, FL_PART1 = 32 /* ... its angular half */, FL_WORLD = 64 /* ... whose bodyA is the world */, FL_FIXANG = 128 /* avbd_set_joint_angular_sign */,
             FL_SPRING = 256 /* the lane holds a Spring (one row; FL_JOINT is set too: the lane takes the generic-force path) */ };

__host__ __device__ constexpr int fast_class_lanes(int cls) { return cls == 0 ? 1 : (2 << cls); }
__device__ __forceinline__ int fast_class_of(int deg) { return deg == 0 ? 0 : (deg <= 4 ? 1 : (deg <= 8 ? 2 : (deg <= 16 ? 3 : (deg <= 32 ? 4 : 5)))); }

/* ---- memory access flavours --------------------------------------------------------------------------------------
 * Bytes another workgroup of the SAME launch reads or writes move with sc1 accesses only (the per-XCD L2s are not
 * coherent inside a launch, DESIGN.md "Three memory rules"): one lane, one 16-byte request. */
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000); }
__device__ __forceinline__ fv4u fld(__amdgpu_buffer_rsrc_t r, int byteOff) { return __builtin_amdgcn_raw_buffer_load_b128(r, byteOff, 0, 16); }
__device__ __forceinline__ void fst(__amdgpu_buffer_rsrc_t r, int byteOff, fv4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, byteOff, 0, 16); }
/* ... or, when every wave of the launch sits on ONE XCD (k_fast_sweep's one-XCD form): a plain store — the line stays in that XCD's L2, where the sc1 loads of the
 * other CUs find it (a hand-off of 0.3 us instead of 0.5 - 0.7, tools/ubench/xcd_pingpong.hip) */
__device__ __forceinline__ void fst_local(__amdgpu_buffer_rsrc_t r, int byteOff, fv4u v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, byteOff, 0, 0); }
__device__ __forceinline__ unsigned fast_xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }
__device__ __forceinline__ unsigned fu(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float uf(unsigned u) { return __uint_as_float(u); }

/* ---- grid barrier ------------------------------------------------------------------------------------------------
 * Every workgroup stores the target into its own word; wave 0 of every workgroup polls the words of all workgroups
 * (up to 64: one load per lane) or, beyond 64 workgroups, the leader of each group of 64 polls its group and raises the
 * group's word, and everybody polls the group words: one or two hops of ~0.7 us, no atomics on a shared word (13 ns
 * apiece, serialised).  FENCE: plain loads / stores around it are made coherent with agent-scope fences (k_fast_plan);
 * without it the caller moves shared bytes with sc1 accesses and only drains its stores (k_fast_sweep). */
template <bool FENCE>
__device__ __forceinline__ void fast_grid_barrier(unsigned* sync, int nWG, unsigned target, int* bad)
{
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nWG > 1 && threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) __hip_atomic_store(sync + blockIdx.x, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nWG <= 64) {
            for (;;) {
                const unsigned v = lane < nWG ? __hip_atomic_load(sync + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                if (__all((int)(v - target) >= 0)) break;
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > kFastTimeoutTicks) { if (lane == 0) atomicOr(bad, 256); break; }
            }
        } else {
            const int group = blockIdx.x >> 6, nGroups = (nWG + 63) >> 6;
            if ((blockIdx.x & 63) == 0) {
                const int first = group << 6;
                for (;;) {
                    const unsigned v = first + lane < nWG ? __hip_atomic_load(sync + first + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                    if (__all((int)(v - target) >= 0)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > kFastTimeoutTicks) { if (lane == 0) atomicOr(bad, 256); break; }
                }
                if (lane == 0) __hip_atomic_store(sync + 1024 + group, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (;;) {
                const unsigned v = lane < nGroups ? __hip_atomic_load(sync + 1024 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                if (__all((int)(v - target) >= 0)) break;
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > kFastTimeoutTicks) { if (lane == 0) atomicOr(bad, 256); break; }
            }
        }
    }
    __syncthreads();
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

/* ---- pose records --------------------------------------------------------------------------------------------------
 * Two 16-byte records per swept body, each self-validating: (p.x p.y p.z | e16 : qw[31:16]) and (q.x q.y q.z | e16 : qw[15:0]),
 * e16 = the body's epoch (primal updates applied this step, 0 .. 61) — seven floats and the stamp in 32 bytes. */
__device__ __forceinline__ void pose_encode(f3 p, q4 q, unsigned epoch, fv4u& r0, fv4u& r1)
{
    const unsigned w = fu(q.w);
    r0 = fv4u{fu(p.x), fu(p.y), fu(p.z), (epoch << 16) | (w >> 16)};
    r1 = fv4u{fu(q.x), fu(q.y), fu(q.z), (epoch << 16) | (w & 0xFFFFu)};
}
__device__ __forceinline__ bool pose_decode(fv4u r0, fv4u r1, unsigned epoch, f3& p, q4& q)
{
    p = mk3(uf(r0.x), uf(r0.y), uf(r0.z));
    q = mkq(uf(r1.x), uf(r1.y), uf(r1.z), uf((r0.w << 16) | (r1.w & 0xFFFFu)));
    return (r0.w >> 16) == epoch && (r1.w >> 16) == epoch;
}

/* =====================================================================================================================
 * k_fast_plan
 * ===================================================================================================================== */
__device__ __forceinline__ unsigned fast_hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

constexpr int kFastRecolorAbove = 10;         /* last step's colouring is dropped (every body coloured afresh) once it has grown beyond this many colours */
constexpr int kPlanNbrRegs = 12;             /* neighbours of a body its thread keeps in registers through the rounds of the colouring */

__device__ __forceinline__ unsigned fast_prio(const int* __restrict__ adjStart, int b, unsigned seed)
{
    const int degAdj = adjStart[b + 1] - adjStart[b];
    return ((unsigned)min(degAdj, 127) << 24) | (fast_hash((unsigned)b + seed * 0x9e3779b9u) >> 8);
}

/* k_fast_plan: colours, (colour, class) groups, tile table, body constants — one thread per body, persistent, grid barriers between the stages.
 * SINGLE: one workgroup (launched alone, up to 1024 bodies): colours in LDS, every barrier a workgroup barrier.  Otherwise the colours move with
 * sc1 accesses and the group counters with atomics; everything else a stage writes is read by LATER launches only (k_fast_fill, k_fast_sweep), so no
 * fence is needed anywhere. */
template <bool SINGLE>
__global__ __launch_bounds__(kPlanThreads) void k_fast_plan(FastLaunch L)
{
    const BodyTable& B = L.B;
    const FastBuffers& F = L.F;
    FastPlan* plan = F.plan;
    const int n = B.n;
    const int nWG = gridDim.x;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;              /* this thread's body */
    unsigned target = L.syncBase;
    __shared__ int sBodyBase[kFastMaxColors * kFastClasses], sTileBase[kFastMaxColors * kFastClasses], sCount[kFastMaxColors * kFastClasses];
    __shared__ int sNative[kFastMaxColors * kFastClasses], sIn[kFastMaxColors * kFastClasses], sOut[kFastMaxColors * kFastClasses];   /* (the leftovers of a class that move up one: stage 3) */
    __shared__ int sColorStart[kFastMaxColors + 1], sTotals[3];
    __shared__ int sColor[SINGLE ? kPlanSingleBodies : 1];

    auto barrier = [&]() {
        if (SINGLE) __syncthreads();
        else fast_grid_barrier<false>(F.sync, nWG, ++target, &plan->bad);
    };
    auto get_color = [&](int i) -> int { return SINGLE ? sColor[i] : __hip_atomic_load(F.color + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto set_color = [&](int i, int c) { if (SINGLE) sColor[i] = c; else __hip_atomic_store(F.color + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    /* ---- stage 0: reset, the body's neighbours, pose record, vetoes ---- */
    for (int i = threadIdx.x; i < kFastMaxColors * kFastClasses; i += blockDim.x) sCount[i] = 0;
    if (blockIdx.x == 0) {
        if (!SINGLE) { for (int i = threadIdx.x; i < kFastMaxColors * kFastClasses; i += blockDim.x) atomicExch(&plan->groupCount[i], 0); }
        if (threadIdx.x == 0) {
            atomicExch(&plan->bad, n > (int)(gridDim.x * blockDim.x) ? 16 : 0);
            atomicExch(&plan->remaining[0], 0); atomicExch(&plan->remaining[1], 0); atomicExch(&plan->remaining[2], 0);
        }
    }
    const bool inRange = b < n;
    const bool swept = inRange && B.solved[b] != 0;
    int nbr[kPlanNbrRegs];                                            /* swept neighbours among the first adjacency entries, -1: none */
    unsigned higher = 0u;                                             /* bit k: nbr[k] has the higher priority */
    int adjS = 0, adjE = 0, deg = 0, veto = 0;
    bool isolated = false;                                            /* a swept body none of whose forces leads to another swept body: nobody waits for it, it waits for nobody */
#pragma unroll
    for (int k = 0; k < kPlanNbrRegs; ++k) nbr[k] = -1;
    /* Last step's colouring is kept where it is still one: a body keeps its colour unless a neighbour of higher priority held the same colour
     * (a new contact between two bodies of one colour: the lower priority gives way) — decided on LAST step's colours alone, so the result does not
     * depend on timing; the rounds below then colour only what is left.  A settled pile keeps every colour and the colouring costs one look at the
     * neighbours.  (reuse = 0: the first step after an upload; or last step's colouring had grown beyond kFastRecolorAbove colours) */
    /* (a scheduled slot for a fresh colouring is passed over while the kept colouring needs no more colours than the best fresh one of late did: a fresh colouring is 11 us at
     * Stress1000, 40 - 60 beyond one workgroup) */
    const bool reuse = L.reuse != 0 && plan->numColors > 0 && plan->numColors <= kFastRecolorAbove &&
                       (L.freshDue == 0 || (L.freshDue == 1 && plan->lastFreshColors > 0 && plan->numColors <= plan->lastFreshColors));
    /* the hash that breaks the ties of the priorities has a seed: greedy colourings of one pile differ by a colour from seed to seed (Stress1000: 4 for three
     * seeds in four, 5 for the fourth; a colour is 20 dependent solves per step), so a fresh colouring that needed MORE colours than the best of the fresh colourings
     * before it moves on to the next seed for the colourings to come — decided on the device from the plan's own history, the same in every run */
    const unsigned seed = (unsigned)plan->seed + (unsigned)L.seedBase;
    /* (one workgroup: every body's priority once, through LDS — a thread read the degrees of its twelve neighbours from memory otherwise) */
    __shared__ unsigned sPrio[SINGLE ? kPlanSingleBodies : 1];
    if (SINGLE) { if (inRange) sPrio[b] = fast_prio(L.adjStart, b, seed); __syncthreads(); }
    auto prio_of = [&](int i) -> unsigned { return SINGLE ? sPrio[i] : fast_prio(L.adjStart, i, seed); };
    int myColor = swept ? -1 : -2;
    if (swept) {
        adjS = L.adjStart[b]; adjE = L.adjStart[b + 1];
        const unsigned pb = prio_of(b);
        const int old = reuse ? F.colorPrev[b] : -1;
        /* a body gives its colour up when a neighbour of higher priority held the same one (a new contact between two bodies of one colour), and also when a
         * LOWER colour is free among its neighbours' old colours (contacts that went away): the colouring stays "greedy-stable" — as good as a fresh greedy
         * colouring, which a colouring that is only ever repaired is not (it grew from 5 to 8 colours while the 8 000-cube pile formed) */
        bool keep = old >= 0;
        unsigned long long usedOld = 0ull;
#pragma unroll
        for (int k = 0; k < kPlanNbrRegs; ++k) {
            if (adjS + k < adjE) {
                const int p = L.adjOther[adjS + k];
                if (p >= 0 && inc_kind(L.adjRef[adjS + k]) != FK_IGNORE && B.solved[p] != 0) {
                    nbr[k] = p;
                    const unsigned pp = prio_of(p);
                    const bool hi = pp > pb || (pp == pb && p > b);
                    if (hi) higher |= 1u << k;
                    if (keep) {
                        const int po = F.colorPrev[p];
                        if (po >= 0) usedOld |= 1ull << po;
                        if (hi && po == old) keep = false;
                    }
                }
            }
        }
        if (keep && adjE - adjS > kPlanNbrRegs) {
            for (int i = adjS + kPlanNbrRegs; i < adjE && keep; ++i) {
                const int p = L.adjOther[i];
                if (p < 0 || inc_kind(L.adjRef[i]) == FK_IGNORE || B.solved[p] == 0) continue;
                const unsigned pp = prio_of(p);
                const int po = F.colorPrev[p];
                if (po >= 0) usedOld |= 1ull << po;
                if ((pp > pb || (pp == pb && p > b)) && po == old) keep = false;
            }
        }
        if (keep && (~usedOld & ((1ull << old) - 1ull)) != 0ull) keep = false;
        if (keep) myColor = old;
        {
            bool any = false;
#pragma unroll
            for (int k = 0; k < kPlanNbrRegs; ++k) any |= nbr[k] >= 0;
            for (int i = adjS + kPlanNbrRegs; i < adjE && !any; ++i) {
                const int p = L.adjOther[i];
                any = p >= 0 && inc_kind(L.adjRef[i]) != FK_IGNORE && B.solved[p] != 0;
            }
            isolated = !any;
        }
        const int e0 = L.incStart[b], e1 = L.incStart[b + 1];
        deg = e1 - e0;
        veto = deg > 64 ? 1 : 0;
        const float4 pr4 = B.plane(BP_POSR)[b], qr = B.plane(BP_QUAT)[b];
        fv4u r0, r1;
        pose_encode(mk3(pr4.x, pr4.y, pr4.z), mkq(qr.x, qr.y, qr.z, qr.w), 0u, r0, r1);
        reinterpret_cast<fv4u*>(F.xpose)[2 * b] = r0;
        reinterpret_cast<fv4u*>(F.xpose)[2 * b + 1] = r1;
    }
    if (inRange) set_color(b, myColor);
    barrier();
    if (veto) atomicOr(&plan->bad, veto);                             /* (behind the barrier: workgroup 0 reset the word in front of it) */

    /* ---- stage 1: Jones-Plassmann colouring.  A body takes the smallest colour no coloured neighbour holds once every
     * neighbour of HIGHER priority is coloured; lower-priority neighbours wait for it, so its colour is a function of
     * final colours only: the sequential greedy colouring in priority order, whatever the timing of the rounds. ---- */
    int rounds = 0;
    for (;; ++rounds) {
        if (myColor == -1) {
            unsigned long long used = 0ull;
            bool blocked = false;
            int cn[kPlanNbrRegs];
#pragma unroll
            for (int k = 0; k < kPlanNbrRegs; ++k) cn[k] = nbr[k] >= 0 ? get_color(nbr[k]) : -2;
#pragma unroll
            for (int k = 0; k < kPlanNbrRegs; ++k) {
                if (cn[k] >= 0) used |= 1ull << cn[k];
                else if (cn[k] == -1 && ((higher >> k) & 1u)) blocked = true;
            }
            if (adjE - adjS > kPlanNbrRegs) {                         /* a long adjacency list: the rest from memory */
                const unsigned pb = prio_of(b);
                for (int i = adjS + kPlanNbrRegs; i < adjE && !blocked; ++i) {
                    const int p = L.adjOther[i];
                    if (p < 0 || inc_kind(L.adjRef[i]) == FK_IGNORE) continue;
                    const int cp = get_color(p);
                    if (cp == -2) continue;
                    if (cp == -1) {
                        const unsigned pp = prio_of(p);
                        if (pp > pb || (pp == pb && p > b)) blocked = true;
                    } else used |= 1ull << cp;
                }
            }
            if (!blocked) {
                int c = used == ~0ull ? 64 : __ffsll((long long)~used) - 1;
                if (c >= kFastIsolatedRow) { atomicOr(&plan->bad, 4); c = kFastIsolatedRow - 1; }   /* (the last row of the group table is the isolated bodies') */
                myColor = c;
                set_color(b, c);
            }
        }
        const int wgLeft = __syncthreads_count(myColor == -1 ? 1 : 0) > 0 ? 1 : 0;
        if (SINGLE) {
            if (wgLeft == 0 || rounds > 4096) break;
        } else {
            if (threadIdx.x == 0) {
                if (wgLeft) atomicAdd(&plan->remaining[rounds % 3], 1);
                if (blockIdx.x == 0) atomicExch(&plan->remaining[(rounds + 1) % 3], 0);
            }
            barrier();
            const int rem = __hip_atomic_load(&plan->remaining[rounds % 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rem == 0 || rounds > 4096) break;
        }
    }
    if (SINGLE && inRange) F.color[b] = myColor;

    /* ---- stage 2: bodies per (colour, class) ---- */
    /* (counted per workgroup in LDS first, one global atomic per group and workgroup: 64 000 bodies are 64 000 atomics on a dozen words otherwise, 13 ns apiece) */
    int pos = 0, grp = 0;
    if (myColor >= 0) {
        grp = (isolated ? kFastIsolatedRow : myColor) * kFastClasses + fast_class_of(min(deg, 64));
        pos = atomicAdd(&sCount[grp], 1);
    }
    __syncthreads();
    if (!SINGLE) {
        for (int i = threadIdx.x; i < kFastMaxColors * kFastClasses; i += blockDim.x) {
            const int mine = sCount[i];
            sBodyBase[i] = mine ? atomicAdd(&plan->groupCount[i], mine) : 0;     /* this workgroup's first position in the group */
        }
        __syncthreads();
        if (myColor >= 0) pos += sBodyBase[grp];
    }
    barrier();

    /* ---- stage 3: the tile table (colour-major, class-minor), by every workgroup for itself: lane c of the first wave takes colour c's five classes,
     * a wave scan gives the offsets ---- */
    if (!SINGLE) { for (int i = threadIdx.x; i < kFastMaxColors * kFastClasses; i += blockDim.x) sCount[i] = __hip_atomic_load(&plan->groupCount[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        /* The bodies of a (colour, class) group that do not fill its last tile move up one class when the next class's last tile has the room (a body with 3 contacts
         * sits in 8 lanes as well as in 4): one tile less per group that does — Stress1000: 131 tiles instead of 137, and below the 128 waves of one XCD most steps.
         * In a group the bodies that came up stand first, its own behind them, so the ones that move on are always its own. */
        int cnt[kFastClasses], bodiesC = 0, tilesC = 0, in = 0;
#pragma unroll
        for (int k = 0; k < kFastClasses; ++k) {
            const int native = sCount[c * kFastClasses + k];
            const int per = 64 / fast_class_lanes(k);
            const int tot = native + in;
            const int r = tot % per;
            int out = 0;
            if (k >= 1 && k + 1 < kFastClasses && r > 0 && r <= native) {
                const int perUp = 64 / fast_class_lanes(k + 1), nativeUp = sCount[c * kFastClasses + k + 1];
                const int slack = (perUp - nativeUp % perUp) % perUp;
                if (r <= slack) out = r;
            }
            sNative[c * kFastClasses + k] = native; sIn[c * kFastClasses + k] = in; sOut[c * kFastClasses + k] = out;
            cnt[k] = tot - out;
            in = out;
            bodiesC += cnt[k]; tilesC += (cnt[k] + per - 1) / per;
        }
        int bodiesIncl = bodiesC, tilesIncl = tilesC;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int bo = __shfl_up(bodiesIncl, o, 64), to = __shfl_up(tilesIncl, o, 64);
            if (c >= o) { bodiesIncl += bo; tilesIncl += to; }
        }
        int bodies = bodiesIncl - bodiesC, tiles = tilesIncl - tilesC;
        sColorStart[c] = tiles;
#pragma unroll
        for (int k = 0; k < kFastClasses; ++k) {
            const int per = 64 / fast_class_lanes(k);
            sBodyBase[c * kFastClasses + k] = bodies; sTileBase[c * kFastClasses + k] = tiles;
            sCount[c * kFastClasses + k] = cnt[k];                     /* (what stage 4 and the plan hold: the groups as they are laid out) */
            bodies += cnt[k]; tiles += (cnt[k] + per - 1) / per;
        }
        const unsigned long long usedColors = __ballot(bodiesC > 0) & ~(1ull << kFastIsolatedRow);
        if (c == 63) { sColorStart[kFastMaxColors] = tilesIncl; sTotals[0] = bodiesIncl; sTotals[1] = tilesIncl; sTotals[2] = usedColors ? 64 - __builtin_clzll(usedColors) : 0; }
    }
    __syncthreads();
    const int numTiles = sTotals[1];
    const bool overflow = numTiles > F.tileCap;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < kFastMaxColors * kFastClasses; i += blockDim.x) { plan->groupBodyBase[i] = sBodyBase[i]; plan->groupTileBase[i] = sTileBase[i]; if (SINGLE) plan->groupCount[i] = sCount[i]; }
        for (int i = threadIdx.x; i <= kFastMaxColors; i += blockDim.x) plan->colorTileStart[i] = sColorStart[i];
        if (threadIdx.x == 0) {
            plan->numSwept = sTotals[0]; plan->numTiles = overflow ? 0 : numTiles; plan->numColors = sTotals[2]; plan->rounds = rounds + 1;
            if (!reuse) {
                /* (the fewest colours a fresh colouring needed lately — forgotten after 64 of them, the pile changes —: a colouring that needs more tries the next seed) */
                int best = plan->lastFreshColors, age = plan->pad[0] + 1;
                if (best <= 0 || sTotals[2] < best || age > 64) { best = sTotals[2]; age = 0; }
                else if (sTotals[2] > best) plan->seed = plan->seed + 1;
                plan->lastFreshColors = best; plan->pad[0] = age;
            }
            if (overflow) atomicOr(&plan->bad, 8);
        }
    }
    if (overflow) return;

    /* ---- stage 4: bodies into their groups, body constants, tile descriptors (read by the next launches) ---- */
    if (myColor >= 0) {
        const int stay = sNative[grp] - sOut[grp];
        const int at = pos >= stay ? sBodyBase[grp + 1] + (pos - stay) : sBodyBase[grp] + sIn[grp] + pos;
        F.cbody[at] = b;
        const float4 np = B.plane(BP_NPOS)[b], nq = B.plane(BP_NQUAT)[b], mp = B.plane(BP_MASS)[b];
        const float4 i0 = B.plane(BP_I0)[b], i1 = B.plane(BP_I1)[b], i2 = B.plane(BP_I2)[b];
        const float4 j0 = B.plane(BP_J0)[b], j1 = B.plane(BP_J1)[b], j2 = B.plane(BP_J2)[b];
        F.bodyC[(size_t)0 * F.bodyCap + at] = make_float4(np.x, np.y, np.z, mp.x);
        F.bodyC[(size_t)1 * F.bodyCap + at] = nq;
        F.bodyC[(size_t)2 * F.bodyCap + at] = make_float4(i0.x, i1.y, i2.z, mp.y);     /* box inertia is diagonal (rigid.cpp:24-40) */
        F.bodyC[(size_t)3 * F.bodyCap + at] = make_float4(j0.x, j1.y, j2.z, __int_as_float(b));
    }
    const int gtid = b, stride = gridDim.x * blockDim.x;
    for (int gi = 0; gi < (sTotals[2] + 1) * kFastClasses; ++gi) {
        const int g = gi < sTotals[2] * kFastClasses ? gi : kFastIsolatedRow * kFastClasses + (gi - sTotals[2] * kFastClasses);   /* the colours' rows, then the isolated bodies' */
        const int cnt = sCount[g];
        if (!cnt) continue;
        const int k = g % kFastClasses, per = 64 / fast_class_lanes(k), nt = (cnt + per - 1) / per;
        for (int t = gtid; t < nt; t += stride)
            F.tiles[sTileBase[g] + t] = make_int2(sBodyBase[g] + t * per, min(per, cnt - t * per) | (k << 16) | ((g / kFastClasses) << 24));
    }
}

/* k_fast_fill: the per-lane constants of every tile (one thread per lane), from the Manifold table of the step */
__global__ __launch_bounds__(256) void k_fast_fill(FastLaunch L)
{
    const BodyTable& B = L.B;
    const ManifoldTable& M = L.M;
    const FastBuffers& F = L.F;
    const FastPlan* plan = F.plan;
    if (plan->bad != 0) return;
    const int numTiles = plan->numTiles;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    const size_t SC = (size_t)F.slotCap;
    for (long long idx = gtid; idx < (long long)numTiles * 64; idx += stride) {
        const int t = (int)(idx >> 6), lane = (int)(idx & 63);
        const int2 td = F.tiles[t];
        const int cls = (td.y >> 16) & 0xFF, G = fast_class_lanes(cls), nb = td.y & 0xFFFF;
        if (cls == 0) continue;                                       /* bodies without contacts: no lane constants */
        const int j = lane / G, r = lane % G;
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
        int partner = -1, slot = 0, flags = 0;
        if (j < nb) {
            const int body = F.cbody[td.x + j];
            const int e0 = L.incStart[body], deg = L.incStart[body + 1] - e0;
            if (r < deg && r < 64) {
                const int ent = L.inc[e0 + r];
                if (inc_kind(ent) == FK_MANIFOLD) {
                    const int s = inc_index(ent), m = s >> 2;
                    const bool isA = inc_isA(ent) != 0;
                    const int2 pr = M.pair[m];
                    const int other = isA ? pr.y : pr.x;
                    const float4 ra = M.cRA[s], rb = M.cRB[s], nn = M.cN[s];
                    const float imOwn = B.plane(BP_MASS)[body].y, imOth = B.plane(BP_MASS)[other].y;
                    /* the contact basis of this step: the fixed point of the reference's re-normalisation of the stored normal
                     * (manifold.cpp:185-186 runs on every computeConstraint call; Manifold::initialize left it normalised once, so the
                     * sequence is constant after one or two calls), and its tangents (manifold.cpp:43-49) */
                    f3 nrm = mk3(nn.x, nn.y, nn.z);
                    for (int k = 0; k < 4; ++k) {
                        const f3 n2 = normalize_safe(nrm, mk3(0.0f, 1.0f, 0.0f));
                        const bool same = fu(n2.x) == fu(nrm.x) && fu(n2.y) == fu(nrm.y) && fu(n2.z) == fu(nrm.z);
                        nrm = n2;
                        if (same) break;
                    }
                    f3 t1, t2;
                    contact_tangents(nrm, t1, t2);
                    const float invMassSum = isA ? imOwn + imOth : imOth + imOwn;      /* manifold.cpp:199-201: invMassA + invMassB */
                    const float massScale = (invMassSum > 1.0e-6f) ? (1.0f / invMassSum) : 1.0f;
                    const float cap = kNormalForceCap * massScale;
                    const bool otherSwept = B.solved[other] != 0;
                    const bool first = !otherSwept || F.color[body] < F.color[other];
                    const f3 rOwn = isA ? mk3(ra.x, ra.y, ra.z) : mk3(rb.x, rb.y, rb.z);
                    const f3 rOth = isA ? mk3(rb.x, rb.y, rb.z) : mk3(ra.x, ra.y, ra.z);
                    c0 = make_float4(rOwn.x, rOwn.y, rOwn.z, ra.w);
                    c1 = make_float4(rOth.x, rOth.y, rOth.z, rb.w);
                    c2 = make_float4(nrm.x, nrm.y, nrm.z, nn.w);
                    c3 = make_float4(t1.x, t1.y, t1.z, M.fric[m]);
                    c4 = make_float4(t2.x, t2.y, t2.z, cap);
                    partner = other; slot = s;
                    flags = FL_VALID | (isA ? FL_ISA : 0) | (first ? FL_FIRST : 0) | (otherSwept ? 0 : FL_SOLE);
                    if (first) {                                      /* the contact's state records start from the Manifold table */
                        const float4 lam = M.cLam[s], pen = M.cPen[s];
                        const int stick = M.cFS[s].y != 0 ? 1 : 0;
                        F.xS[2 * (size_t)s] = make_float4(lam.x, lam.y, lam.z, __int_as_float(stick));
                        F.xS[2 * (size_t)s + 1] = make_float4(pen.x, pen.y, pen.z, __int_as_float(0));
                    }
                } else if (inc_kind(ent) == FK_JOINT) {
                    /* one half of a Joint (joint.cpp:68-139): rows 0-2 (anchor) or 3-5 (orientation) */
                    const int idx = inc_index(ent), g = idx >> 1, part = idx & 1;
                    const bool isA = inc_isA(ent) != 0;
                    const GForce& jf = F.gf[g];
                    const int other = isA ? jf.bodyB : jf.bodyA;
                    const bool world = other < 0;
                    const bool otherSwept = !world && B.solved[other] != 0;
                    const bool first = !otherSwept || F.color[body] < F.color[other];
                    const float* rO = isA ? jf.rA : jf.rB; const float* rT = isA ? jf.rB : jf.rA;
                    c0 = make_float4(rO[0], rO[1], rO[2], jf.stiffness[3 * part + 0]);
                    c1 = make_float4(rT[0], rT[1], rT[2], jf.stiffness[3 * part + 1]);
                    c2 = make_float4(jf.q0[0], jf.q0[1], jf.q0[2], jf.stiffness[3 * part + 2]);
                    c3 = make_float4(jf.q0[3], jf.motor[3 * part + 0], jf.motor[3 * part + 1], jf.motor[3 * part + 2]);
                    partner = other; slot = F.jointSlotBase + idx;
                    flags = FL_VALID | FL_JOINT | (part ? FL_PART1 : 0) | (world ? FL_WORLD : 0) | ((jf.pad & 1) ? FL_FIXANG : 0) | (isA ? FL_ISA : 0) | (first ? FL_FIRST : 0) | (otherSwept ? 0 : FL_SOLE);
                    if (first) {
                        F.xS[2 * (size_t)slot] = make_float4(jf.lambda[3 * part + 0], jf.lambda[3 * part + 1], jf.lambda[3 * part + 2], __int_as_float(0));
                        F.xS[2 * (size_t)slot + 1] = make_float4(jf.penalty[3 * part + 0], jf.penalty[3 * part + 1], jf.penalty[3 * part + 2], __int_as_float(0));
                    }
                } else if (inc_kind(ent) == FK_SPRING && F.gf[inc_index(ent) & 0xFFFFFF].kind == FK_SPRING) {
                    /* a Spring (spring.cpp:33-90): one row */
                    const int g = inc_index(ent) & 0xFFFFFF;
                    const bool isA = inc_isA(ent) != 0;
                    const GForce& sf = F.gf[g];
                    const int other = isA ? sf.bodyB : sf.bodyA;
                    const bool world = other < 0;
                    const bool otherSwept = !world && B.solved[other] != 0;
                    const bool first = !otherSwept || F.color[body] < F.color[other];
                    const float* rO = isA ? sf.rA : sf.rB; const float* rT = isA ? sf.rB : sf.rA;
                    c0 = make_float4(rO[0], rO[1], rO[2], sf.stiffness[0]);
                    c1 = make_float4(rT[0], rT[1], rT[2], sf.restLength);
                    c2 = make_float4(sf.fmin[0], sf.fmax[0], sf.motor[0], 0.f);
                    partner = other; slot = F.jointSlotBase + 2 * g;
                    flags = FL_VALID | FL_JOINT | FL_SPRING | (world ? FL_WORLD : 0) | (isA ? FL_ISA : 0) | (first ? FL_FIRST : 0) | (otherSwept ? 0 : FL_SOLE);
                    if (first) {
                        F.xS[2 * (size_t)slot] = make_float4(sf.lambda[0], 0.f, 0.f, __int_as_float(0));
                        F.xS[2 * (size_t)slot + 1] = make_float4(sf.penalty[0], 0.f, 0.f, __int_as_float(0));
                    }
                }
            }
        }
        c5 = make_float4(__int_as_float(partner), __int_as_float(slot), __int_as_float(flags), 0.f);
        const size_t at = (size_t)t * 64 + lane;
        /* (a wave of this kernel is one tile: the lanes that hold an entry, as a mask beside the tile — the sweep does not load the constants of the others: a
         * quarter of a settled pile's lanes are padding) */
        const unsigned long long vm = __ballot((flags & FL_VALID) != 0);
        if (lane == 0) F.tileMask[t] = vm;
        if (flags & FL_VALID) {
            F.slotC[0 * SC + at] = c0; F.slotC[1 * SC + at] = c1; F.slotC[2 * SC + at] = c2;
            F.slotC[3 * SC + at] = c3; F.slotC[4 * SC + at] = c4; F.slotC[5 * SC + at] = c5;
        }
    }
}

/* =====================================================================================================================
 * k_fast_sweep
 * ===================================================================================================================== */

/* ---- sums over the lanes of a sub-group: xor butterflies on the DPP network (quad_perm, row_half_mirror, row_mirror),
 * ds_bpermute beyond a row of 16.  Every lane of the sub-group ends with the same bits (IEEE addition is commutative). ---- */
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }

template <int G>
__device__ __forceinline__ float group_sum(float v)
{
    if (G >= 4) {
        v = v + dpp_f<0xB1>(v);       /* quad_perm [1,0,3,2] */
        v = v + dpp_f<0x4E>(v);       /* quad_perm [2,3,0,1] */
    }
    if (G >= 8) v = v + dpp_f<0x141>(v);    /* row_half_mirror */
    if (G >= 16) v = v + dpp_f<0x140>(v);   /* row_mirror */
    if (G >= 32) v = v + __shfl_xor(v, 16, 64);
    if (G >= 64) v = v + __shfl_xor(v, 32, 64);
    return v;
}

template <int G>
__device__ __forceinline__ void group_sum27(float* acc)
{
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = group_sum<G>(acc[i]);
}

/* Manifold::computeConstraint for one contact (manifold.cpp:183-243) on the step's contact basis: bias terms, force cap and
 * basis come in as constants, the rest is contact_constraint_post of avbd_rows.h on scalars. */
struct FastRows { float lam[3], pen[3], C[3], fmin[3], fmax[3]; bool stick; float penetration; };

__device__ __forceinline__ void fast_constraint(FastRows& c, f3 n, f3 t1, f3 t2, float biasN, float biasT1, float biasT2, float cap, f3 pA, f3 pB, float fric)
{
    const f3 d = pA - pB;
    const float separation = dot3(d, n) - kNormalMargin;
    const float slip1 = dot3(d, t1);
    const float slip2 = dot3(d, t2);
    c.C[0] = separation + biasN;
    c.fmin[0] = -cap;
    c.fmax[0] = 0.0f;
    c.C[1] = slip1 + biasT1;
    c.C[2] = slip2 + biasT2;
    const float warmMag = fabsf(rmin(c.lam[0], 0.0f));
    const float trial = c.pen[0] * c.C[0] + c.lam[0];
    const float trialMag = fabsf(rmin(trial, 0.0f));
    float nMag = rmax(warmMag, trialMag);
    nMag = rmin(nMag, cap);
    float mu = fric;
    if (!c.stick) mu *= 0.9f;
    const float limit = mu * nMag;
    const float lt1 = c.lam[1], lt2 = c.lam[2];
    const float tanMag = sqrtf(lt1 * lt1 + lt2 * lt2);
    if (tanMag > limit && tanMag > 1.0e-8f) {
        const float s = limit / tanMag;
        c.lam[1] *= s;
        c.lam[2] *= s;
    }
    c.fmin[1] = -limit; c.fmax[1] = limit;
    c.fmin[2] = -limit; c.fmax[2] = limit;
    const float slipSq = c.C[1] * c.C[1] + c.C[2] * c.C[2];
    const float tanSq = c.lam[1] * c.lam[1] + c.lam[2] * c.lam[2];
    const float stickSq = kStickThresh * kStickThresh;
    c.stick = (slipSq <= stickSq) && (tanSq <= limit * limit + 1.0e-8f);
    c.penetration = rmax(0.0f, -dot3(d, n));
}

/* the dual update of one contact (solver.cpp:411-430, rowPenaltyGain :94-125) */
__device__ __forceinline__ void fast_dual(FastRows& c, f3 n, f3 t1, f3 t2, float biasN, float biasT1, float biasT2, float cap, f3 pA, f3 wA, f3 pB, f3 wB, float fric, float beta)
{
    fast_constraint(c, n, t1, t2, biasN, biasT1, biasT2, cap, pA, pB, fric);
    const f3 basis[3] = {n, t1, t2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lu = rclamp(c.pen[k] * c.C[k] + c.lam[k], c.fmin[k], c.fmax[k]);
        const bool active = lu > c.fmin[k] && lu < c.fmax[k];
        c.lam[k] = lu;
        if (active) {
            /* computeDerivatives of both endpoints (manifold.cpp:247-271): Jl = +-basis, Ja = +-(w x basis) */
            const f3 JlA = basis[k] * 1.0f, JaA = cross3(wA, basis[k]) * 1.0f;
            const f3 JlB = basis[k] * -1.0f, JaB = cross3(wB, basis[k]) * -1.0f;
            float lw = 0.0f, aw = 0.0f;
            lw += lensq3(JlA); aw += lensq3(JaA);
            lw += lensq3(JlB); aw += lensq3(JaB);
            const float betaRow = row_penalty_gain(lw, aw, beta);
            c.pen[k] = rmin(c.pen[k] + betaRow * fabsf(c.C[k]), kManifoldPenaltyCap);
        }
    }
}

/* -DAVBD_ASM_MARKS: comments in the generated assembly (one set per lanes-per-body instantiation) that tools/kernel_resources.py splits the instruction mix at */
#ifdef AVBD_ASM_MARKS
#define FAST_MARK(name) asm volatile("; AVBD_MARK " #name "_g%0" :: "n"(G) : "memory")
#else
#define FAST_MARK(name) do { } while (0)
#endif

#ifdef AVBD_DF_STATS
constexpr int kFastStatTiles = 2048, kFastStatIts = 64;
__device__ unsigned gFastTimes[kFastStatTiles * kFastStatIts * 4];     /* per (tile, iteration): visit start, first pass done, inputs of the second pass in, published (10 ns ticks) */
#define FAST_STAMP_(k) do { if (t < kFastStatTiles && it < kFastStatIts - 1 && (threadIdx.x & 63) == 0) gFastTimes[((size_t)t * kFastStatIts + it) * 4 + (k)] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
#if AVBD_DF_STATS == 2      /* the path behind the last arrival: inputs in (0), rows done (1), solved (2), published (3) */
#define FAST_STAMP(k) do { if ((k) == 3) FAST_STAMP_(3); } while (0)
#define FAST_STAMPB(k) FAST_STAMP_(k)
#else
#define FAST_STAMP(k) FAST_STAMP_(k)
#define FAST_STAMPB(k) do { } while (0)
#endif
#define FAST_STAT_COLOUR(colour) do { if (t < kFastStatTiles && it == 0 && (threadIdx.x & 63) == 0) gFastTimes[((size_t)t * kFastStatIts + kFastStatIts - 1) * 4] = (unsigned)(colour); } while (0)   /* (the last iteration slot holds the tile's colour) */
#else
#define FAST_STAMP(k) do { } while (0)
#define FAST_STAMPB(k) do { } while (0)
#define FAST_STAT_COLOUR(colour) do { } while (0)
#endif

/* One half of a Joint for one endpoint (joint.cpp:68-139; rows 3p .. 3p + 2): computeConstraint (no alpha, no mutation: C alone), the dual update of the
 * previous iteration when this is the joint's first visitor (solver.cpp:411-430 with the two-endpoint rowPenaltyGain :94-125), computeDerivatives for THIS
 * endpoint and the row forces (solver.cpp:379-381 with the rows' stiffness and motor).  Bug-compatible angular Jacobian sign unless FL_FIXANG. */
struct FastJointOut { f3 Jl[3], Ja[3]; float f[3], C[3]; };
__device__ __forceinline__ void fast_joint_rows(FastRows& c, FastJointOut& o, int flags, f3 wOwn, f3 pOwn, q4 rot, f3 rOth, f3 opos, q4 orot, q4 q0,
                                                const float stiff[3], const float motor[3], bool dual, float beta)
{
    const bool isA = (flags & FL_ISA) != 0, part1 = (flags & FL_PART1) != 0, world = (flags & FL_WORLD) != 0, fix = (flags & FL_FIXANG) != 0;
    const bool hasA = !world;
    const f3 wOth = world ? zero3() : qrotate(orot, rOth);
    const f3 pOth = world ? rOth : opos + wOth;                      /* bodyA == nullptr: rA is the world anchor (joint.cpp:78-80) */
    const f3 wA = isA ? wOwn : wOth, wB = isA ? wOth : wOwn;
    if (!part1) {
        const f3 lc = isA ? pOwn - pOth : pOth - pOwn;               /* pA - pB */
        o.C[0] = lc.x; o.C[1] = lc.y; o.C[2] = lc.z;
    } else {
        const q4 qA = isA ? rot : (world ? identq() : orot), qB = isA ? orot : rot;
        const q4 cur = qconj(qA) * qB;
        const q4 dq = cur * qconj(q0);
        const f3 ac = mk3(dq.x, dq.y, dq.z) * 2.0f;
        o.C[0] = ac.x; o.C[1] = ac.y; o.C[2] = ac.z;
    }
    const float sign = isA ? 1.0f : -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const f3 ax = mk3(k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f);
        if (dual && stiff[k] == FLT_MAX) {
            const float lu = rclamp(c.pen[k] * o.C[k] + c.lam[k], -FLT_MAX, FLT_MAX);
            const bool active = lu > -FLT_MAX && lu < FLT_MAX;
            c.lam[k] = lu;
            if (active) {
                float lw = 0.0f, aw = 0.0f;
                if (!part1) {
                    if (hasA) { lw += lensq3(ax * 1.0f); aw += lensq3(cross3(wA, ax) * 1.0f); }
                    lw += lensq3(ax * -1.0f); aw += lensq3(cross3(wB, ax) * -1.0f);
                } else {
                    if (hasA) { lw += lensq3(zero3()); aw += lensq3(ax * (fix ? -1.0f : 1.0f)); }
                    lw += lensq3(zero3()); aw += lensq3(ax * (fix ? 1.0f : -1.0f));
                }
                const float betaRow = row_penalty_gain(lw, aw, beta);
                c.pen[k] = rmin(c.pen[k] + betaRow * fabsf(o.C[k]), kPenaltyMax);
            }
        }
        if (!part1) { o.Jl[k] = ax * sign; o.Ja[k] = cross3(wOwn, ax) * sign; }
        else { o.Jl[k] = zero3(); o.Ja[k] = ax * (fix ? -sign : sign); }
        const float lw = (stiff[k] == FLT_MAX) ? c.lam[k] : 0.0f;
        const float desired = c.pen[k] * o.C[k] + lw + motor[k];
        o.f[k] = rclamp(desired, -FLT_MAX, FLT_MAX);
    }
}

/* A Spring for one endpoint (spring.cpp:33-90; one row): computeConstraint (C = |pA - pB| - rest), the dual update of the previous iteration when the row is a hard
 * one (stiffness FLT_MAX; a spring's usual finite stiffness has no lambda and no dual, solver.cpp:416-418), computeDerivatives for THIS endpoint, the row force. */
__device__ __forceinline__ void fast_spring_rows(FastRows& c, FastJointOut& o, int flags, f3 pos, q4 rot, f3 rOwn, f3 opos, q4 orot, f3 rOth,
                                                 float rest, float stiff, float motor, float fmin, float fmax, bool dual, float beta)
{
    const bool isA = (flags & FL_ISA) != 0, hasA = (flags & FL_WORLD) == 0;
    const f3 posA = isA ? pos : opos, posB = isA ? opos : pos;
    const q4 rotA = isA ? rot : orot, rotB = isA ? orot : rot;
    SpringRows sr;
    sr.rA = isA ? rOwn : rOth; sr.rB = isA ? rOth : rOwn; sr.restLength = rest; sr.C = 0.0f;
    spring_compute_constraint(sr, hasA, posA, rotA, posB, rotB);
    o.C[0] = sr.C; o.C[1] = 0.0f; o.C[2] = 0.0f;
    if (dual && stiff == FLT_MAX) {
        const float lu = rclamp(c.pen[0] * sr.C + c.lam[0], fmin, fmax);
        const bool active = lu > fmin && lu < fmax;
        c.lam[0] = lu;
        if (active) {
            float lw = 0.0f, aw = 0.0f;
            f3 Jl, Ja;
            if (hasA) { spring_compute_derivatives(sr, true, hasA, posA, rotA, posB, rotB, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja); }
            spring_compute_derivatives(sr, false, hasA, posA, rotA, posB, rotB, Jl, Ja); lw += lensq3(Jl); aw += lensq3(Ja);
            const float betaRow = row_penalty_gain(lw, aw, beta);
            c.pen[0] = rmin(c.pen[0] + betaRow * fabsf(sr.C), kPenaltyMax);
        }
    }
    spring_compute_derivatives(sr, isA, hasA, posA, rotA, posB, rotB, o.Jl[0], o.Ja[0]);
    const float lw = (stiff == FLT_MAX) ? c.lam[0] : 0.0f;
    o.f[0] = rclamp(c.pen[0] * sr.C + lw + motor, fmin, fmax);
    o.Jl[1] = zero3(); o.Ja[1] = zero3(); o.f[1] = 0.0f; o.Jl[2] = zero3(); o.Ja[2] = zero3(); o.f[2] = 0.0f;
    c.pen[1] = 0.0f; c.pen[2] = 0.0f; c.lam[1] = 0.0f; c.lam[2] = 0.0f;
}

struct FastCtx {
    __amdgpu_buffer_rsrc_t rPose, rS;
    const FastBuffers* F;
    FastPlan* plan;
    Counters* cnt;
    BodyTable B; ManifoldTable M;
    float invDt2, beta;
    int stamps;
    GForce* gf; int jointSlotBase;
    int sleepSel;
    bool local;          /* every wave of this launch runs on one XCD: records are handed over with plain stores */
};

__device__ __forceinline__ float fast_alpha(const SolverParams& P, int it) { return P.postStabilize ? (it < P.iterations ? 1.0f : 0.0f) : P.alpha; }

/* quad_perm:[c,c,c,c]: the value lane (lane & ~3) + c holds */
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), Q | (Q << 2) | (Q << 4) | (Q << 6), 0xf, 0xf, true)); }
template <int Q>
__device__ __forceinline__ f3 quad_bcast3(f3 v) { return mk3(quad_bcast<Q>(v.x), quad_bcast<Q>(v.y), quad_bcast<Q>(v.z)); }

/* one of four values by the lane's position in its quad, as three v_cndmask (the empty asm keeps the compiler from turning the chain into a table in scratch) */
__device__ __forceinline__ float pick4(float a, float b, float c, float d, int q)
{
    float r = a;
    r = q == 1 ? b : r; asm volatile("" : "+v"(r));
    r = q == 2 ? c : r; asm volatile("" : "+v"(r));
    r = q == 3 ? d : r;
    return r;
}

/* maths.h:104 with the independent divisions side by side on the lanes of a quad (c = lane & 3): L10 | L20, then L21, then z.x | z.y | z.z —
 * the same quotients of the same bits, every lane of the quad returns x (all four hold the same A and b) */
__device__ __forceinline__ f3 solve_ldlt3_quadlanes(m33 A, f3 b, int c)
{
    const f3 L0 = A.c0;
    bool bad = fabsf(L0.x) < FLT_EPSILON;
    const float d0 = L0.x;
    const float q1 = (c == 0 ? L0.y : L0.z) / d0;
    const float L10 = quad_bcast<0>(q1), L20 = quad_bcast<1>(q1);
    const f3 L1 = A.c1 - L0 * L10;
    bad |= fabsf(L1.y) < FLT_EPSILON;
    const float d1 = L1.y;
    const float L21 = L1.z / d1;
    const f3 L2 = A.c2 - L0 * L20 - L1 * L21;
    bad |= fabsf(L2.z) < FLT_EPSILON;
    const float d2 = L2.z;
    f3 y; y.x = b.x; y.y = b.y - L10 * y.x; y.z = b.z - L20 * y.x - L21 * y.y;
    const float zq = (c == 0 ? y.x : (c == 1 ? y.y : y.z)) / (c == 0 ? d0 : (c == 1 ? d1 : d2));
    f3 z; z.x = quad_bcast<0>(zq); z.y = quad_bcast<1>(zq); z.z = quad_bcast<2>(zq);
    f3 x; x.z = z.z; x.y = z.y - L21 * x.z; x.x = z.x - L10 * x.y - L20 * x.z;
    return bad ? zero3() : x;
}

/* maths.h:104 in two halves for the latency form: the factorisation (what solve_ldlt3_branchless computes before it touches b; L10 | L20 side by side on the lanes
 * of a quad) and the substitution — per lane for its own right-hand side, or with its three divisions side by side on the lanes of a quad that share b.  The same
 * quotients, products and differences as the one-piece solve. */
__device__ __forceinline__ void fast_ldlt3_factor(m33 A, int c, float& L10, float& L20, float& L21, float& d0, float& d1, float& d2, bool& bad)
{
    const f3 L0 = A.c0;
    bad = fabsf(L0.x) < FLT_EPSILON;
    d0 = L0.x;
    const float q1 = (c == 0 ? L0.y : L0.z) / d0;
    L10 = quad_bcast<0>(q1); L20 = quad_bcast<1>(q1);
    const f3 L1 = A.c1 - L0 * L10;
    bad |= fabsf(L1.y) < FLT_EPSILON;
    d1 = L1.y;
    L21 = L1.z / d1;
    const f3 L2 = A.c2 - L0 * L20 - L1 * L21;
    bad |= fabsf(L2.z) < FLT_EPSILON;
    d2 = L2.z;
}
__device__ __forceinline__ f3 fast_ldlt3_subst_lane(float L10, float L20, float L21, float d0, float d1, float d2, bool bad, f3 b)
{
    f3 y; y.x = b.x; y.y = b.y - L10 * y.x; y.z = b.z - L20 * y.x - L21 * y.y;
    f3 z; z.x = y.x / d0; z.y = y.y / d1; z.z = y.z / d2;
    f3 x; x.z = z.z; x.y = z.y - L21 * x.z; x.x = z.x - L10 * x.y - L20 * x.z;
    return bad ? zero3() : x;
}
__device__ __forceinline__ f3 fast_ldlt3_subst_quad(float L10, float L20, float L21, float d0, float d1, float d2, bool bad, f3 b, int c)
{
    f3 y; y.x = b.x; y.y = b.y - L10 * y.x; y.z = b.z - L20 * y.x - L21 * y.y;
    const float zq = (c == 0 ? y.x : (c == 1 ? y.y : y.z)) / (c == 0 ? d0 : (c == 1 ? d1 : d2));
    f3 z; z.x = quad_bcast<0>(zq); z.y = quad_bcast<1>(zq); z.z = quad_bcast<2>(zq);
    f3 x; x.z = z.z; x.y = z.y - L21 * x.z; x.x = z.x - L10 * x.y - L20 * x.z;
    return bad ? zero3() : x;
}

/* what a visit reads that no other wave writes: the tile's descriptor, the lane's constants (only lanes that hold an entry load them), the sub-group's body
 * constants.  A wave loads the NEXT tile's while it computes the current one (fast_visit's `next`); a wave with one tile loads them once for the launch. */
struct FastTileRegs { float4 c0, c1, c2, c3, c4, c5, b0, b1, b2, b3; int2 td; int t; };
template <int GK = 0>   /* GK: the tile's lanes per body when the caller knows them (the descriptor is in R.td already), 0: read the descriptor */
__device__ __forceinline__ void fast_load_tile(const FastBuffers& F, int t, FastTileRegs& R)
{
    const int lane = threadIdx.x & 63;
    const size_t SC = (size_t)F.slotCap, at = (size_t)t * 64 + lane;
    R.t = t;
    if (GK == 0) R.td = F.tiles[t];
    const int cls = GK == 0 ? ((R.td.y >> 16) & 0xFF) : (GK == 1 ? 0 : 1), G = GK == 0 ? fast_class_lanes(cls) : GK, nb = R.td.y & 0xFFFF;
    const int j = lane / G;
    R.c0 = make_float4(0.f, 0.f, 0.f, 0.f); R.c1 = R.c0; R.c2 = R.c0; R.c3 = R.c0; R.c4 = R.c0; R.c5 = make_float4(__int_as_float(-1), 0.f, 0.f, 0.f);
    if (cls > 0) {
        const unsigned long long vm = F.tileMask[t];                  /* lanes that hold an entry */
        if ((vm >> lane) & 1ull) { R.c5 = F.slotC[5 * SC + at]; R.c0 = F.slotC[0 * SC + at]; R.c1 = F.slotC[1 * SC + at]; R.c2 = F.slotC[2 * SC + at]; R.c3 = F.slotC[3 * SC + at]; R.c4 = F.slotC[4 * SC + at]; }
    }
    const int bp = R.td.x + (j < nb ? j : 0);
    R.b0 = F.bodyC[(size_t)0 * F.bodyCap + bp]; R.b1 = F.bodyC[(size_t)1 * F.bodyCap + bp]; R.b2 = F.bodyC[(size_t)2 * F.bodyCap + bp]; R.b3 = F.bodyC[(size_t)3 * F.bodyCap + bp];
}

/* One visit of tile `t` in iteration `it`: the primal solve (solver.cpp:344-409) of its 64 / G bodies.
 * The lanes of a body are evaluated in two passes: first the contacts whose partner comes LATER in the colour order (and those with a partner nobody
 * sweeps) — everything they read was published in the previous iteration, and they run the dual update of that iteration —, then the contacts whose
 * partner has already moved in this iteration.  Only the second pass, the sums, the solve and the publish stand between the arrival of the last pose
 * and this body's own. */
template <int G, bool TWOPASS, bool HOIST, bool JOINTS, bool PRE>
__device__ __forceinline__ void fast_visit(const FastCtx& X, const SolverParams& P, const FastTileRegs& TRin, FastTileRegs& next, int nextT, int it, bool lastOfLaunch)
{
    const int lane = threadIdx.x & 63;
    const int j = G == 1 ? lane : lane / G, r = G == 1 ? 0 : lane % G;
    FastTileRegs TRown;
    if (!PRE) { TRown.t = TRin.t; TRown.td = TRin.td; fast_load_tile<G>(*X.F, TRin.t, TRown); }   /* (PRE: the caller loaded them behind the previous visit's last wait) */
    const FastTileRegs& TR = PRE ? TRin : TRown;
    const int t = TR.t;
    const int2 td = TR.td;
    const int nb = td.y & 0xFFFF;
    const bool hasBody = j < nb;
    const FastBuffers& F = *X.F;
    bool prefetched = false;

    /* constants of the lane's incidence entry (coalesced) and of the sub-group's body */
    const float4 c0 = TR.c0, c1 = TR.c1, c2 = TR.c2, c3 = TR.c3, c4 = TR.c4, c5 = TR.c5;
    const int partner = __float_as_int(c5.x), slot = __float_as_int(c5.y), flags = __float_as_int(c5.z);
    const bool valid = (flags & FL_VALID) != 0, isA = (flags & FL_ISA) != 0, first = (flags & FL_FIRST) != 0, sole = (flags & FL_SOLE) != 0;
    const float4 b0 = TR.b0, b1 = TR.b1, b2 = TR.b2, b3 = TR.b3;
    const int body = __float_as_int(b3.w);
    const f3 npos = mk3(b0.x, b0.y, b0.z); const float mass = b0.w;
    const q4 nrot = mkq(b1.x, b1.y, b1.z, b1.w);
    const f3 Id = mk3(b2.x, b2.y, b2.z), Jd = mk3(b3.x, b3.y, b3.z);

    FAST_MARK(fv_start);
    FAST_STAMP(0); FAST_STAT_COLOUR((td.y >> 24) & 0xFF);
    /* the body's own pose: this wave published it (epoch it) */
    f3 pos; q4 rot;
    {
        const fv4u o0 = fld(X.rPose, body * 32), o1 = fld(X.rPose, body * 32 + 16);
        (void)pose_decode(o0, o1, (unsigned)it, pos, rot);
    }
    /* world inertia and its inverse (rigid.cpp:51-59: R * I * transpose(R), I diagonal) */
    const m33 R = quat_to_mat(rot);
    const m33 Rt = mtranspose(R);
    const m33 RI = mkm(R.c0 * Id.x, R.c1 * Id.y, R.c2 * Id.z), RJ = mkm(R.c0 * Jd.x, R.c1 * Jd.y, R.c2 * Jd.z);
    const m33 Iw = RI * Rt, invIw = RJ * Rt;

    /* the body's system, summed per lane and then over the sub-group: acc[0..20] the J J^T penalty part of lhs (ll lower, la, aa lower), gy[] the gyroscopic
     * diagonal, rh[] the right-hand side.  Every form of the kernel adds them up in this one order: (sum of acc) + the body's own terms, then + (sum of gy). */
    float acc[21], gy[3], rh[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) gy[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) rh[i] = 0.0f;

    /* the body's own terms (solver.cpp:350-369).  (Computed here, ahead of the waits, in every form: behind them the one-pass form takes 230 registers instead of
     * 193 and the two-waves-per-SIMD form spills 80 B instead of 32) */
    const float invDt2 = X.invDt2;
    const float mdt = mass * invDt2;
    const m33 aa0 = Iw * invDt2;
    f3 rl0 = ((pos - npos) * invDt2) * mass;
    const q4 qErr = rot * qconj(nrot);
    f3 rotErr = mk3(qErr.x, qErr.y, qErr.z);
    rotErr = rotErr * 2.0f;
    if (qErr.w < 0.0f) rotErr = -rotErr;
    f3 ra0 = Iw * (rotErr * invDt2);
    /* latency form: evaluated HERE — the compiler sinks what is first used behind the waits (these sixty instructions sat between the nine sums and the substitution,
     * on the path behind the last arrival); the empty asm pins the six values in front of the pass loop */
    if (HOIST || !JOINTS) asm volatile("" : "+v"(rl0.x), "+v"(rl0.y), "+v"(rl0.z), "+v"(ra0.x), "+v"(ra0.y), "+v"(ra0.z));

    /* SPLIT (latency form, contacts): what the second pass's lanes need arrives in two parts — the contact's records (lambda, penalty: written by the partner
     * BEFORE its sums and its solve) and the partner's new pose (after them).  Everything that needs only the first — the J J^T penalty part of lhs, its 21
     * sums, the factorisation of ll, ll^-1 la and the Schur complement up to the gyroscopic diagonal — runs while the pose is still on its way; behind the pose
     * come the rows' forces, nine sums, one substitution, the 3x3 Schur solve and the pose update.  Same operations on the same values as the other forms. */
    constexpr bool SPLIT = TWOPASS && HOIST && !JOINTS && G >= 4;
    /* (measured and not kept: the split taken at run time only where a tile HAS a second pass — the first colour's and the isolated row's tiles have everything behind their one
     * wait anyway, and the one-piece sums and solve are shorter than the two halves back to back —: both paths in one kernel, 172 -> 206 KB of code and 146 -> 179 spill
     * registers, Stress1000 208 -> 224 us, 8 000 cubes 430 -> 455) */
    float fL10 = 0.0f, fL20 = 0.0f, fL21 = 0.0f, fd0 = 1.0f, fd1 = 1.0f, fd2 = 1.0f; bool fbad = false;
    f3 col0 = zero3(), col1 = zero3(), col2 = zero3(), schurOff = zero3(), pDiag = zero3(), aaDiag = zero3();
    m33 al = mkm(zero3(), zero3(), zero3());

    if (G > 1) {
        const f3 rOwn = mk3(c0.x, c0.y, c0.z), rOth = mk3(c1.x, c1.y, c1.z);
        const f3 n = mk3(c2.x, c2.y, c2.z), t1 = mk3(c3.x, c3.y, c3.z), t2 = mk3(c4.x, c4.y, c4.z);
        const float fric = c3.w, cap = c4.w;
        const f3 wOwn = qrotate(rot, rOwn);
        const f3 pOwn = pos + wOwn;
        const bool moving = valid && !sole;
        const unsigned needPartner = (unsigned)(first ? it : it + 1);
        const unsigned needVisit = (unsigned)(first ? 2 * it : 2 * it + 1);
        /* lhs += (J J^T) penalty: ll lower, la, aa lower (solver.cpp:387-391; al = la^T, the upper triangles are never read, maths.h:104).  (Measured: a
         * branch-free form — a failing row adds zero products — takes 9 more registers and spills in the two-waves-per-SIMD form; no time gained) */
        auto lhs_add = [&](f3 Jl, f3 Ja, float pen) {
            const float J6[6] = {Jl.x, Jl.y, Jl.z, Ja.x, Ja.y, Ja.z};
            acc[0] += (Jl.x * Jl.x) * pen; acc[1] += (Jl.y * Jl.x) * pen; acc[2] += (Jl.z * Jl.x) * pen;
            acc[3] += (Jl.y * Jl.y) * pen; acc[4] += (Jl.z * Jl.y) * pen; acc[5] += (Jl.z * Jl.z) * pen;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
#pragma unroll
                for (int ii = 0; ii < 3; ++ii) acc[6 + 3 * jj + ii] += (J6[ii] * J6[3 + jj]) * pen;
            acc[15] += (Ja.x * Ja.x) * pen; acc[16] += (Ja.y * Ja.x) * pen; acc[17] += (Ja.z * Ja.x) * pen;
            acc[18] += (Ja.y * Ja.y) * pen; acc[19] += (Ja.z * Ja.y) * pen; acc[20] += (Ja.z * Ja.z) * pen;
        };
        /* two-waves-per-SIMD form: the angular Jacobians and the gyroscopic factors of the lane's three rows (functions of the body's own pose) are kept across the wait — 18
         * registers; their 63 products per row still are recomputed in each pass (ninety more would spill) */
        constexpr bool KEEPJ = TWOPASS && !HOIST && !JOINTS;
        f3 keepJa[3] = {zero3(), zero3(), zero3()}, keepG[3] = {zero3(), zero3(), zero3()};
        if (KEEPJ) {
            const float sign = isA ? 1.0f : -1.0f;
            const f3 basis[3] = {n, t1, t2};
#pragma unroll
            for (int k = 0; k < 3; ++k) { keepJa[k] = cross3(wOwn, basis[k]) * sign; keepG[k] = abs3(cross3(keepJa[k], invIw * keepJa[k])); }
        }
        fv4u sl = fv4u{0u, 0u, 0u, 0u}, sp = sl;
#pragma unroll 1
        for (int pass = TWOPASS ? 0 : 1; pass < 2; ++pass) {
            const bool mine = valid && (!TWOPASS || first == (pass == 0));
            if (pass == 1) FAST_STAMP(1);
            FAST_MARK(fv_pass_top);
            const bool early = SPLIT && pass == 1;
            if (early) {
                if (__any(mine)) {
                    /* the records of the second pass's contacts (their first visitor wrote them in this iteration) ... */
                    const unsigned long long t0 = X.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
                    bool ok = !mine;
                    for (;;) {
                        if (!ok) {
                            sl = fld(X.rS, slot * 32); sp = fld(X.rS, slot * 32 + 16);
                            ok = (sl.w >> 1) == needVisit && sp.w == (unsigned)(it + 1);
                        }
                        if (!X.stamps || __all(ok)) break;
                        if (X.sleepSel == 0) __builtin_amdgcn_s_sleep(1); else if (X.sleepSel == 1) __builtin_amdgcn_s_sleep(4); else if (X.sleepSel == 2) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(64);
                        if (__builtin_amdgcn_s_memrealtime() - t0 > kFastTimeoutTicks) {
                            if (!ok) { atomicOr(&X.plan->bad, 256); X.plan->errInfo[0] = body; X.plan->errInfo[1] = partner; X.plan->errInfo[2] = it; X.plan->errInfo[3] = (int)sl.w; X.plan->errInfo[4] = (int)needVisit; X.plan->errInfo[5] = 2; X.plan->errInfo[6] = (int)needPartner; X.plan->errInfo[7] = t; }
                            break;
                        }
                    }
                    /* ... and their part of lhs */
                    if (mine) {
                        const float sign = isA ? 1.0f : -1.0f;
                        const f3 basis[3] = {n, t1, t2};
                        const float pen3[3] = {uf(sp.x), uf(sp.y), uf(sp.z)};
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const bool penOk = (pen3[k] > 0.0f) && (fabsf(pen3[k]) <= FLT_MAX);
                            if (penOk) lhs_add(basis[k] * sign, cross3(wOwn, basis[k]) * sign, pen3[k]);
                        }
                    }
                }
                FAST_MARK(fv_presolve);
                /* lhs without the gyroscopic diagonal, the factorisation of ll, ll^-1 la, the Schur complement's pose-independent part */
#pragma unroll
                for (int i = 0; i < 21; ++i) acc[i] = group_sum<G>(acc[i]);
                acc[0] += mdt; acc[3] += mdt; acc[5] += mdt;
                acc[15] += aa0.c0.x; acc[16] += aa0.c0.y; acc[17] += aa0.c0.z; acc[18] += aa0.c1.y; acc[19] += aa0.c1.z; acc[20] += aa0.c2.z;
                m33 ll, la;
                ll.c0 = mk3(acc[0], acc[1], acc[2]); ll.c1 = mk3(acc[1], acc[3], acc[4]); ll.c2 = mk3(acc[2], acc[4], acc[5]);
                la.c0 = mk3(acc[6], acc[7], acc[8]); la.c1 = mk3(acc[9], acc[10], acc[11]); la.c2 = mk3(acc[12], acc[13], acc[14]);
                al = mtranspose(la);
                fast_ldlt3_factor(ll, lane & 3, fL10, fL20, fL21, fd0, fd1, fd2, fbad);
                {
                    const int q = lane & 3;
                    const f3 rhsq = mk3(pick4(la.c0.x, la.c1.x, la.c2.x, la.c2.x, q), pick4(la.c0.y, la.c1.y, la.c2.y, la.c2.y, q), pick4(la.c0.z, la.c1.z, la.c2.z, la.c2.z, q));
                    const f3 sol = fast_ldlt3_subst_lane(fL10, fL20, fL21, fd0, fd1, fd2, fbad, rhsq);
                    col0 = quad_bcast3<0>(sol); col1 = quad_bcast3<1>(sol); col2 = quad_bcast3<2>(sol);
                }
                {
                    const f3 p0 = al * col0, p1 = al * col1, p2 = al * col2;
                    schurOff = mk3(acc[16] - p0.y, acc[17] - p0.z, acc[19] - p1.z);       /* schur c0.y, c0.z, c1.z */
                    pDiag = mk3(p0.x, p1.y, p2.z);
                    aaDiag = mk3(acc[15], acc[18], acc[20]);
                }
            }
            if (!__any(mine)) { if (pass == 1) FAST_STAMP(2); continue; }
            FAST_MARK(fv_poll);
            /* ---- inputs of the pass: partner pose and the contact's records, one round trip; with stamps the lane polls until both carry this visit's stamps ---- */
            f3 opos = zero3(); q4 orot = identq();
            {
                /* (measured and not kept, latency form: TWO looks in flight — the second issued before the first is examined — so that a record landing just behind a
                 * look is seen half a round trip earlier: Stress1000 272 -> 299 us per launch, 8 000 cubes 476 -> 531: the second look's traffic costs more) */
                const unsigned long long t0 = X.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
                bool ok = !mine;                                       /* a lane that has its records stops looking: a waiting wave re-reads only what it still misses */
                for (;;) {
                    if (!ok) {
                        if (!early) { sl = fld(X.rS, slot * 32); sp = fld(X.rS, slot * 32 + 16); }
                        ok = true;
                        if (moving) {
                            const fv4u p0 = fld(X.rPose, partner * 32), p1 = fld(X.rPose, partner * 32 + 16);
                            ok = pose_decode(p0, p1, needPartner, opos, orot);
                            if (!early) {
                                ok &= (sl.w >> 1) == needVisit;
                                if (!first) ok &= sp.w == (unsigned)(it + 1);
                            }
                        }
                    }
                    if (!X.stamps || __all(ok)) break;
                    if (X.sleepSel == 0) __builtin_amdgcn_s_sleep(1); else if (X.sleepSel == 1) __builtin_amdgcn_s_sleep(4); else if (X.sleepSel == 2) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(64);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > kFastTimeoutTicks) {
                        if (!ok) { atomicOr(&X.plan->bad, 256); X.plan->errInfo[0] = body; X.plan->errInfo[1] = partner; X.plan->errInfo[2] = it; X.plan->errInfo[3] = (int)sl.w; X.plan->errInfo[4] = (int)needVisit; X.plan->errInfo[5] = pass; X.plan->errInfo[6] = (int)needPartner; X.plan->errInfo[7] = t; }
                        break;
                    }
                }
            }
            FAST_MARK(fv_rows);
            if (pass == 1) FAST_STAMP(2);
            FAST_STAMPB(0);
            if (pass == 1 && nextT >= 0 && nextT != t) { fast_load_tile(F, nextT, next); prefetched = true; }   /* (its loads return while this pass, the sums and the solve run) */
            if (mine) {
                f3 wOwnP = wOwn;
                float signP = isA ? 1.0f : -1.0f;
                if (TWOPASS && !HOIST && JOINTS) {   /* (the contacts-only two-pass form: since the unit is built without the SLP vectoriser the compiler may hoist there too — 256 registers, 16 B of scratch, 2 % faster) */
                    /* (two waves per SIMD: what depends on the body's own pose alone — Jacobians, their products, gyroscopic factors — is recomputed in each pass
                     * instead of being kept in ninety registers across the wait; the empty asm hides the loop invariance from the compiler) */
                    asm volatile("" : "+v"(wOwnP.x), "+v"(wOwnP.y), "+v"(wOwnP.z), "+v"(signP));
                }
                const bool isJoint = JOINTS && (flags & FL_JOINT) != 0;
                if (sole && !(JOINTS && (flags & FL_WORLD) != 0)) {   /* a partner this context does not sweep: its pose is in the body table (fixed during the launch) */
                    const float4 pp = X.B.plane(BP_POSR)[partner], pq = X.B.plane(BP_QUAT)[partner];
                    opos = mk3(pp.x, pp.y, pp.z); orot = mkq(pq.x, pq.y, pq.z, pq.w);
                }
                FastRows c;
                c.lam[0] = uf(sl.x); c.lam[1] = uf(sl.y); c.lam[2] = uf(sl.z); c.stick = (sl.w & 1u) != 0;
                c.pen[0] = uf(sp.x); c.pen[1] = uf(sp.y); c.pen[2] = uf(sp.z);
                f3 rowJl[3], rowJa[3]; float rowF[3];
                if (!isJoint) {
                    const f3 wOth = qrotate(orot, rOth);
                    const f3 pOth = opos + wOth;
                    const f3 pA = isA ? pOwn : pOth, pB = isA ? pOth : pOwn;
                    if (first && it > 0) {
                        /* the dual update of iteration it - 1 (solver.cpp:411-430), by the force's first visitor of iteration it: no pose has
                         * changed since the sweep of it - 1 ended */
                        const float bs = rclamp(1.0f - fast_alpha(P, it - 1), 0.0f, 1.0f);
                        fast_dual(c, n, t1, t2, bs * c0.w, bs * c1.w, bs * c2.w, cap, pA, isA ? wOwnP : wOth, pB, isA ? wOth : wOwnP, fric, X.beta);
                    }
                    const float bs = rclamp(1.0f - fast_alpha(P, it), 0.0f, 1.0f);
                    fast_constraint(c, n, t1, t2, bs * c0.w, bs * c1.w, bs * c2.w, cap, pA, pB, fric);
                    const float sign = signP;
                    const f3 basis[3] = {n, t1, t2};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        rowJl[k] = basis[k] * sign; rowJa[k] = KEEPJ ? keepJa[k] : cross3(wOwnP, basis[k]) * sign;   /* manifold.cpp:265-270 */
                        const float desired = c.pen[k] * c.C[k] + c.lam[k] + 0.0f;             /* solver.cpp:379-381: stiffness FLT_MAX, motor 0 */
                        rowF[k] = rclamp(desired, c.fmin[k], c.fmax[k]);
                    }
                } else {
                    FastJointOut jo;
                    if (flags & FL_SPRING) fast_spring_rows(c, jo, flags, pos, rot, rOwn, opos, orot, rOth, c1.w, c0.w, c2.z, c2.x, c2.y, first && it > 0, X.beta);
                    else {
                        const float stiff[3] = {c0.w, c1.w, c2.w}, motor[3] = {c3.y, c3.z, c3.w};
                        fast_joint_rows(c, jo, flags, wOwnP, pos + wOwnP, rot, rOth, opos, orot, mkq(c2.x, c2.y, c2.z, c3.x), stiff, motor, first && it > 0, X.beta);
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { rowJl[k] = jo.Jl[k]; rowJa[k] = jo.Ja[k]; rowF[k] = jo.f[k]; }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f3 Jl = rowJl[k], Ja = rowJa[k];
                    const float f = rowF[k];
                    const float pen = c.pen[k];
                    const bool penOk = (pen > 0.0f) && (fabsf(pen) <= FLT_MAX);
                    const float J6[6] = {Jl.x, Jl.y, Jl.z, Ja.x, Ja.y, Ja.z};
#pragma unroll
                    for (int i = 0; i < 6; ++i) rh[i] += J6[i] * f;                        /* rhs += J f (solver.cpp:383-384) */
                    if (penOk) {
                        if (!early) lhs_add(Jl, Ja, pen);
                        if (!isJoint) {
                            const f3 gyro = (KEEPJ ? keepG[k] : abs3(cross3(Ja, invIw * Ja))) * fabsf(f);        /* solver.cpp:393-397: Manifold rows only (latency form: the factor is hoisted out of the pass loop by the compiler; written out by hand it changes nothing) */
                            gy[0] += gyro.x; gy[1] += gyro.y; gy[2] += gyro.z;
                        }
                    }
                }
                /* the visit's mutations of the force (lambda cone clamp, stick; after a dual also lambda_n and the penalties); every record
                 * validates itself, so nothing is drained before the pose is published */
                const unsigned visit = (unsigned)(sole ? 0 : (first ? 2 * it + 1 : 2 * it + 2));
                const fv4u recL = fv4u{fu(c.lam[0]), fu(c.lam[1]), fu(c.lam[2]), (visit << 1) | (c.stick ? 1u : 0u)}, recP = fv4u{fu(c.pen[0]), fu(c.pen[1]), fu(c.pen[2]), (unsigned)(it + 1)};
                if (HOIST && X.local) { fst_local(X.rS, slot * 32, recL); if (first) fst_local(X.rS, slot * 32 + 16, recP); }
                else { fst(X.rS, slot * 32, recL); if (first) fst(X.rS, slot * 32 + 16, recP); }
            }
        }
        FAST_STAMPB(1);
        FAST_MARK(fv_sums);
        if (!SPLIT) {
#pragma unroll
            for (int i = 0; i < 21; ++i) acc[i] = group_sum<G>(acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) gy[i] = group_sum<G>(gy[i]);
#pragma unroll
        for (int i = 0; i < 6; ++i) rh[i] = group_sum<G>(rh[i]);
    }

    FAST_MARK(fv_solve);
    /* ---- solve6x6 (solver.cpp:68-83: five 3x3 LDL^T solves, the first four on one matrix), pose update (:402-408) ---- */
    rh[0] += rl0.x; rh[1] += rl0.y; rh[2] += rl0.z;
    rh[3] += ra0.x; rh[4] += ra0.y; rh[5] += ra0.z;
    const f3 bl = mk3(rh[0], rh[1], rh[2]), ba = mk3(rh[3], rh[4], rh[5]);
    f3 xl, xa;
    if (SPLIT) {
        const int q = lane & 3;
        const f3 x0 = fast_ldlt3_subst_quad(fL10, fL20, fL21, fd0, fd1, fd2, fbad, bl, q);
        m33 schur;                                                    /* (maths.h:104 reads the lower triangle only: c0, c1.y, c1.z, c2.z) */
        schur.c0 = mk3((aaDiag.x + gy[0]) - pDiag.x, schurOff.x, schurOff.y);
        schur.c1 = mk3(0.0f, (aaDiag.y + gy[1]) - pDiag.y, schurOff.z);
        schur.c2 = mk3(0.0f, 0.0f, (aaDiag.z + gy[2]) - pDiag.z);
        const f3 rhs_s = ba - al * x0;
        const f3 y = solve_ldlt3_quadlanes(schur, rhs_s, q);
        xl = x0 - mkm(col0, col1, col2) * y;
        xa = y;
    } else {
        acc[0] += mdt; acc[3] += mdt; acc[5] += mdt;
        acc[15] += aa0.c0.x; acc[16] += aa0.c0.y; acc[17] += aa0.c0.z; acc[18] += aa0.c1.y; acc[19] += aa0.c1.z; acc[20] += aa0.c2.z;
        acc[15] += gy[0]; acc[18] += gy[1]; acc[20] += gy[2];
        m33 ll, la, aa;
        ll.c0 = mk3(acc[0], acc[1], acc[2]); ll.c1 = mk3(acc[1], acc[3], acc[4]); ll.c2 = mk3(acc[2], acc[4], acc[5]);
        la.c0 = mk3(acc[6], acc[7], acc[8]); la.c1 = mk3(acc[9], acc[10], acc[11]); la.c2 = mk3(acc[12], acc[13], acc[14]);
        aa.c0 = mk3(acc[15], acc[16], acc[17]); aa.c1 = mk3(acc[16], acc[18], acc[19]); aa.c2 = mk3(acc[17], acc[19], acc[20]);
        const m33 alT = mtranspose(la);
        f3 c0s, c1s, c2s, x0, y;
        if (G >= 4) {
            /* every lane of the sub-group holds the whole system: lane q of a quad solves right-hand side q, the quad shares the four solutions */
            const int q = lane & 3;
            const f3 rhsq = mk3(pick4(la.c0.x, la.c1.x, la.c2.x, bl.x, q), pick4(la.c0.y, la.c1.y, la.c2.y, bl.y, q), pick4(la.c0.z, la.c1.z, la.c2.z, bl.z, q));
            const f3 sol = solve_ldlt3_branchless(ll, rhsq);
            c0s = quad_bcast3<0>(sol); c1s = quad_bcast3<1>(sol); c2s = quad_bcast3<2>(sol); x0 = quad_bcast3<3>(sol);
            m33 schur;                                                /* (maths.h:104 reads the lower triangle only: c0, c1.y, c1.z, c2.z) */
            schur.c0 = aa.c0 - alT * c0s;
            { const f3 p1 = alT * c1s, p2 = alT * c2s; schur.c1 = mk3(0.0f, aa.c1.y - p1.y, aa.c1.z - p1.z); schur.c2 = mk3(0.0f, 0.0f, aa.c2.z - p2.z); }
            const f3 rhs_s = ba - alT * x0;
            y = solve_ldlt3_quadlanes(schur, rhs_s, q);
        } else {
            c0s = solve_ldlt3_branchless(ll, la.c0); c1s = solve_ldlt3_branchless(ll, la.c1); c2s = solve_ldlt3_branchless(ll, la.c2);
            x0 = solve_ldlt3_branchless(ll, bl);
            const m33 AinvB = mkm(c0s, c1s, c2s);
            const m33 schur = aa - alT * AinvB;
            const f3 rhs_s = ba - alT * x0;
            y = solve_ldlt3_branchless(schur, rhs_s);
        }
        xl = x0 - mkm(c0s, c1s, c2s) * y;
        xa = y;
    }
    FAST_STAMPB(2);
    FAST_MARK(fv_pose);
    if (!prefetched && nextT >= 0 && nextT != t) fast_load_tile(F, nextT, next);
    f3 p = pos - xl;
    const q4 dq = mkq(xa.x, xa.y, xa.z, 0.0f);
    q4 q = qnormalize(rot - (dq * rot) * 0.5f);
    if (hasBody && r == 0) {
        sanitize3(p, X.cnt, warn_key(1, it, body, 0), body, WK_POS);
        sanitizeq(q, X.cnt, warn_key(1, it, body, 1), body, WK_ROT);
        fv4u r0, r1;
        pose_encode(p, q, (unsigned)(it + 1), r0, r1);
        if (HOIST && X.local) { fst_local(X.rPose, body * 32, r0); fst_local(X.rPose, body * 32 + 16, r1); }
        else { fst(X.rPose, body * 32, r0); fst(X.rPose, body * 32 + 16, r1); }
        FAST_STAMP(3);
        FAST_MARK(fv_published);
        if (lastOfLaunch) {                                           /* what the tail (and, sharded, the exchange) reads */
            const float4 old = X.B.plane(BP_POSR)[body];
            X.B.plane(BP_POSR)[body] = pack3(p, old.w);
            X.B.plane(BP_QUAT)[body] = packq(q);
        }
    }
}

/* After the last iteration: the dual update of that iteration (when the reference runs one: it < iterations) by the first
 * visitors, and the write-back of every contact's rows into the Manifold table. */
template <bool JOINTS>
__device__ __forceinline__ void fast_final(const FastCtx& X, const SolverParams& P, int t, int2 td, int totalIt)
{
    const int lane = threadIdx.x & 63;
    const FastBuffers& F = *X.F;
    const size_t SC = (size_t)F.slotCap, at = (size_t)t * 64 + lane;
    const int cls = (td.y >> 16) & 0xFF;
    if (cls == 0) return;
    const int G = fast_class_lanes(cls), j = lane / G, nb = td.y & 0xFFFF;
    const bool held = ((F.tileMask[t] >> lane) & 1ull) != 0;
    float4 c5 = make_float4(__int_as_float(-1), 0.f, 0.f, 0.f), c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0, c2 = c0, c3 = c0, c4 = c0;
    if (held) { c5 = F.slotC[5 * SC + at]; c0 = F.slotC[0 * SC + at]; c1 = F.slotC[1 * SC + at]; c2 = F.slotC[2 * SC + at]; c3 = F.slotC[3 * SC + at]; c4 = F.slotC[4 * SC + at]; }
    const int partner = __float_as_int(c5.x), slot = __float_as_int(c5.y), flags = __float_as_int(c5.z);
    const bool valid = (flags & FL_VALID) != 0 && j < nb, isA = (flags & FL_ISA) != 0, first = (flags & FL_FIRST) != 0, sole = (flags & FL_SOLE) != 0;
    const bool mine = valid && first;
    const int body = __float_as_int(F.bodyC[(size_t)3 * F.bodyCap + td.x + (j < nb ? j : 0)].w);
    f3 pos = zero3(), opos = zero3(); q4 rot = identq(), orot = identq();
    fv4u sl = fv4u{0u, 0u, 0u, 0u}, sp = sl;
    {
        const unsigned long long t0 = X.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
        for (;;) {
            bool ok = true;
            if (mine) {
                const fv4u o0 = fld(X.rPose, body * 32), o1 = fld(X.rPose, body * 32 + 16);
                ok = pose_decode(o0, o1, (unsigned)totalIt, pos, rot);
                sl = fld(X.rS, slot * 32); sp = fld(X.rS, slot * 32 + 16);
                if (!sole) {
                    const fv4u p0 = fld(X.rPose, partner * 32), p1 = fld(X.rPose, partner * 32 + 16);
                    ok &= pose_decode(p0, p1, (unsigned)totalIt, opos, orot);
                    ok &= (sl.w >> 1) == (unsigned)(2 * totalIt);
                }
            }
            if (!X.stamps || __all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > kFastTimeoutTicks) { if (!ok) atomicOr(&X.plan->bad, 256); break; }
        }
    }
    if (!mine) return;
    if (sole && !(JOINTS && (flags & FL_WORLD) != 0)) {
        const float4 pp = X.B.plane(BP_POSR)[partner], pq = X.B.plane(BP_QUAT)[partner];
        opos = mk3(pp.x, pp.y, pp.z); orot = mkq(pq.x, pq.y, pq.z, pq.w);
    }
    FastRows c;
    c.lam[0] = uf(sl.x); c.lam[1] = uf(sl.y); c.lam[2] = uf(sl.z); c.stick = (sl.w & 1u) != 0;
    c.pen[0] = uf(sp.x); c.pen[1] = uf(sp.y); c.pen[2] = uf(sp.z);
    const f3 rOwn = mk3(c0.x, c0.y, c0.z), rOth = mk3(c1.x, c1.y, c1.z);
    if (JOINTS && (flags & FL_JOINT) != 0) {
        /* a Joint half: the last dual (solver.cpp:411-430) and the rows back into the GForce record (C / fmin / fmax as the last computeConstraint leaves them, joint.cpp:68-106) */
        FastJointOut jo;
        if (flags & FL_SPRING) {
            fast_spring_rows(c, jo, flags, pos, rot, rOwn, opos, orot, rOth, c1.w, c0.w, c2.z, c2.x, c2.y, !P.postStabilize, X.beta);
            GForce& sf = X.gf[(slot - X.jointSlotBase) >> 1];
            sf.lambda[0] = c.lam[0]; sf.penalty[0] = c.pen[0]; sf.C[0] = jo.C[0];
            return;
        }
        const float stiff[3] = {c0.w, c1.w, c2.w}, motor[3] = {c3.y, c3.z, c3.w};
        const f3 wOwnJ = qrotate(rot, rOwn);
        fast_joint_rows(c, jo, flags, wOwnJ, pos + wOwnJ, rot, rOth, opos, orot, mkq(c2.x, c2.y, c2.z, c3.x), stiff, motor, !P.postStabilize, X.beta);
        const int idx = slot - X.jointSlotBase, part = idx & 1;
        GForce& jf = X.gf[idx >> 1];
        for (int k = 0; k < 3; ++k) {
            jf.lambda[3 * part + k] = c.lam[k]; jf.penalty[3 * part + k] = c.pen[k];
            jf.C[3 * part + k] = jo.C[k]; jf.fmin[3 * part + k] = -FLT_MAX; jf.fmax[3 * part + k] = FLT_MAX;
        }
        return;
    }
    const f3 n = mk3(c2.x, c2.y, c2.z), t1 = mk3(c3.x, c3.y, c3.z), t2 = mk3(c4.x, c4.y, c4.z);
    const f3 wOwn = qrotate(rot, rOwn), wOth = qrotate(orot, rOth);
    const f3 pOwn = pos + wOwn, pOth = opos + wOth;
    const f3 pA = isA ? pOwn : pOth, pB = isA ? pOth : pOwn;
    const float bs = rclamp(1.0f - fast_alpha(P, totalIt - 1), 0.0f, 1.0f);
    if (!P.postStabilize) {
        fast_dual(c, n, t1, t2, bs * c0.w, bs * c1.w, bs * c2.w, c4.w, pA, isA ? wOwn : wOth, pB, isA ? wOth : wOwn, c3.w, X.beta);
    } else {
        /* post-stabilisation: the extra sweep has no dual (solver.cpp:411); C / fmin / fmax / penetration as a computeConstraint on the final
         * poses leaves them (API visibility only: Manifold::initialize recomputes all four next step), lambda and stick as the last visit left them */
        FastRows probe = c;
        fast_constraint(probe, n, t1, t2, bs * c0.w, bs * c1.w, bs * c2.w, c4.w, pA, pB, c3.w);
        for (int k = 0; k < 3; ++k) { c.C[k] = probe.C[k]; c.fmin[k] = probe.fmin[k]; c.fmax[k] = probe.fmax[k]; }
        c.penetration = probe.penetration;
    }
    const ManifoldTable& M = X.M;
    M.cN[slot] = make_float4(n.x, n.y, n.z, c2.w);
    M.cU1[slot] = make_float4(t1.x, t1.y, t1.z, 0.f);
    M.cU2[slot] = make_float4(t2.x, t2.y, t2.z, 0.f);
    M.cLam[slot] = make_float4(c.lam[0], c.lam[1], c.lam[2], c.penetration);
    M.cPen[slot] = make_float4(c.pen[0], c.pen[1], c.pen[2], 0.f);
    M.cFS[slot].y = c.stick ? 1 : 0;
    M.cC[slot] = make_float4(c.C[0], c.C[1], c.C[2], 0.f);
    M.cFmin[slot] = make_float4(c.fmin[0], c.fmin[1], c.fmin[2], 0.f);
    M.cFmax[slot] = make_float4(c.fmax[0], c.fmax[1], c.fmax[2], 0.f);
}

template <int FORM, bool JOINTS, bool PRE>
__device__ __forceinline__ void fast_visit_any(const FastCtx& X, const SolverParams& P, const FastTileRegs& R, FastTileRegs& next, int nextT, int it, bool lastOfLaunch)
{
    constexpr bool TWOPASS = FORM != 0, HOIST = FORM == 1;
    const int cls = __builtin_amdgcn_readfirstlane((R.td.y >> 16) & 0xFF);
    switch (cls) {
    case 0: fast_visit<1, false, false, false, PRE>(X, P, R, next, nextT, it, lastOfLaunch); break;
    case 1: fast_visit<4, TWOPASS, HOIST, JOINTS, PRE>(X, P, R, next, nextT, it, lastOfLaunch); break;
    case 2: fast_visit<8, TWOPASS, HOIST, JOINTS, PRE>(X, P, R, next, nextT, it, lastOfLaunch); break;
    case 3: fast_visit<16, TWOPASS, HOIST, JOINTS, PRE>(X, P, R, next, nextT, it, lastOfLaunch); break;
    case 4: fast_visit<32, TWOPASS, HOIST, JOINTS, PRE>(X, P, R, next, nextT, it, lastOfLaunch); break;
    default: fast_visit<64, TWOPASS, HOIST, JOINTS, PRE>(X, P, R, next, nextT, it, lastOfLaunch); break;
    }
}

/* FORM 1: the latency form — workgroups of four waves, one wave per SIMD with the whole register file, contacts in two passes (what can be
 * evaluated before the last pose arrives is), everything that depends on the body's own pose alone kept across the wait; FORM 2: two passes, eight
 * waves per workgroup (two per SIMD), nothing kept across the wait; FORM 0: one pass, eight waves per workgroup. */
template <int FORM, bool JOINTS>
__global__ __launch_bounds__(FORM == 1 ? kSweepThreadsLat : (FORM == 0 && !JOINTS ? kSweepThreadsOne : kSweepThreads)) void k_fast_sweep(FastLaunch L)
{
    constexpr int kWaves = (FORM == 1 ? kSweepThreadsLat : (FORM == 0 && !JOINTS ? kSweepThreadsOne : kSweepThreads)) / 64;
    const FastBuffers& F = L.F;
    FastPlan* plan = F.plan;
    if (plan->bad != 0) return;                                      /* the plan declined the step: the exact schedule queued behind this launch runs it */
    const int T = plan->numTiles, numColors = plan->numColors;
    int nWG = gridDim.x;
    int wg = blockIdx.x;
    bool local = false;
    const int Treal = plan->colorTileStart[kFastIsolatedRow];        /* tiles [Treal, T): the bodies without a swept neighbour */
    int isoFirst = -1, isoStep = 0;                                  /* one-XCD form, a wave outside that XCD: its tiles among those */
    const int oneXcdTiles = L.oneXcdTiles & 0xFFFFF;                  /* (bit 30: debug, the roll call answers "several XCDs" whatever it finds) */
    if (FORM == 1 && L.stamps && oneXcdTiles > 0 && Treal <= oneXcdTiles && Treal <= (nWG >> 3) * kWaves && (nWG & 7) == 0 && (nWG >> 3) <= 64) {
        /* a small step — one tile per wave of an XCD at most — on ONE XCD: the workgroups with blockIdx % 8 == 0 (dealt to one XCD, round robin) call the roll — every one
         * leaves its HW_REG_XCC_ID, stamped with this launch, and reads everybody's — and hand records over with plain stores when all of them answer from the same XCD
         * (otherwise: sc1 stores as everywhere else).  The other seven eighths of the grid take the tiles of the bodies without a swept neighbour (a third of Stress1000's
         * cubes lie alone on the ground): nothing they write is read by anybody before the launch ends. */
        if ((wg & 7) != 0) {
            const int ow = ((wg >> 3) * 7 + (wg & 7) - 1) * kWaves + (threadIdx.x >> 6);
            isoFirst = Treal + ow; isoStep = (nWG - (nWG >> 3)) * kWaves;
        } else {
            wg >>= 3; nWG >>= 3;
            __shared__ int sLocal;
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                const unsigned tag = L.syncBase | ((unsigned)(L.itBegin + 1) << 4), mine = fast_xcc_id();
                unsigned* roll = F.sync + kFastXcdSlots;
                if (lane == 0) __hip_atomic_store(roll + wg, tag | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned v = tag | mine;
                for (;;) {
                    if (lane < nWG) v = __hip_atomic_load(roll + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all((v & ~15u) == tag)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > kFastTimeoutTicks) { if (lane == 0) atomicOr(&plan->bad, 256); v = tag | 15u; break; }
                }
                const bool same = __all((v & 15u) == mine) && ((L.oneXcdTiles >> 30) & 1) == 0;
                if (lane == 0) sLocal = same ? 1 : 0;
            }
            __syncthreads();
            local = sLocal != 0;
        }
    }
    const int W = nWG * kWaves;
    int gw = (threadIdx.x >> 6) * nWG + wg;                           /* consecutive tiles go to different workgroups */
    if (FORM == 1 && L.xcdLocal && !local && (nWG & 7) == 0) {
        /* latency form: workgroup b sits on XCD b % 8 (round robin, tools/ubench/xcd_pingpong.hip), and a hand-off inside one XCD costs 0.5 us where one across
         * XCDs costs 0.7: consecutive tiles fill the SIMDs of ONE XCD before the next (a small scene lives on one or two of them) */
        const int x = wg & 7, cu = wg >> 3, perX = nWG >> 3;
        gw = x * (perX * kWaves) + (threadIdx.x >> 6) * perX + cu;
    }
    FastCtx X;
    X.rPose = frsrc(F.xpose); X.rS = frsrc(F.xS);
    X.F = &F; X.plan = plan; X.cnt = L.cnt; X.B = L.B; X.M = L.M;
    X.invDt2 = 1.0f / (L.P.dt * L.P.dt); X.beta = L.P.beta; X.stamps = L.stamps; X.gf = F.gf; X.jointSlotBase = F.jointSlotBase; X.sleepSel = L.sleepSel; X.local = local;
    const SolverParams P = L.P;
    const int totalIt = P.iterations + (P.postStabilize ? 1 : 0);
    unsigned target = L.syncBase;
    if (T == 0) return;
    /* this wave's tiles: t = tFirst, tFirst + tStep, ... below tEnd */
    int tFirst = gw, tStep = W, tEnd = T;
    if (isoFirst >= 0) { tFirst = isoFirst; tStep = isoStep; }
    else if (local || (FORM == 1 && nWG != (int)gridDim.x)) tEnd = Treal;     /* (the XCD's own waves: the tiles with neighbours) */
    if (L.stamps && tFirst >= tEnd) return;                            /* nothing to do for this wave, nobody waits for it */

    if (L.stamps && FORM == 1) {
        /* stamped records: a wave's tiles in ascending order are in colour order (tiles are laid out colour by colour) — no phases, every visit waits for
         * its own inputs.  Latency form (one wave per SIMD, registers to spare): the constants of the next visit's tile are loaded behind the current visit's
         * last wait; a wave with one tile keeps them for the whole launch */
        /* (measured and not kept: the one-XCD form with TWO tiles per wave — Stress1000's 155 tiles on the 128 waves of an XCD — also with a wave's second tile chosen half
         * the colours away from its first: 274 / 294 us against 250 on all XCDs.  A wave starts the later tile's visit — own pose, first pass, pre-solve: 3.3 us of work —
         * only when the earlier one has published, and two colours further on is less than that away) */
        /* (also measured and not kept, one or two tiles per wave on all XCDs, 6 000 - 12 000 cubes: a wave's second tile chosen by the plan at least two colours away from its
         * first — the last colours' tiles off the waves of the first colour's — instead of tile + W: 8 000 cubes 431 -> 422 us, 10 648 cubes 489 -> 560) */
        FastTileRegs cur, nxt;
        fast_load_tile(F, tFirst, cur);
        for (int it = L.itBegin; it < L.itEnd; ++it) {
            for (int t = tFirst; t < tEnd; t += tStep) {
                const bool more = t + tStep < tEnd;
                const int nextT = more ? t + tStep : (it + 1 < L.itEnd ? tFirst : -1);
                nxt = cur;
                fast_visit_any<FORM, JOINTS, true>(X, P, cur, nxt, nextT, it, it == L.itEnd - 1);
                cur = nxt;
            }
        }
    } else if (L.stamps) {
        for (int it = L.itBegin; it < L.itEnd; ++it) {
            for (int t = gw; t < T; t += W) {
                FastTileRegs cur, nxt;
                cur.t = t; cur.td = F.tiles[t];
                fast_visit_any<FORM, JOINTS, false>(X, P, cur, nxt, -1, it, it == L.itEnd - 1);
            }
        }
    } else {
        for (int it = L.itBegin; it < L.itEnd; ++it) {
            for (int c = 0; c < max(1, numColors); ++c) {              /* (no colour at all: only bodies without a swept neighbour) */
                const int t0 = plan->colorTileStart[c], t1 = c < numColors ? plan->colorTileStart[c + 1] : t0;
                /* this wave's tiles of the colour: t = gw (mod W); the bodies without a swept neighbour go with the first colour */
                for (int part = 0; part < (c == 0 ? 2 : 1); ++part) {
                    const int p0 = part == 0 ? t0 : Treal, p1 = part == 0 ? t1 : T;
                    int t = p0 + ((gw - p0) % W + W) % W;
                    for (; t < p1; t += W) {
                        FastTileRegs cur, nxt;
                        cur.t = t; cur.td = F.tiles[t];
                        fast_visit_any<FORM, JOINTS, false>(X, P, cur, nxt, -1, it, it == L.itEnd - 1);
                    }
                }
                fast_grid_barrier<false>(F.sync, nWG, ++target, &plan->bad);
            }
        }
    }
    if (L.finalPass) {
        if (L.stamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int t = tFirst; t < tEnd; t += tStep) fast_final<JOINTS>(X, P, t, F.tiles[t], totalIt);
    }
}

#ifdef AVBD_DF_STATS
extern "C" int avbd_debug_fast_times(unsigned* out, int words)
{
    const int n = std::min(words, kFastStatTiles * kFastStatIts * 4);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gFastTimes), (size_t)n * sizeof(unsigned)) == hipSuccess ? n : -1;
}
#endif

/* =====================================================================================================================
 * host side
 * ===================================================================================================================== */
int fast_query(int device, int* planBlocksPerCU, int* sweepBlocksPerCU)
{
    (void)device;
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_fast_plan<false>, kPlanThreads, 0) != hipSuccess) a = 0;
    auto occ = [](const void* fn, int threads) { int v = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, fn, threads, 0) != hipSuccess) v = 0; return v; };
    b = std::min(std::min(occ((const void*)k_fast_sweep<0, false>, kSweepThreadsOne), occ((const void*)k_fast_sweep<2, false>, kSweepThreads)),
                 std::min(occ((const void*)k_fast_sweep<0, true>, kSweepThreads), occ((const void*)k_fast_sweep<2, true>, kSweepThreads)));
    if (occ((const void*)k_fast_sweep<1, false>, kSweepThreadsLat) < 1 || occ((const void*)k_fast_sweep<1, true>, kSweepThreadsLat) < 1) b = 0;
    if (planBlocksPerCU) *planBlocksPerCU = a;
    if (sweepBlocksPerCU) *sweepBlocksPerCU = b;
    return (a > 0 && b > 0) ? 1 : 0;
}

hipError_t fast_launch_plan(hipStream_t stream, const FastLaunch& L, int planBlocksPerCU)
{
    const int maxWG = std::max(1, std::min(1024, L.numCU * std::max(1, planBlocksPerCU)));
    if (L.B.n <= kPlanSingleBodies) hipLaunchKernelGGL(k_fast_plan<true>, dim3(1), dim3(kPlanThreads), 0, stream, L);
    else {
        const int nWG = std::max(2, std::min(maxWG, (L.B.n + kPlanThreads - 1) / kPlanThreads));
        hipLaunchKernelGGL(k_fast_plan<false>, dim3(nWG), dim3(kPlanThreads), 0, stream, L);
    }
    /* (the tile count is known on the device only: a grid for the bodies' worst case would be mostly idle, so a grid-stride loop over what the plan holds) */
    const int fillWG = std::max(1, std::min(4 * L.numCU, (int)(((long long)L.B.n * 12 + 255) / 256)));
    hipLaunchKernelGGL(k_fast_fill, dim3(fillWG), dim3(256), 0, stream, L);
    return hipGetLastError();
}

hipError_t fast_launch_sweep(hipStream_t stream, const FastLaunch& Lin, int sweepBlocksPerCU, int tilesHint, int forceBlocks, int form)
{
    FastLaunch L = Lin;
    /* `form`: 0 by size, 1 latency (one wave per SIMD), 2 two waves per SIMD in two passes, 3 one pass */
    const long long tilesGuess = tilesHint > 0 ? tilesHint : ((long long)L.B.n * 3) / 16;   /* (a settled pile of cubes: 0.17 tiles per body) */
    /* measured (ms per step, forms 1 / 2 / 3; tiles; final build): 8 000 cubes 0.596 / 0.615 / 0.724 (1 377), 10 648 cubes 0.652 / 0.649 / 0.742 (1 800), 13 824 cubes
     * 0.748 / 0.712 / 0.840 (2 333), 21 952 cubes 0.921 / 0.745 / 0.923 (3 700), 32 768 cubes 1.274 / 0.922 / 1.026 (5 596), 39 304 cubes - / 0.971 / 1.051 (6 700), 46 656 cubes
     * - / 1.117 / 1.114 (8 000), 64 000 cubes - / 1.454 / 1.289 (11 174): the latency form up to 1.75 tiles per wave, one pass from 7.75 on */
    int f = form != 0 ? form : (tilesGuess * 4 <= (long long)7 * 4 * L.numCU ? 1 : (tilesGuess * 4 <= (long long)31 * 4 * L.numCU ? 2 : 3));
    if (L.joints != 0 && f == 2) f = 3;                                /* (the joint-aware two-pass kernel at two waves per SIMD spills: 144 B per lane) */
    const bool lat = f == 1;
    if (L.sleepSel < 0) L.sleepSel = f == 3 ? 2 : 0;                   /* (64 000 cubes, one pass: s_sleep 16 between two looks 1 201 -> 1 160 us; small scenes: s_sleep 1, 272 against 298) */
    const int waves = (lat ? kSweepThreadsLat : (f == 3 && L.joints == 0 ? kSweepThreadsOne : kSweepThreads)) / 64;
    const int maxWG = lat ? L.numCU : std::max(1, std::min(1024, L.numCU * std::max(1, sweepBlocksPerCU)));
    int nWG = maxWG;
    if (!L.stamps && tilesHint > 0) nWG = (int)(((long long)tilesHint * 5 / 4 + waves - 1) / waves) + 1;   /* (grid barrier: no more workgroups than the tiles need) */
    if (forceBlocks > 0) nWG = forceBlocks;
    nWG = std::max(1, std::min(maxWG, nWG));
    const bool joints = L.joints != 0;                                 /* (the joint-aware code is a kernel of its own: the contacts-only kernels keep their registers) */
    if (lat) { if (joints) hipLaunchKernelGGL((k_fast_sweep<1, true>), dim3(nWG), dim3(kSweepThreadsLat), 0, stream, L); else hipLaunchKernelGGL((k_fast_sweep<1, false>), dim3(nWG), dim3(kSweepThreadsLat), 0, stream, L); }
    else if (f == 2) { if (joints) hipLaunchKernelGGL((k_fast_sweep<2, true>), dim3(nWG), dim3(kSweepThreads), 0, stream, L); else hipLaunchKernelGGL((k_fast_sweep<2, false>), dim3(nWG), dim3(kSweepThreads), 0, stream, L); }
    else { if (joints) hipLaunchKernelGGL((k_fast_sweep<0, true>), dim3(nWG), dim3(kSweepThreads), 0, stream, L); else hipLaunchKernelGGL((k_fast_sweep<0, false>), dim3(nWG), dim3(kSweepThreadsOne), 0, stream, L); }
    return hipGetLastError();
}

} // namespace avbd
