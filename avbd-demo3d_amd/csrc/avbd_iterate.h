/*
 * avbd_iterate.h — the whole iteration loop of Solver::step() (solver.cpp:340-431) as ONE persistent
 * data-flow launch.
 *
 * The reference runs `iterations` x (sequential primal sweep over bodies, then dual over forces).  The
 * result of that schedule is reproduced exactly by any execution that respects these dependencies
 * (n = a dynamic neighbour of body b through a non-marker force; "later" = created later = earlier in
 * the reference's body list):
 *
 *   primal(b, it)  after  primal(n, it)      for every later neighbour n   (it reads n's new pose)
 *   primal(b, it)  after  dual(f, it-1)      for every incident force f    (it reads lambda / penalty)
 *   dual(f, it)    after  primal of both endpoints at `it`
 *
 * dual(f, it) is executed by the wave that runs the LAST primal visit of f in the sweep (the earlier-created
 * dynamic endpoint, "owner"), fused behind that body's pose update.  Task (b, it) is therefore
 *   wait( primalDone[n] >= it+1  for later neighbours,  dualDone[n] >= it  for earlier neighbours and for b itself )
 *   primal(b, it);  publish primalDone[b] = it+1;  dual of owned forces;  publish dualDone[b] = it+1
 * Tasks are numbered t = it * Nd + slot over the level-sorted body order (a linear extension of the
 * dependency order) and wave w of the launch runs tasks w, w+W, w+2W, ... in that order; with every wave
 * resident the smallest unfinished task is always at the head of its wave's list, so the schedule cannot
 * deadlock.  Iterations pipeline through the dependency DAG instead of meeting at grid-wide barriers.
 *
 * Hand-offs between workgroups use the placement-independent protocol of cdna_hip_programming.md
 * Guideline 16 / MI355X_MICROARCH.md "Valid forms": every shared byte is stored `sc1` (relaxed agent-scope
 * atomics), the storing wave drains with s_waitcnt vmcnt(0), then one lane publishes the counter with an
 * `sc1` store; consumers poll the counter with relaxed `sc1` loads and read the payload with `sc1` loads
 * only.  Every spin is bounded by wall-clock time and raises an error word instead of hanging.
 */
#pragma once
#include "avbd_primal.h"

#pragma clang fp contract(off)

namespace avbd {

struct IterateState {
    int* primalDone;   /* [nBodies] iterations whose primal visit of the body is published */
    int* dualDone;     /* [nBodies] iterations whose owned dual updates are published */
    int* error;        /* 0 ok, 1 spin timeout */
};

constexpr int kIterWaves = 4;                      /* waves per workgroup */
constexpr unsigned long long kSpinTimeoutTicks = 400000000ull; /* 4 s of s_memrealtime (100 MHz) */

} // namespace avbd
#include "avbd_resident.h"
namespace avbd {

__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

/* is this incidence entry's force owned by `body` (= body is its earliest-created dynamic endpoint)? */
__device__ __forceinline__ bool entry_owned(int ent, int body, const BodyTable& B, const ManifoldTable& M, const GForce* gf, int& other)
{
    const int kind = inc_kind(ent), idx = inc_index(ent);
    const bool isA = inc_isA(ent) != 0;
    if (kind == FK_MANIFOLD) { const int2 pr = M.pair[idx >> 2]; other = isA ? pr.y : pr.x; }
    else if (kind == FK_JOINT) { const GForce& f = gf[idx >> 1]; other = isA ? f.bodyB : f.bodyA; }
    else { const GForce& f = gf[idx & 0xFFFFFF]; other = isA ? f.bodyB : f.bodyA; }
    if (other < 0) return true;                                   /* world */
    if (!B.solved[other]) return true;                            /* static partner, or a body another GPU solves */
    return other > body;                                           /* partner is visited earlier in the sweep */
}

#ifdef AVBD_DF_STATS
#define DF_STAMP(i) do { unsigned long long _n = __builtin_amdgcn_s_memrealtime(); stat[i] += _n - tprev; tprev = _n; } while (0)
#else
#define DF_STAMP(i) do { } while (0)
#endif

/* dependency wait of task (body, it); returns false when the launch must be abandoned */
__device__ __forceinline__ bool wait_dependencies(int body, int it, int lane, const int* adjStart, const int* adjRef, const int* adjOther,
                                                  const int* solved, const IterateState& S, int sleepSel, int which /* 0: all, 1: own previous task only, 2: neighbours only */)
{
    const int aBeg = adjStart[body], aEnd = adjStart[body + 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    /* lane 63 of the first round also waits for this body's own previous task (pose + owned duals) */
    const int aLast = which == 1 ? aBeg : aEnd;
    for (int base = aBeg; base < aLast || base == aBeg; base += 63) {
        const int k = base + lane;
        int need = 0; const int* word = nullptr;
        if (lane == 63) {
            if (base == aBeg && it > 0 && which != 2) { word = S.dualDone + body; need = it; }
        } else if (k < aEnd && which != 1) {
            const int o = adjOther[k];
            if (o >= 0 && inc_kind(adjRef[k]) != FK_IGNORE && solved[o]) {
                if (o > body) { word = S.primalDone + o; need = it + 1; }
                else if (it > 0) { word = S.dualDone + o; need = it; }
            }
        }
        bool ok = (word == nullptr);
        unsigned spins = 0;
        while (true) {
            if (!ok) ok = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
            if (__all(ok)) break;
            if (sleepSel == 0) __builtin_amdgcn_s_sleep(1); else if (sleepSel == 1) __builtin_amdgcn_s_sleep(2);
            else if (sleepSel == 2) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(8);
            if ((++spins & 1023u) == 0) {
                bool dead = (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTimeoutTicks) ||
                            __hip_atomic_load(S.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (__any(dead)) { if (lane == 0) __hip_atomic_store(S.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); /* keep the payload loads below the polls */
    return true;
}

/* One wavefront per body for the whole loop and nothing else in the kernel: the launch of scenes whose swept bodies all find a
 * resident wave (Stress1000).  Same code as the resident branch of k_iterate<1> below, compiled on its own so that the register
 * allocation of the loop that bounds the headline workload does not carry the other schedules' state.  Declines (returns at once)
 * when the topology kernels raised any veto bit; the k_iterate launch queued behind it with runIf = &cnt->residentBad then does
 * the step. */
#ifndef AVBD_TWOSTAGE
#define AVBD_TWOSTAGE 0             /* 1: solve_pose_pre during the wait, solve_pose_post behind it (avbd_primal.h).  Built, bit-identical (golden traces), and
                                     * not faster: 85 instructions leave the hop but only the 24 of two division levels were on its dependent chain — 2.92 -> 2.86 us from
                                     * detect to publish in the stamped build, 0.583 -> 0.586 ms per launch in the product (44 AGPRs instead of 4 carry the rest) */
#endif
template <bool JOINTS>
__global__ __launch_bounds__(64 * kIterWaves) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_iterate_single(BodyTable B, ManifoldTable M, GForce* gf, const int* __restrict__ order,
                                                              const int* __restrict__ lvlStart, const Counters* __restrict__ cnt, const int* __restrict__ incStart,
                                                              const int* __restrict__ inc, SolverParams P, int* errWord, Counters* cntW, int sleepSel, int itEnd, ResidentExchange X)
{
    if (cnt->residentBad != 0) return;
    __shared__ float sRows[kIterWaves][66 * kEntryRec];
    __shared__ float sAcc[kIterWaves][96];                           /* [64..88]: what stage one of the solve leaves for stage two (solve_pose_pre) */
    __shared__ int sBodyOf[kIterWaves];
    __shared__ v4u sPoseRec[kIterWaves][3];
    __shared__ unsigned sPoseFlag[kIterWaves];
    extern __shared__ float4 sDynamic[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = gridDim.x * kIterWaves;
    const int w = blockIdx.x * kIterWaves + wv;
    const int nd = lvlStart[cnt->numLevels];
    float* products = reinterpret_cast<float*>(sDynamic) + wv * 27 * X.prStride;
    const int slot = X.spread ? wv * (int)gridDim.x + (int)blockIdx.x : w;
    /* the body of this wave: k_place_chains' table (chains of hand-offs share a workgroup) or the slot of the sweep order */
    int mine = -1;
    if (X.place) mine = __builtin_amdgcn_readfirstlane(X.place[w]);
    else if (slot < nd) mine = __builtin_amdgcn_readfirstlane(order[slot]);
    if (lane == 0) { sBodyOf[wv] = mine; sPoseFlag[wv] = X.base; }
    __syncthreads();
    CoLocal CL{(const volatile int AVBD_LDS*)sBodyOf, (volatile v4u AVBD_LDS*)&sPoseRec[0][0], (volatile unsigned AVBD_LDS*)sPoseFlag, wv};
    if (mine >= 0) iterate_resident<false, JOINTS, true, !JOINTS && AVBD_TWOSTAGE>(B, M, gf, order, nd, slot, W, incStart, inc, P, errWord, cntW, X, sleepSel, lane, sRows[wv], sAcc[wv], products, 0, itEnd, CL, mine);
}

/* WPE = wavefronts per SIMD the launch is sized for: 1 (latency mode, one workgroup per CU: the critical wave never shares
 * its SIMD with a spinning one) or 2 (throughput mode for scenes with thousands of tasks per wave). */
template <int WPE, bool JOINTS>
__global__ __launch_bounds__(64 * kIterWaves) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_iterate(BodyTable B, ManifoldTable M, GForce* gf, const int* __restrict__ order,
                                                              const int* __restrict__ lvlStart, const Counters* __restrict__ cnt,
                                                              const int* __restrict__ adjStart, const int* __restrict__ adjRef,
                                                              const int* __restrict__ adjOther, const int* __restrict__ incStart,
                                                              const int* __restrict__ inc, SolverParams P, IterateState S, Counters* cntW, int sleepSel, int itBegin, int itEnd,
                                                              ResidentExchange X, const int* __restrict__ runIf)
{
    if (runIf && *runIf == 0) return;                              /* queued behind k_iterate_groups: runs only when that launch declined the step */
    __shared__ float sRows[kIterWaves][66 * kEntryRec];
    __shared__ float sAcc[kIterWaves][64];
    extern __shared__ float4 sDynamic[];   /* latency mode: pins one workgroup per CU; the resident schedule keeps its product lines here */
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   /* wave-uniform by construction: tell the compiler */
    float* rows = sRows[wv];
    float* accs = sAcc[wv];
    const int W = gridDim.x * kIterWaves;
    const int w = blockIdx.x * kIterWaves + wv;
    const int nd = lvlStart[cnt->numLevels];                       /* dynamic bodies in `order` */
    const int totalIt = P.iterations + (P.postStabilize ? 1 : 0);
    /* [itBegin, itEnd) selects the iterations of this launch: the whole loop on one GPU, one iteration per launch
     * when body state is exchanged between GPUs after every iteration (the done-counters persist in between) */
    const long long nTasks = (long long)nd * (itEnd < totalIt ? itEnd : totalIt);
    const float4* massPlane = B.plane(BP_MASS);
    {
        /* a plain pile of contacts: the resident schedule (avbd_resident.h).  veto bits set by the topology kernels —
         * bit0: a swept body has > 64 contacts or a joint/spring, bit1: more swept bodies than resident waves
         * (bit2, a swept body with > 33 contacts, is informational: the two-workgroups-per-CU launch multiplies the rows
         * that do not fit its short product lines after the wait) */
        const int veto = cnt->residentBad;
        const bool lite = WPE == 2;
        if (X.enabled && (veto & 1) == 0 && ((veto & 2) == 0 || X.multi)) {
            float* products = reinterpret_cast<float*>(sDynamic) + wv * 27 * X.prStride;
            const bool multi = lite || (veto & 2) || X.forceMulti;
            /* one body per wave: the four waves of a CU take slots a quarter of the sweep order apart, so that they are
             * busy at different times (waves of one CU share its LDS, L1 and instruction front end) */
            const int slot = X.spread ? wv * (int)gridDim.x + (int)blockIdx.x : w;
            /* JOINTS is a property of the launch (the host instantiates the joint-aware resident code only for scenes that have
             * joints: it costs registers on the path that bounds the contacts-only headline workload) */
            if (multi) iterate_resident<true, JOINTS>(B, M, gf, order, nd, w, W, incStart, inc, P, S.error, cntW, X, sleepSel, lane, rows, accs, products, itBegin, itEnd);
            else if (slot < nd) {
                if constexpr (WPE == 1) iterate_resident<false, JOINTS>(B, M, gf, order, nd, slot, W, incStart, inc, P, S.error, cntW, X, sleepSel, lane, rows, accs, products, itBegin, itEnd);
            }
            return;
        }
    }

#ifdef AVBD_DF_STATS
    unsigned long long stat[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tstart = tprev;
    const unsigned long long cstart = __builtin_amdgcn_s_memtime();
#endif
    for (long long t = (long long)nd * itBegin + w; t < nTasks; t += W) {
        DF_STAMP(7);
        const int it = (int)(t / nd);
        const int slot = (int)(t - (long long)it * nd);
        const int body = order[slot];
        const float alpha = P.postStabilize ? (it < P.iterations ? 1.0f : 0.0f) : P.alpha;
        const bool writeRows = it >= P.iterations;
        const bool doDual = it < P.iterations;
        const bool lastDual = it == P.iterations - 1;              /* rows are API-visible only after the step */
        const int eBeg = incStart[body], eEnd = incStart[body + 1];
        const int nEnt = eEnd - eBeg;

        /* ---- everything that is constant during the launch is fetched BEFORE the wait ---- */
        const bool has = lane < nEnt;
        const int ent = has ? inc[eBeg + lane] : 0;
        const bool fast = nEnt <= 64 && __all(!has || inc_kind(ent) == FK_MANIFOLD);
        int idx = 0, other = 0; bool isA = false, owned = false;
        float oinv = 0.0f, fric = 0.0f;
        float4 sra = make_float4(0.f, 0.f, 0.f, 0.f), srb = sra;
        if (fast && has) {
            idx = inc_index(ent); isA = inc_isA(ent) != 0;
            const int m = idx >> 2;
            const int2 pr = M.pair[m];
            other = isA ? pr.y : pr.x;
            oinv = massPlane[other].y;
            fric = M.fric[m];
            sra = M.cRA[idx]; srb = M.cRB[idx];
            owned = !B.solved[other] || other > body;
        }
        const float4 snp = B.plane(BP_NPOS)[body], snq = B.plane(BP_NQUAT)[body], smp = massPlane[body];
        const float4 i0 = B.plane(BP_I0)[body], i1 = B.plane(BP_I1)[body], i2 = B.plane(BP_I2)[body];
        const float4 j0 = B.plane(BP_J0)[body], j1 = B.plane(BP_J1)[body], j2 = B.plane(BP_J2)[body];

        BodyLocal L;
        AccLane a;
        ContactRows c;
        f3 wOwn = zero3(), wOther = zero3(), pOther = zero3(), opos = zero3(); q4 orot = identq();
        bool otherFixed = false;
        if (fast) {
            /* ---- stage 1: needs only this body's previous task: own pose, world inertia, accumulator seeds, own lever arms ---- */
            if (!wait_dependencies(body, it, lane, adjStart, adjRef, adjOther, B.solved, S, sleepSel, 1)) return;
            const float4 pp = ld4<true>(B.plane(BP_POSR) + body), pq = ld4<true>(B.plane(BP_QUAT) + body);
            L.pos = mk3(pp.x, pp.y, pp.z); L.rot = mkq(pq.x, pq.y, pq.z, pq.w);
            L.npos = mk3(snp.x, snp.y, snp.z); L.nrot = mkq(snq.x, snq.y, snq.z, snq.w);
            L.mass = smp.x; L.invMass = smp.y;
            {
                const m33 Ib = mkm(mk3(i0.x, i0.y, i0.z), mk3(i1.x, i1.y, i1.z), mk3(i2.x, i2.y, i2.z));
                const m33 Jb = mkm(mk3(j0.x, j0.y, j0.z), mk3(j1.x, j1.y, j1.z), mk3(j2.x, j2.y, j2.z));
                world_inertia_parallel(lane, L.rot, Ib, Jb, accs + 32, L.Iw, L.invIw);   /* rigid.cpp:51-59 */
                L.invDt2 = 1.0f / (P.dt * P.dt);
            }
            a = init_acc_lane(lane, L);
            if (has) {
                c.rA = mk3(sra.x, sra.y, sra.z); c.C0n = sra.w;
                c.rB = mk3(srb.x, srb.y, srb.z); c.C0t1 = srb.w;
                wOwn = qrotate(L.rot, isA ? c.rA : c.rB);                     /* manifold.cpp:27 and :265 use the same product */
                otherFixed = !B.solved[other];                                 /* static, or swept by another GPU: constant in this launch */
                if (otherFixed) {
                    const float4 op = B.plane(BP_POSR)[other], oq = B.plane(BP_QUAT)[other];
                    opos = mk3(op.x, op.y, op.z); orot = mkq(oq.x, oq.y, oq.z, oq.w);
                    wOther = qrotate(orot, isA ? c.rB : c.rA);
                    pOther = opos + wOther;
                }
            }
        }

        /* ---- stage 2: the neighbours ---- */
        if (!wait_dependencies(body, it, lane, adjStart, adjRef, adjOther, B.solved, S, sleepSel, fast ? 2 : 0)) return;
        DF_STAMP(0);

        if (fast) {
            /* one stage of sc1 loads: partner pose (if it moves in this launch) and the mutable contact fields */
            float4 nn = sra, lam = sra, pen = sra, mu1 = sra, mu2 = sra; int stick = 0;
            if (has) {
                float4 op = sra, oq = sra;
                if (!otherFixed) { op = ld4<true>(B.plane(BP_POSR) + other); oq = ld4<true>(B.plane(BP_QUAT) + other); }
                nn = ld4<true>(M.cN + idx); lam = ld4<true>(M.cLam + idx); pen = ld4<true>(M.cPen + idx);
                mu1 = ld4<true>(M.cU1 + idx); mu2 = ld4<true>(M.cU2 + idx);
                stick = ldi<true>(&M.cFS[idx].y);
                if (!otherFixed) {
                    opos = mk3(op.x, op.y, op.z); orot = mkq(oq.x, oq.y, oq.z, oq.w);
                    wOther = qrotate(orot, isA ? c.rB : c.rA);
                    pOther = opos + wOther;
                }
            }
#ifdef AVBD_DF_STATS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DF_STAMP(8);
#endif
            DF_STAMP(9);
            bool rowsOn = true;
            if (has) {
                c.normal = mk3(nn.x, nn.y, nn.z); c.C0t2 = nn.w;
                c.u1 = mk3(mu1.x, mu1.y, mu1.z); c.u2 = mk3(mu2.x, mu2.y, mu2.z);
                c.lambda[0] = lam.x; c.lambda[1] = lam.y; c.lambda[2] = lam.z; c.penetration = lam.w;
                c.penalty[0] = pen.x; c.penalty[1] = pen.y; c.penalty[2] = pen.z;
                c.stick = stick != 0;
                rowsOn = eval_contact_pts(c, isA, L, wOwn, pOther, oinv, fric, alpha, rows + lane * kEntryRec);
                /* a contact this wave is about to update again in its dual pass needs no intermediate store unless the
                 * partner reads it first: the partner of an owned contact has already finished this iteration */
                if (!(owned && doDual)) store_contact_mutations<true>(M, idx, c, writeRows);
            } else if (lane < nEnt + 2) {
                clear_row_flags(rows + lane * kEntryRec);
            }
            const bool allOn = __all(rowsOn);
            wave_sync();
            DF_STAMP(10);
            if (allOn) accumulate_rows_all_on(a, rows, nEnt); else accumulate_rows(a, rows, nEnt);
            if (lane < 27) accs[lane] = a.acc;
            wave_sync();
            DF_STAMP(1);
            f3 p; q4 q;
            solve_pose(accs, accs + 32, lane, L, p, q);
            if (lane == 0) {
                sanitize3(p, cntW, warn_key(1, it, body, 0), body, WK_POS);
                sanitizeq(q, cntW, warn_key(1, it, body, 1), body, WK_ROT);
                st4<true>(B.plane(BP_POSR) + body, pack3(p, smp.w));
                st4<true>(B.plane(BP_QUAT) + body, packq(q));
                if (X.send) { const int k = X.sendSlot[body]; X.send[2 * k] = pack3(p, smp.w); X.send[2 * k + 1] = packq(q); }   /* this rank's all-gather rows */
                accs[56] = p.x; accs[57] = p.y; accs[58] = p.z; accs[59] = q.x; accs[60] = q.y; accs[61] = q.z; accs[62] = q.w;
            }
            DF_STAMP(2);
            drain_stores();                                            /* every lane's sc1 stores have left the wave */
            if (lane == 0) __hip_atomic_store(S.primalDone + body, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            DF_STAMP(3);
            if (doDual) {
                wave_sync();
                if (has && owned) {
                    const f3 np = mk3(accs[56], accs[57], accs[58]); const q4 nq = mkq(accs[59], accs[60], accs[61], accs[62]);
                    const f3 wNew = qrotate(nq, isA ? c.rA : c.rB);
                    const f3 pNew = np + wNew;
                    dual_contact_pts(c, isA ? pNew : pOther, isA ? wNew : wOther, isA ? L.invMass : oinv, isA ? pOther : pNew, isA ? wOther : wNew,
                                     isA ? oinv : L.invMass, fric, P, alpha);
                    store_contact_dual<true>(M, idx, c, lastDual);
                }
                drain_stores();
            }
            if (lane == 0) __hip_atomic_store(S.dualDone + body, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            DF_STAMP(4);
        } else {
            /* ---- generic path: any entry kinds, any count (chunks of 64), state re-read per phase ---- */
            const BodyLocal L = load_body_local<true>(B, body, P);
            AccLane a = init_acc_lane(lane, L);
            for (int base = eBeg; base < eEnd; base += 64) {
                const int nHere = min(64, eEnd - base);
                if (lane < nHere) eval_entry<true>(inc[base + lane], B, M, gf, L, alpha, writeRows, rows + lane * kEntryRec);
                else if (lane < nHere + 2) clear_row_flags(rows + lane * kEntryRec);
                wave_sync();
                accumulate_rows(a, rows, nHere);
                wave_sync();
            }
            if (lane < 27) accs[lane] = a.acc;
            wave_sync();
            DF_STAMP(1);
            f3 p; q4 q;
            solve_pose(accs, accs + 32, lane, L, p, q);
            if (lane == 0) {
                sanitize3(p, cntW, warn_key(1, it, body, 0), body, WK_POS);
                sanitizeq(q, cntW, warn_key(1, it, body, 1), body, WK_ROT);
                st4<true>(B.plane(BP_POSR) + body, pack3(p, smp.w));
                st4<true>(B.plane(BP_QUAT) + body, packq(q));
                if (X.send) { const int k = X.sendSlot[body]; X.send[2 * k] = pack3(p, smp.w); X.send[2 * k + 1] = packq(q); }   /* this rank's all-gather rows */
            }
            DF_STAMP(2);
            drain_stores();
            if (lane == 0) __hip_atomic_store(S.primalDone + body, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            DF_STAMP(3);
            if (doDual) {
                for (int base = eBeg; base < eEnd; base += 64) {
                    const int e = base + lane;
                    if (e < eEnd) {
                        const int en = inc[e];
                        int oth;
                        if (entry_owned(en, body, B, M, gf, oth)) {
                            const int kind = inc_kind(en), ix = inc_index(en);
                            if (kind == FK_MANIFOLD) dual_contact<true>(B, M, ix, P, alpha, lastDual);
                            else if (kind == FK_JOINT) { if ((ix & 1) == 0) dual_gforce<true>(B, gf[ix >> 1], P, lastDual); }
                            else dual_gforce<true>(B, gf[ix], P, lastDual);
                        }
                    }
                }
                drain_stores();
            }
            if (lane == 0) __hip_atomic_store(S.dualDone + body, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            DF_STAMP(4);
        }
#ifdef AVBD_DF_STATS
        stat[5] += 1;
#endif
    }
#ifdef AVBD_DF_STATS
    if (lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(S.error + 16) + (size_t)w * 16;
        stat[6] = __builtin_amdgcn_s_memrealtime() - tstart;
        stat[11] = __builtin_amdgcn_s_memtime() - cstart;
        for (int i = 0; i < 16; ++i) o[i] = stat[i];
    }
#endif
}

#ifndef AVBD_TU_SINGLE
/* Forces with no dynamic endpoint (static-static manifolds, joints between statics) take no part in any
 * primal visit; their dual updates (which only feed the diagnostics and the API-visible rows) run here. */
__global__ void k_dual_static(BodyTable B, ManifoldTable M, const Counters* cnt, GForce* gf, int ng, SolverParams P)
{
    const int nm = cnt->numManifolds;
    const int total = nm * 4 + ng;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < nm * 4) {
            const int m = i >> 2;
            if ((i & 3) >= M.nc[m]) continue;
            const int2 pr = M.pair[m];
            if (B.solved[pr.x] || B.solved[pr.y]) continue;
            for (int it = 0; it < P.iterations; ++it) dual_contact<false>(B, M, i, P, P.postStabilize ? 1.0f : P.alpha, true);
        } else {
            GForce& f = gf[i - nm * 4];
            bool dynA = f.bodyA >= 0 && B.solved[f.bodyA], dynB = f.bodyB >= 0 && B.solved[f.bodyB];
            if (dynA || dynB) continue;
            for (int it = 0; it < P.iterations; ++it) dual_gforce<false>(B, f, P, true);
        }
    }
}

#endif

} // namespace avbd
