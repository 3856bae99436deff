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

__device__ __forceinline__ void drain_stores()
{
#ifdef AVBD_DF_FENCES
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

/* is this incidence entry's force owned by `body` (= body is its earliest-created dynamic endpoint)? */
__device__ __forceinline__ bool entry_owned(int ent, int body, const BodyTable& B, const ManifoldTable& M, const GForce* gf, int& other)
{
    const int kind = inc_kind(ent), idx = inc_index(ent);
    const bool isA = inc_isA(ent) != 0;
    if (kind == FK_MANIFOLD) { const int2 pr = M.pair[idx >> 2]; other = isA ? pr.y : pr.x; }
    else if (kind == FK_JOINT) { const GForce& f = gf[idx >> 1]; other = isA ? f.bodyB : f.bodyA; }
    else { const GForce& f = gf[idx]; other = isA ? f.bodyB : f.bodyA; }
    if (other < 0) return true;                                   /* world */
    if (!(B.plane(BP_MASS)[other].y > 0.0f)) return true;         /* static partner */
    return other > body;                                           /* partner is visited earlier in the sweep */
}

__global__ __launch_bounds__(64 * kIterWaves) void k_iterate(BodyTable B, ManifoldTable M, GForce* gf, const int* __restrict__ order,
                                                              const int* __restrict__ lvlStart, const Counters* __restrict__ cnt,
                                                              const int* __restrict__ adjStart, const int* __restrict__ adjRef,
                                                              const int* __restrict__ adjOther, const int* __restrict__ incStart,
                                                              const int* __restrict__ inc, SolverParams P, IterateState S, Counters* cntW)
{
    __shared__ float sRows[kIterWaves][64 * kEntryRec];
    __shared__ float sAcc[kIterWaves][32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* rows = sRows[wv];
    float* accs = sAcc[wv];
    const int W = gridDim.x * kIterWaves;
    const int w = blockIdx.x * kIterWaves + wv;
    const int nd = lvlStart[cnt->numLevels];                       /* dynamic bodies in `order` */
    const int totalIt = P.iterations + (P.postStabilize ? 1 : 0);
    const long long nTasks = (long long)nd * totalIt;
    const float4* massPlane = B.plane(BP_MASS);

    for (long long t = w; t < nTasks; t += W) {
        const int it = (int)(t / nd);
        const int slot = (int)(t - (long long)it * nd);
        const int body = order[slot];
        const float alpha = P.postStabilize ? (it < P.iterations ? 1.0f : 0.0f) : P.alpha;
        const bool writeRows = it >= P.iterations;

        /* ---- wait for the dependencies of task (body, it) ---- */
        {
            const int aBeg = adjStart[body], aEnd = adjStart[body + 1];
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            /* lane 63 of the first round also waits for this body's own previous task (pose + owned duals) */
            for (int base = aBeg; base < aEnd || base == aBeg; base += 63) {
                const int k = base + lane;
                int need = 0; const int* word = nullptr;
                if (lane == 63) {
                    if (base == aBeg && it > 0) { word = S.dualDone + body; need = it; }
                } else if (k < aEnd) {
                    const int o = adjOther[k];
                    if (o >= 0 && inc_kind(adjRef[k]) != FK_IGNORE && massPlane[o].y > 0.0f) {
                        if (o > body) { word = S.primalDone + o; need = it + 1; }
                        else if (it > 0) { word = S.dualDone + o; need = it; }
                    }
                }
                bool ok = (word == nullptr);
                unsigned spins = 0;
                while (true) {
                    if (!ok) ok = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023u) == 0) {
                        bool dead = (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTimeoutTicks) ||
                                    __hip_atomic_load(S.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                        if (__any(dead)) { if (lane == 0) __hip_atomic_store(S.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                    }
                }
            }
#ifdef AVBD_DF_FENCES
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); /* keep the payload loads below the polls */
#endif
        }

        /* ---- primal visit ---- */
        const BodyLocal L = load_body_local<true>(B, body, P);
#ifdef AVBD_DF_DEBUG
        if (body == 1 && lane == 63) printf("task t=%lld it=%d body=%d wave=%d W=%d nd=%d dualDone=%d primalDone=%d posy=%.7f\n", t, it, body, w, W, nd,
            __hip_atomic_load(S.dualDone + body, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(S.primalDone + body, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), L.pos.y);
#endif
        AccLane a = init_acc_lane(lane, L);
        const int eBeg = incStart[body], eEnd = incStart[body + 1];
        for (int base = eBeg; base < eEnd; base += 64) {
            const int nHere = min(64, eEnd - base);
            if (lane < nHere) eval_entry<true>(inc[base + lane], B, M, gf, L, alpha, writeRows, rows + lane * kEntryRec);
            wave_sync();
            accumulate_rows(a, rows, nHere);
            wave_sync();
        }
        if (lane < 27) accs[lane] = a.acc;
        wave_sync();
        if (lane == 0) {
            f3 p; q4 q;
            solve_pose(accs, L, p, q);
            sanitize3(p, cntW);
            sanitizeq(q, cntW);
            const float radius = B.plane(BP_MASS)[body].w;
            st4<true>(B.plane(BP_POSR) + body, pack3(p, radius));
            st4<true>(B.plane(BP_QUAT) + body, packq(q));
        }
        drain_stores();                                            /* every lane's sc1 stores have left the wave */
        if (lane == 0) __hip_atomic_store(S.primalDone + body, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        /* ---- dual update of the forces this body owns (solver.cpp:411-430) ---- */
        if (it < P.iterations) {
            for (int base = eBeg; base < eEnd; base += 64) {
                const int e = base + lane;
                if (e < eEnd) {
                    const int ent = inc[e];
                    int other;
                    if (entry_owned(ent, body, B, M, gf, other)) {
                        const int kind = inc_kind(ent), idx = inc_index(ent);
                        const bool lastDual = it == P.iterations - 1; /* rows are API-visible only after the step */
                        if (kind == FK_MANIFOLD) dual_contact<true>(B, M, idx, P, alpha, lastDual);
                        else if (kind == FK_JOINT) { if ((idx & 1) == 0) dual_gforce<true>(B, gf[idx >> 1], P, lastDual); }
                        else dual_gforce<true>(B, gf[idx], P, lastDual);
                    }
                }
            }
            drain_stores();
        }
        if (lane == 0) __hip_atomic_store(S.dualDone + body, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* Forces with no dynamic endpoint (static-static manifolds, joints between statics) take no part in any
 * primal visit; their dual updates (which only feed the diagnostics and the API-visible rows) run here. */
__global__ void k_dual_static(BodyTable B, ManifoldTable M, const Counters* cnt, GForce* gf, int ng, SolverParams P)
{
    const int nm = cnt->numManifolds;
    const float4* mass = B.plane(BP_MASS);
    const int total = nm * 4 + ng;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < nm * 4) {
            const int m = i >> 2;
            if ((i & 3) >= M.nc[m]) continue;
            const int2 pr = M.pair[m];
            if (mass[pr.x].y > 0.0f || mass[pr.y].y > 0.0f) continue;
            for (int it = 0; it < P.iterations; ++it) dual_contact<false>(B, M, i, P, P.postStabilize ? 1.0f : P.alpha, true);
        } else {
            GForce& f = gf[i - nm * 4];
            bool dynA = f.bodyA >= 0 && mass[f.bodyA].y > 0.0f, dynB = f.bodyB >= 0 && mass[f.bodyB].y > 0.0f;
            if (dynA || dynB) continue;
            for (int it = 0; it < P.iterations; ++it) dual_gforce<false>(B, f, P, true);
        }
    }
}

} // namespace avbd
