/*
 * avbd_fast_api.h — host-side interface of the throughput sweep ("fast mode", avbd_set_option(ctx, "sweep", 1) / --mode fast).
 *
 * The exact schedules (avbd_resident.h, avbd_groups.h, avbd_iterate.h) replay the reference's Gauss-Seidel sweep
 * (solver.cpp:344-409: bodies in list order, row sums in list order) bit for bit.  The fast mode keeps every row's
 * arithmetic (avbd_rows.h: same expression trees, -ffp-contract=off, IEEE divide / sqrt) and changes only
 *   - the ORDER OF BODIES: a graph colouring of the swept bodies, rebuilt on the device every step; the bodies of a
 *     colour do not share a force, so a colour is one bulk pass (multi-colour Gauss-Seidel);
 *   - the ORDER OF SUMS: the 27 entries of a body's 6x6 system are reduced over the lanes of its sub-group with DPP
 *     butterflies instead of in list order;
 *   - WHO runs a force's dual update (solver.cpp:411-430): the first visitor of the next iteration, right before its
 *     own computeConstraint — no pose changes in between, so every force still sees the reference's sequence
 *     computeConstraint / dual / computeConstraint on the same poses.
 * Results differ from the exact mode within the tolerance DESIGN.md section 2b states; tests/test_gpu_fast.py measures it.
 *
 * This translation unit (avbd_fast.hip) is compiled on its own; avbd_capi.hip owns the buffers and calls fast_launch_*.
 */
#pragma once
#include "avbd_types.h"

namespace avbd {

constexpr int kFastMaxColors = 64;
constexpr int kFastClasses = 6;                 /* lanes per body: 1 (no incidence), 4, 8, 16, 32, 64 */
constexpr int kFastSyncWords = 2048;            /* grid barrier: [0, 1024) one word per workgroup, [1024, 1088) one per group of 64 workgroups; [1536, 1600) the XCD roll call of the one-XCD form */
constexpr int kFastXcdSlots = 1536;
constexpr int kFastIsolatedRow = 63;            /* the last row of the (colour, class) group table holds the swept bodies without a swept neighbour (their colour is 0): their tiles come last,
                                                 * FastPlan.colorTileStart[kFastIsolatedRow] is where — no visit of theirs ever waits, and nobody waits for them */
constexpr int kPlanSingleBodies = 1024;         /* up to this many bodies ONE workgroup plans the step: colours in LDS, workgroup barriers */
constexpr int kFastSlotPlanes = 6;
constexpr int kFastBodyPlanes = 4;

/* the plan of one step, written by k_fast_plan, read by k_fast_sweep (and, for statistics, by the host) */
struct FastPlan {
    int numColors;
    int numTiles;
    int bad;            /* 1: a swept body has more than 64 incidence entries, 2: an incidence entry that is not a Manifold contact,
                         * 4: more than kFastMaxColors colours, 8: tile table overflow, 256: a grid barrier / stamp wait timed out */
    int rounds;         /* rounds the colouring took */
    int numSwept;
    int seed;           /* seed of the priorities' tie-breaking hash (moves on when a fresh colouring needed more colours than the one before) */
    int lastFreshColors;
    int pad[1];
    int colorTileStart[kFastMaxColors + 1];
    int groupCount[kFastMaxColors * kFastClasses];
    int groupBodyBase[kFastMaxColors * kFastClasses];
    int groupTileBase[kFastMaxColors * kFastClasses];
    int remaining[4];   /* uncoloured bodies after a round (three slots in rotation) */
    int errInfo[8];
};

struct FastBuffers {
    FastPlan* plan;
    int*    color;      /* [bodyCap] colour of a swept body, -2: not swept */
    const int* colorPrev; /* [bodyCap] last step's colours (the two arrays alternate) */
    int*    cbody;      /* [bodyCap] bodies by (colour, class) */
    int2*   tiles;      /* [tileCap] {first index into cbody, bodies | class << 16} */
    unsigned long long* tileMask; /* [tileCap] lanes of the tile that hold an entry */
    float4* slotC;      /* [kFastSlotPlanes][slotCap] per-lane incidence constants, slot = tile * 64 + lane */
    float4* bodyC;      /* [kFastBodyPlanes][bodyCap] per-body constants by cbody position */
    float4* xpose;      /* [2 * bodyCap] stamped pose records of the swept bodies */
    float4* xS;         /* [2 * (4 * mCap + 2 * gCap)] per contact slot (a Joint half behind the contacts'): (lambda n t1 t2, visit << 1 | stick), (penalty n t1 t2, stamp) —
                         * neighbours in memory: one 64-byte sector per lane and visit */
    unsigned* sync;     /* [kFastSyncWords] */
    GForce* gf;         /* Joint records (kind FK_JOINT); a joint half's state records sit behind the contacts': slot jointSlotBase + 2 * g + part */
    int jointSlotBase;
    int tileCap, slotCap, bodyCap, mCap;
};

struct FastLaunch {
    BodyTable B; ManifoldTable M; SolverParams P;
    const int* adjStart; const int* adjRef; const int* adjOther; const int* incStart; const int* inc;
    Counters* cnt;
    FastBuffers F;
    unsigned syncBase;      /* unique per launch: the barrier targets of the launch are syncBase + 1 .. */
    int itBegin, itEnd;     /* iterations [itBegin, itEnd) of solver.cpp:340-431 */
    int finalPass;          /* after the last iteration: the last dual + the write-back into the Manifold / body tables */
    int stamps;             /* 1: waves wait on the stamps of the records they read; 0: grid barrier per colour */
    int numCU;
    int joints;             /* the scene has Joint forces: the joint-aware kernels */
    int sleepSel;           /* back-off between two looks at the records: s_sleep 1 / 4 / 16 / 64 */
    int xcdLocal;           /* latency form: consecutive tiles on one XCD */
    int oneXcdTiles;        /* latency form: a step of at most this many tiles runs on the workgroups of ONE XCD (those with blockIdx % 8 == 0, checked against HW_REG_XCC_ID) and hands
                             * its records over with plain stores that stay in that XCD's L2 (0: never) */
    int seedBase;           /* option fast_seed: added to the plan's own seed */
    int reuse;              /* k_fast_plan: colorPrev holds last step's colouring of the same bodies */
    int freshDue;           /* ... and this step is one of the scheduled slots for a fresh colouring (every fast_recolor-th step): taken when the kept colouring has more colours than the best
                             * fresh one lately; 2: taken in any case (every 32nd slot: a pile that changed may colour with fewer) */
};

/* occupancy of the kernels on `device` (workgroups per CU), 0 when one does not fit */
int fast_query(int device, int* planBlocksPerCU, int* sweepBlocksPerCU);
/* colouring + tile table + per-lane constants for this step's topology (after the narrow phase, topology and predictor) */
hipError_t fast_launch_plan(hipStream_t stream, const FastLaunch& L, int planBlocksPerCU);
/* the iteration loop; `tilesHint` = tiles of the last known plan (sizes the grid; the kernel itself reads the plan) */
hipError_t fast_launch_sweep(hipStream_t stream, const FastLaunch& L, int sweepBlocksPerCU, int tilesHint, int forceBlocks, int form);

} // namespace avbd
