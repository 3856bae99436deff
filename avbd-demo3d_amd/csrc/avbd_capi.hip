/*
 * avbd_capi.hip — implementation of the C ABI in include/avbd_hip.h (libavbd_hip.so).
 *
 * Host orchestration of one Solver::step() (solver.cpp:255-514) on one gfx950 device / one stream:
 *
 *   begin → K1 broadphase → K2 narrowphase(+decay) → topology (scan, fill, sort/expand, levels)
 *         → K4 predict → [readback: counters + level offsets]
 *         → iterations × ( K5 primal per level, K6 dual ) → K7 velocity → K8 diagnostics
 *
 * The only host<->device synchronisation inside a step is the small readback of the level
 * schedule.  No CPU fallback exists: every entry point fails with AVBD_E_HIP / AVBD_E_NODEVICE
 * when the device or the kernels are unavailable.
 */
#include "../../include/avbd_hip.h"
#include "avbd_kernels.hip"
#include "avbd_fast_api.h"
namespace avbd {   /* instantiated in avbd_single.hip (compiled with -O3) */
extern template __global__ void k_iterate_single<false>(BodyTable, ManifoldTable, GForce*, const int* __restrict__, const int* __restrict__, const Counters* __restrict__,
                                                        const int* __restrict__, const int* __restrict__, SolverParams, int*, Counters*, int, int, ResidentExchange);
extern template __global__ void k_iterate_single<true>(BodyTable, ManifoldTable, GForce*, const int* __restrict__, const int* __restrict__, const Counters* __restrict__,
                                                       const int* __restrict__, const int* __restrict__, SolverParams, int*, Counters*, int, int, ResidentExchange);
}
#include <hip/hip_ext.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>

using namespace avbd;

namespace {

constexpr int kReadbackLevels = 8192;

struct Readback {
    Counters cnt;
    GroupPlan plan;
    int fastHdr[8];            /* first words of the FastPlan of the last fast step (numColors, numTiles, bad, rounds, numSwept), copied behind every fast sweep */
    int lvlStart[kReadbackLevels + 1];
};

template <class T>
struct DevArr {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = count;
        if (!count) return hipSuccess;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } n = 0; }
};

struct ManifoldBuf {
    DevArr<int2> pair; DevArr<uint64_t> key; DevArr<int> nc; DevArr<float> fric;
    DevArr<float4> cRA, cRB, cN, cU1, cU2, cLam, cPen, cC, cFmin, cFmax; DevArr<int2> cFS;
    int cap = 0;
    hipError_t alloc(int c)
    {
        cap = c;
        hipError_t e;
        if ((e = pair.alloc(c)) != hipSuccess) return e;
        if ((e = key.alloc(c)) != hipSuccess) return e;
        if ((e = nc.alloc(c)) != hipSuccess) return e;
        if ((e = fric.alloc(c)) != hipSuccess) return e;
        size_t k = (size_t)c * 4;
        if ((e = cRA.alloc(k)) != hipSuccess) return e;
        if ((e = cRB.alloc(k)) != hipSuccess) return e;
        if ((e = cN.alloc(k)) != hipSuccess) return e;
        if ((e = cU1.alloc(k)) != hipSuccess) return e;
        if ((e = cU2.alloc(k)) != hipSuccess) return e;
        if ((e = cLam.alloc(k)) != hipSuccess) return e;
        if ((e = cPen.alloc(k)) != hipSuccess) return e;
        if ((e = cC.alloc(k)) != hipSuccess) return e;
        if ((e = cFmin.alloc(k)) != hipSuccess) return e;
        if ((e = cFmax.alloc(k)) != hipSuccess) return e;
        if ((e = cFS.alloc(k)) != hipSuccess) return e;
        return hipSuccess;
    }
    void release()
    {
        pair.release(); key.release(); nc.release(); fric.release(); cRA.release(); cRB.release(); cN.release(); cU1.release(); cU2.release();
        cLam.release(); cPen.release(); cC.release(); cFmin.release(); cFmax.release(); cFS.release(); cap = 0;
    }
    ManifoldTable view() const
    {
        ManifoldTable t;
        t.pair = pair.p; t.key = key.p; t.nc = nc.p; t.fric = fric.p; t.cRA = cRA.p; t.cRB = cRB.p; t.cN = cN.p; t.cU1 = cU1.p; t.cU2 = cU2.p;
        t.cLam = cLam.p; t.cPen = cPen.p; t.cFS = cFS.p; t.cC = cC.p; t.cFmin = cFmin.p; t.cFmax = cFmax.p; t.cap = cap;
        return t;
    }
};

} // namespace

struct AvbdContext {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    AvbdParams params;
    int stepIndex = 0;
    int stepsSinceLoad = 0;                  /* steps this context has run on its present content (avbd_clear / a restore start it again): ownership may be set while it is 0 */

    /* host mirror of bodies (authoritative only while !deviceAhead) */
    std::vector<AvbdBody> hBodies;
    std::vector<float> hI, hJ; /* 9 floats per body: inertiaTensor / invInertiaTensor, column-major */
    bool bodiesStale = true;   /* host mirror must be (re)uploaded */
    bool deviceAhead = false;  /* device holds newer dynamic state than hBodies */

    /* device body table */
    DevArr<float4> dPlanes;
    int bodyCap = 0;

    /* manifolds */
    ManifoldBuf mbuf[2];
    int cur = 0;
    int mCap = 0;

    /* generic forces */
    std::vector<GForce> hG;
    DevArr<GForce> dG;
    int gCap = 0;
    bool gStale = false;       /* host list changed → upload */
    bool gDeviceAhead = false; /* device rows newer than hG */
    uint64_t hostKeyCounter = 1;

    /* topology */
    DevArr<int> adjStart, incStart, degForce, degInc, level, order, lvlStart, lvlCount, adjRef, adjOther, inc;
    DevArr<uint64_t> adjKey;
    DevArr<int2> adjTmp;       /* (force reference, other body) in scatter order: k_topology_small */
    DevArr<int2> cand;
    /* processing order of k_narrowphase (NarrowOrder, avbd_types.h): slots by class and class counters, of the step before and for the step after */
    DevArr<int> dNarrowPerm[2], dNarrowCnt; int narrowPar = 0; bool narrowOrderValid = false; bool narrowOrder = true;
    /* grid broadphase (scenes above 4096 bodies): radius classes from the host mirror, hashed cell table */
    DevArr<unsigned char> dIsBig; DevArr<int> dBigList, dCellHead, dCellNext;
    int nBig = 0, gridBuckets = 0; float gridCell = 0.0f; bool gridBroadphase = true;   /* AVBD_BROADPHASE=n2: always the O(n^2) tiles */
    int candCap = 0, adjCap = 0, incCap = 0, lvlCap = 0;
    bool adjDirty = true;

    DevArr<Counters> dCnt;
    DevArr<DiagAccum> dDiag;
    DevArr<int> dNext;
    DevArr<int> xProbe; size_t probeCap = 0;   /* out-of-order sweep: per incidence entry, who it waits for */
    DevArr<int> dFlags;        /* primalDone[bodyCap] | dualDone[bodyCap] | error word (data-flow kernel) */
    int mode = 1;              /* 1: persistent data-flow iteration kernel (default), 0: one launch per level */
    int iterBlocks = 512;      /* resident workgroups of k_iterate */
    int iterLdsPad = 0;        /* extra dynamic LDS per workgroup: > 43 KB forces one workgroup per CU */
    int iterSleep = 1;         /* poll back-off selector (s_sleep 1/2/4/8) */
    int iterForceMode = 0;     /* 0 auto, 1 latency (1 workgroup/CU), 2 throughput (2 workgroups/CU) */
    DevArr<float4> xPose, xVisit, xDual; /* stamped exchange records of the resident schedule (avbd_resident.h) */
    DevArr<float4> xPrivN, xPrivPose;   /* resident schedule with several bodies per wave: state parked between a body's tasks */
    DevArr<float4> xParkC, xParkS, xParkB; size_t parkCap = 0, parkBodyCap = 0;   /* group schedule, several bodies per sub-group: packed parked records */
    size_t privNCap = 0, privPoseCap = 0;
    bool residentMultiAllowed = true;   /* AVBD_RESIDENT_MULTI=0: scenes with more swept bodies than waves use the task schedule */
    bool residentMultiLaunched = false;
    hipEvent_t iterDone = nullptr;      /* recorded behind every k_iterate of avbd_group_step (contexts sharing a device take turns) */
    bool packFused = false;             /* the last k_iterate launch filled dSend itself */
    bool residentSharded = true;        /* AVBD_RESIDENT_SHARDED=0: sharded runs (one launch per iteration) use the task schedule */
    bool residentForceLite = false;     /* AVBD_RESIDENT_FORCE_LITE=1: the two-workgroups-per-CU launch for every scene size */
    int residentLiteFrom = 1;           /* AVBD_RESIDENT_LITE_FROM=<k>: more than k bodies per wave → two workgroups per CU (0: never) */
    int residentMultiMax = 64;          /* AVBD_RESIDENT_MULTI_MAX: most bodies one wave may own */
    unsigned xSerial = 0;      /* launch serial: stamp base = serial * 64 */
    bool residentLaunched = false; /* the last k_iterate launch had the resident schedule enabled (the device may still veto it) */
    bool fusedFront = true;        /* AVBD_FUSED_FRONT=0: separate broadphase and narrow-phase launches */
    bool orderByIndex = true;      /* data-flow schedules sweep in descending creation index; AVBD_ORDER=levels: dependency levels (always in AVBD_MODE=levels) */
    bool fusedTopology = true;     /* AVBD_FUSED_TOPOLOGY=0: always the multi-launch topology chain */
    bool topologyDidPredict = false;
    bool residentSpread = true;    /* AVBD_RESIDENT_SPREAD=0: slot = linear wave index */
    int residentPatience = 300;     /* AVBD_RESIDENT_PATIENCE: ticks of 10 ns (see ResidentExchange.patience) */
    int residentOutOfOrder = -1;    /* AVBD_RESIDENT_OOO=0/1: a wave with several bodies sweeps them strictly in order / takes whichever is ready; default: by size */
    int residentQuiet = 500;       /* AVBD_RESIDENT_QUIET: permille of the previous wait spent asleep before polling */
    bool residentAllowed = true; /* AVBD_RESIDENT=0 disables the resident schedule */
    bool liteResident = true;    /* two workgroups of k_iterate<2> fit a CU (checked at create); otherwise every launch stays at one workgroup per CU */
    /* group schedule (avbd_groups.h): several bodies in flight per wavefront, for scenes with more swept bodies than resident waves */
    DevArr<GroupPlan> dPlan; DevArr<int> dUnit; int unitCap = 0;   /* dUnit: the bodies of each class in sweep order, [3 * bodyCap] */
    /* scans over tiles of 4096 bodies (k_scan, k_group_plan): one stamped sum per tile; every launch takes a new stamp */
    DevArr<tile_v4u> dTileAgg; unsigned tileStamp = 0;
    int numSolved = 0;             /* bodies this rank sweeps (known at upload) */
    bool orderValid = false;       /* c->order / level / lvlStart hold the sweep by creation index of the bodies on the device */
    int groupsMode = -1;           /* AVBD_GROUPS=0 never, 1 for every eligible scene (tests), default: more swept bodies than one-body-per-wave waves */
    int groupsMaxRounds = 16;      /* AVBD_GROUPS_ROUNDS: most bodies one sub-group may serve, at most 16 (1: only scenes whose bodies all fit a sub-group of their own;
                                    * larger ones then take the resident schedule's parked sweep: 64 000 cubes 8.4 ms per iteration loop against 7.0 ms) */
    int groupsWaves = 0;           /* AVBD_GROUPS_WAVES: resident waves the plan may use (tests: few waves, many rounds); 0 = all */
    /* sharded runs as one persistent launch per step with the exchange on a second stream (ShardExchange, avbd_groups.h) */
    int shardPersistent = -1;      /* AVBD_SHARD_PERSISTENT=0 never, 1 whenever the prerequisites hold; default (-1): wherever the form has been validated —
                                    * one rank, or the contexts of avbd_group_step on ONE device.  With a real communicator (one process per GPU) or contexts on
                                    * different devices it is opt-in (=1) until it has run on a multi-GPU box: the exchange stream would wait on
                                    * and read words that live on another device */
    bool canWaitValue = false;     /* hipStreamWaitValue32 / WriteValue32 work on this device */
    hipStream_t stream2 = nullptr; /* exchange stream: wait for an iteration -> all-gather -> release its readers */
    hipEvent_t evBegin = nullptr, evGathered = nullptr;
    unsigned* sigDone = nullptr;   /* hipMallocSignalMemory word: what the exchange stream waits on (one system-scope atomic per iteration raises it) */
    unsigned* sigGather = nullptr; /* UNCACHED device memory (hipDeviceMallocUncached): written by hipStreamWriteValue32 behind every all-gather, POLLED by every
                                    * wave that reads a remote row.  Not signal memory (thousands of waves polling it made a sharded iteration take 0.7 ms) and not
                                    * plain device memory (the command processor's write does not reach a line an XCD's L2 already holds: the waves of that
                                    * XCD kept reading the old value and the launch timed out) */
    DevArr<unsigned> dDone; DevArr<float4> dGather[2]; DevArr<int> dRow; size_t gatherRows = 0;
    int shardWaves = 0;            /* waves the plan of a sharded step may use (contexts that share a device split them; room for the collective's kernel) */
    bool shardRanPersistent = false;
    /* user-defined Force subclasses: rows evaluated by the caller between launches (avbd_add_host_force) */
    std::vector<int> hostForceG;           /* ordinal -> index in hG */
    std::vector<HostRowsDev> hHostRows; DevArr<HostRowsDev> dHostRows;
    AvbdHostForceFn hostFn = nullptr; void* hostUser = nullptr;
    bool shardMulti = false;       /* the plan of the current sharded step has several rounds */
    bool shardGateKernel = false;  /* AVBD_SHARD_GATE=kernel: the exchange stream waits and publishes with small kernels instead of hipStreamWaitValue32 / WriteValue32 */
    bool singleKernel = true;      /* AVBD_SINGLE_KERNEL=0: one-body-per-wave scenes run inside k_iterate<1> instead of k_iterate_single */
    bool poolDirty = false;        /* pool_per_body changed since the pools were allocated: the next upload of bodies re-allocates them */
    int frontWideFrom = kFrontWideFrom;   /* front_wide_from: k_front_small with 256-thread workgroups and 16-body slices from this many bodies on */
    int gridFrom = 3200;           /* grid_from: the grid broadphase (and the narrow phase as a launch of its own) from this many bodies on: the fused front end's
                                    * broadphase is all pairs (3 375 cubes: front end 0.079 -> 0.071 ms, the step 0.986 -> 0.977; 2 744 cubes 0.060 -> 0.070: it stays below) */
    int poolPerBody = 16;          /* pool_per_body: manifolds (and candidate pairs) the pools hold per body of the context's capacity; read when the pools are (re)allocated */
    bool raceCheck = false;        /* AVBD_RACE_CHECK=1: k_race_check after every topology rebuild (debug) */
    DevArr<int> dSlotOf;
    /* exact cuts (avbd_cuts.h): the stamped records of all ranks in one address space; the first context of a group owns it */
    bool jointAngularFix = false;  /* avbd_set_joint_angular_sign(ctx, 1): Joint angular rows with J = dC/dtheta instead of the reference's sign */
    bool shardExactStrict = true;  /* avbd_set_shard_mode 1: a step that cannot run exact cuts FAILS (AVBD_E_INVALID + reason); 2: it takes the all-gather form, AvbdStepStats.cutMode says so */
    int cutModeRan = 0;            /* AvbdStepStats.cutMode of the last step */
    bool shardExact = false;       /* AVBD_SHARD_EXACT=1: avbd_group_step sweeps every rank's slab with the single-GPU protocol (bit-identical to one GPU) */
    DevArr<float4> xsPose, xsVisit, xsDual; size_t xsLocalSlots = 0, xsTotalSlots = 0; int xsRanks = 0; unsigned xsSerial = 0;
    DevArr<int> dOwner, dCutOf, dCutMan, dRecIdx; DevArr<unsigned long long> dCutKeys; DevArr<CutCounters> dCutCnt;
    size_t cutOfCap = 0, recIdxCap = 0;
    hipEvent_t evLoop = nullptr, evExact = nullptr;
    DevArr<unsigned long long> dSpan; bool spanPending = false;   /* profiling of the persistent sharded launch: the kernel's own start / end stamps (shard_launch_persistent) */
    std::vector<ExactRankArgs> hExactArgs; DevArr<ExactRankArgs> dExactArgs;
    /* chains of hand-offs share a workgroup (avbd_place.h): report of the last iteration launch, tails, this step's table */
    DevArr<int2> dCrit; DevArr<int> dTail, dPlaceSlot, dPlace; DevArr<PlaceInfo> dPlaceInfo;
    bool placeChains = true;       /* AVBD_PLACE=0: waves take the slots of the sweep order */
    bool placeQueued = false;      /* this step: k_place_chains has been queued on the current topology */
    int placeEvery = 8;            /* the table is rebuilt every so many steps while the swept bodies stay the same (place_every; 1 = every step) */
    unsigned sweptSerial = 0;      /* counts the uploads that can change which bodies are swept */
    unsigned placeSerial = ~0u;    /* sweptSerial | waves the table in dPlace was built for */
    int placeWaves = 0, placeAge = 0;
    unsigned smallOrderSerial = ~0u; /* sweptSerial for which k_topology_small's sweep by creation index is in c->order / level / lvlStart */
    bool groupsPlanned = false;    /* this step: k_group_plan has been queued */
    bool rowsStale = true;         /* ownership changed: the gather rows of the bodies must be uploaded again */
    bool groupsLaunched = false;   /* the last iteration loop was offered to k_iterate_groups (the device may still decline: GroupPlan.bad) */
    int numStatic = 0;
    bool statsFresh = true;

    /* sharding over GPUs (one context per GPU): static ownership of the dynamic bodies */
    int nranks = 1, rank = 0;
    std::vector<int> hOwner;            /* per body: owning rank, -1 = replicated (static) */
    DevArr<int> dSolved;                /* per body: 1 = dynamic and owned by this rank */
    std::vector<int> hPerm, hRankStart; /* solved bodies grouped by rank */
    std::vector<int> hRow;              /* row of every body in an all-gather buffer (avbd_exchange_layout) */
    DevArr<int> dPerm, dRankStart, dSendSlot;
    int maxCount = 0;                   /* rows every rank contributes to the all-gather */
    DevArr<float4> dSend, dRecv;
    /* exact cuts between processes: this rank's window, the peers' windows as this process sees them (own window at [rank]) */
    float4* xwOwn = nullptr; float4* xwPeer[kMaxExactRanks] = {}; bool xwAttached = false; bool xwSharedDevice = false; unsigned xwSerial = 0; int xwBodyCap = 0, xwMCap = 0, xwGCap = 0, xwRanks = 0;
    int cutRanks = 0;
    int narrowLanes = 0;           /* narrow_lanes option: 32 / 64 lanes per workgroup of k_narrowphase, 0: by the number of pairs */
    int diagBlocksCap = 256;       /* workgroups of the contact diagnostics: each ends with five atomics on ONE line, 13 ns apiece whoever issues them (64 000 cubes: 1024 -> 50 us, 256 -> 28 us, 128 -> 34 us) */
    bool debugSharedMulti = false;
    int iterBlocksMax = 0;         /* what the occupancy query allows (avbd_create) */
    bool exactAllowed = true;      /* the exact-cut kernels fit two workgroups per CU on this device / this build */
    int groupDepKeep = 1;          /* AVBD_GROUP_DEP_KEEP=0: classes that take a task in hand keep the watched records of round 3 (ResidentExchange::depKeep) */
    int groupDep = 1;              /* several bodies per sub-group on one GPU: arrival counters instead of watched records (AVBD_GROUP_DEP=0: the watched records of round 3) */
    DevArr<unsigned> xDepCnt; DevArr<int> xRankOf;
    int groupPool = 1, groupPoolMinRounds = 3;   /* ResidentExchange::poolMode / poolMinRounds (classes with fewer bodies per sub-group take a task in hand instead) */
    int groupSpecMode = 0, groupSpecRounds = 4;   /* ResidentExchange::specMode / specRounds (AVBD_GROUP_SPEC, AVBD_GROUP_SPEC_ROUNDS) */
    bool capDataflow = true, capLite = true, capGroups = true;   /* what the occupancy query of avbd_create found possible on this device / build: avbd_set_option cannot switch a schedule on that is not */
    bool xwPoisoned = false;            /* a step of the exact-cut form between processes failed after its front end had run: the state is half a step ahead */
    int liveAtCreation = 1;             /* contexts of this process alive on the device when this one made its streams, itself included (hardware queues: avbd_group_step) */
    bool turnReleased = false;          /* avbd_destroy has taken this context out of its device's count (turn_begin / turn_end) */
    int exchangeFloats = 8;             /* avbd_set_exchange_floats: 8 = poses every iteration; 13 = + the velocities once per step */
    DevArr<unsigned> dXchMismatch;      /* words of gathered velocities that differed from the re-derived ones (must stay 0) */
    void* comm = nullptr;               /* ncclComm_t (RCCL, loaded lazily) */
    bool forceSharded = false;          /* AVBD_FORCE_SHARDED=1: per-iteration exchange schedule even with one rank (tests) */
    /* throughput sweep (avbd_fast.hip; option "sweep"): colouring + tiles rebuilt every step, one persistent launch for the loop */
    int sweep = 0;                      /* 0: the exact schedules (default), 1: the fast sweep wherever it can run (contacts and ignore markers, one GPU or the all-gather form) */
    int fastStamps = 1;                 /* fast_stamps: 1 stamped records between colours, 0 a grid barrier per colour (cross-check: same bits) */
    int fastRecolor = 0;                /* fast_recolor: every how many steps every body is coloured afresh (1: every step; 0: by size); in between last step's colouring is kept where it still is one */
    int fastSleep = -1;                 /* fast_sleep: back-off between two looks of a waiting lane, s_sleep 1 / 4 / 16 / 64 (-1: by the kernel's form) */
    int fastXcd = 0;                    /* fast_xcd: latency form, consecutive tiles fill one XCD before the next */
    int debugFastXcdMismatch = 0;       /* debug_fast_xcd_mismatch: the one-XCD form's roll call answers "several XCDs": its workgroups hand over with sc1 stores (the path a dispatcher that deals workgroups otherwise would take) */
    int fastOneXcd = 4096;              /* fast_one_xcd: latency form, a step of at most this many tiles — and at most one per wave of an XCD — runs on one XCD and hands records over through its L2 (0: never) */
    int fastSeed = 0;                   /* fast_seed: added to the seed of the colouring's tie-breaking hash */
    int fastForm = 0;                   /* fast_form: 0 by size, 1 the latency form of k_fast_sweep (a wave per SIMD, contacts in two passes), 2 the throughput form */
    int fastBlocks = 0;                 /* fast_blocks: workgroups of k_fast_sweep (0: by the tile count of the last plan) */
    int fastPlanPerCU = 0, fastSweepPerCU = 0, numCU = 1;
    bool capFast = false;
    unsigned fastSerial = 0;
    bool fastStepDeclined = false;      /* sharded: the plan of the current step declined it (read back once per step) */
    bool fastLaunched = false;          /* the iteration loop of the last step was offered to the fast sweep */
    DevArr<FastPlan> fPlan; DevArr<int> fColor[2], fCbody; int fColorCur = 0; unsigned fastColorSerial = ~0u;   /* fColor: this step's / last step's colours; fastColorSerial: sweptSerial the colours in fColor[fColorCur] belong to */ DevArr<int2> fTiles; DevArr<unsigned long long> fTileMask; DevArr<float4> fSlotC, fBodyC, fXpose, fXS; DevArr<unsigned> fSync;
    int fTileCap = 0, fBodyCap = 0, fMCap = 0, fGCap = 0;
    Readback* hRb = nullptr; /* pinned */
    AvbdStepStats stats{};
    int lastLevels = 0;

    /* profiling */
    bool profiling = false;
    AvbdProfile prof{};
    std::vector<hipEvent_t> evPool;
    size_t evUsed = 0;
    struct Span { int family; size_t e0, e1; };
    std::vector<Span> spans;

    int fail(int code, const char* what, hipError_t e = hipSuccess)
    {
        char buf[512];
        if (e != hipSuccess) std::snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        else std::snprintf(buf, sizeof(buf), "%s", what);
        err = buf;
        return code;
    }
};

#define HIPCHK(ctx, call)                                                           \
    do {                                                                            \
        hipError_t _e = (call);                                                     \
        if (_e != hipSuccess) return (ctx)->fail(AVBD_E_HIP, #call, _e);            \
    } while (0)

namespace {

BodyTable body_table(AvbdContext* c)
{
    BodyTable t; t.planes = c->dPlanes.p; t.solved = c->dSolved.p; t.cap = c->bodyCap; t.n = (int)c->hBodies.size();
    return t;
}

SolverParams solver_params(const AvbdContext* c)
{
    SolverParams p;
    p.dt = c->params.dt; p.gx = c->params.gravity[0]; p.gy = c->params.gravity[1]; p.gz = c->params.gravity[2];
    p.iterations = c->params.iterations; p.alpha = c->params.alpha; p.beta = c->params.beta; p.gamma = c->params.gamma;
    p.postStabilize = c->params.postStabilize;
    return p;
}

int grid_for(long long items, int block, int maxBlocks = 4096)
{
    long long g = (items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > maxBlocks) g = maxBlocks;
    return (int)g;
}

/* (re)allocate everything that scales with the body count; preserves manifolds */
int ensure_capacity(AvbdContext* c, int nBodies)
{
    if (nBodies <= c->bodyCap && c->mCap > 0 && !c->poolDirty) return AVBD_OK;
    c->poolDirty = false;
    int newCap = std::max(1024, c->bodyCap);
    while (newCap < nBodies) newCap *= 2;
    const int newMCap = (int)std::min<long long>(1ll << 28, std::max<long long>(8192, (long long)newCap * c->poolPerBody));   /* (manifolds; as many candidate pairs) */
    HIPCHK(c, hipStreamSynchronize(c->stream));

    HIPCHK(c, c->dPlanes.alloc((size_t)BP_COUNT * newCap));
    HIPCHK(c, c->dSolved.alloc(newCap));
    HIPCHK(c, hipMemsetAsync(c->dSolved.p, 0, newCap * sizeof(int), c->stream));
    HIPCHK(c, c->dPerm.alloc(newCap));
    c->bodyCap = newCap;

    if (newMCap > c->mCap) {
        /* keep live manifolds: copy the current buffer into the new allocation */
        ManifoldBuf nb0, nb1;
        HIPCHK(c, nb0.alloc(newMCap));
        HIPCHK(c, nb1.alloc(newMCap));
        if (c->mCap > 0) {
            ManifoldBuf& o = c->mbuf[c->cur];
            size_t m = (size_t)c->mCap;
            HIPCHK(c, hipMemcpy(nb0.pair.p, o.pair.p, m * sizeof(int2), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.key.p, o.key.p, m * sizeof(uint64_t), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.nc.p, o.nc.p, m * sizeof(int), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.fric.p, o.fric.p, m * sizeof(float), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cRA.p, o.cRA.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cRB.p, o.cRB.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cN.p, o.cN.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cU1.p, o.cU1.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cU2.p, o.cU2.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cLam.p, o.cLam.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cPen.p, o.cPen.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cC.p, o.cC.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cFmin.p, o.cFmin.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cFmax.p, o.cFmax.p, 4 * m * sizeof(float4), hipMemcpyDeviceToDevice));
            HIPCHK(c, hipMemcpy(nb0.cFS.p, o.cFS.p, 4 * m * sizeof(int2), hipMemcpyDeviceToDevice));
        }
        c->mbuf[0].release(); c->mbuf[1].release();
        c->mbuf[0] = nb0; c->mbuf[1] = nb1; c->cur = 0;
        c->mCap = newMCap;
        c->candCap = newMCap;
        HIPCHK(c, c->cand.alloc(c->candCap));
        HIPCHK(c, c->dNarrowPerm[0].alloc((size_t)kNarrowClasses * newMCap)); HIPCHK(c, c->dNarrowPerm[1].alloc((size_t)kNarrowClasses * newMCap));
        if (!c->dNarrowCnt.p) { HIPCHK(c, c->dNarrowCnt.alloc((size_t)2 * kNarrowClasses * kNarrowCntStride)); HIPCHK(c, hipMemsetAsync(c->dNarrowCnt.p, 0, (size_t)2 * kNarrowClasses * kNarrowCntStride * sizeof(int), c->stream)); }
        c->narrowOrderValid = false;
    }
    HIPCHK(c, c->xPose.alloc((size_t)3 * newCap));
    HIPCHK(c, c->xVisit.alloc((size_t)4 * c->mCap));
    HIPCHK(c, c->xDual.alloc((size_t)8 * c->mCap));
    HIPCHK(c, hipMemsetAsync(c->xPose.p, 0, (size_t)3 * newCap * sizeof(float4), c->stream));
    HIPCHK(c, hipMemsetAsync(c->xVisit.p, 0, (size_t)4 * c->mCap * sizeof(float4), c->stream));
    HIPCHK(c, hipMemsetAsync(c->xDual.p, 0, (size_t)8 * c->mCap * sizeof(float4), c->stream));
    c->xSerial = 0;
    c->lvlCap = newCap + 2;
    HIPCHK(c, c->adjStart.alloc(newCap + 1));
    HIPCHK(c, c->incStart.alloc(newCap + 1));
    HIPCHK(c, c->degForce.alloc(newCap));
    HIPCHK(c, c->degInc.alloc(newCap));
    HIPCHK(c, c->level.alloc(newCap));
    HIPCHK(c, c->order.alloc(newCap));
    c->orderValid = false;
    ++c->sweptSerial;
    HIPCHK(c, c->dTileAgg.alloc((size_t)newCap / kTileBodies + 2));
    HIPCHK(c, hipMemsetAsync(c->dTileAgg.p, 0, ((size_t)newCap / kTileBodies + 2) * sizeof(tile_v4u), c->stream));
    c->tileStamp = 0;
    HIPCHK(c, c->lvlStart.alloc(c->lvlCap + 1));
    HIPCHK(c, c->lvlCount.alloc(c->lvlCap + 1));
    HIPCHK(c, hipMemsetAsync(c->adjStart.p, 0, (newCap + 1) * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->incStart.p, 0, (newCap + 1) * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->degForce.p, 0, newCap * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->degInc.p, 0, newCap * sizeof(int), c->stream));
    HIPCHK(c, c->dCrit.alloc(newCap)); HIPCHK(c, c->dTail.alloc(newCap)); HIPCHK(c, c->dPlaceSlot.alloc(newCap));
    HIPCHK(c, hipMemsetAsync(c->dCrit.p, 0xff, (size_t)newCap * sizeof(int2), c->stream));
    HIPCHK(c, hipMemsetAsync(c->dTail.p, 0, (size_t)newCap * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->dPlaceSlot.p, 0xff, (size_t)newCap * sizeof(int), c->stream));
    if (!c->dPlace.p) { HIPCHK(c, c->dPlace.alloc(1024)); HIPCHK(c, c->dPlaceInfo.alloc(1)); HIPCHK(c, hipMemsetAsync(c->dPlaceInfo.p, 0, sizeof(PlaceInfo), c->stream)); }
    HIPCHK(c, c->dFlags.alloc((size_t)2 * newCap + 32 + 8192 * 32));
    HIPCHK(c, hipMemsetAsync(c->dFlags.p, 0, ((size_t)2 * newCap + 32 + 8192 * 32) * sizeof(int), c->stream));
    c->adjDirty = true;
    c->bodiesStale = true;
    return AVBD_OK;
}

int ensure_topology_capacity(AvbdContext* c)
{
    const int g = std::max(64, (int)c->hG.size());
    if (g > c->gCap) {
        int ng = std::max(64, c->gCap);
        while (ng < g) ng *= 2;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        DevArr<GForce> nd;
        HIPCHK(c, nd.alloc(ng));
        if (c->dG.p && c->gCap) HIPCHK(c, hipMemcpy(nd.p, c->dG.p, (size_t)c->gCap * sizeof(GForce), hipMemcpyDeviceToDevice));
        c->dG.release();
        c->dG = nd;
        c->gCap = ng;
    }
    const int needAdj = 2 * (c->mCap + c->gCap);
    const int needInc = 8 * c->mCap + 4 * c->gCap;
    if (needAdj > c->adjCap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, c->adjRef.alloc(needAdj));
        HIPCHK(c, c->adjOther.alloc(needAdj));
        HIPCHK(c, c->adjKey.alloc(needAdj));
        HIPCHK(c, c->adjTmp.alloc(needAdj));
        c->adjCap = needAdj;
        c->adjDirty = true;
    }
    if (needInc > c->incCap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, c->inc.alloc(needInc));
        c->incCap = needInc;
        c->adjDirty = true;
    }
    return AVBD_OK;
}

void pack_planes(const AvbdContext* c, std::vector<float4>& h)
{
    const int n = (int)c->hBodies.size(), cap = c->bodyCap;
    h.assign((size_t)BP_COUNT * cap, make_float4(0.f, 0.f, 0.f, 0.f));
    auto P = [&](int plane, int i) -> float4& { return h[(size_t)plane * cap + i]; };
    for (int i = 0; i < n; ++i) {
        const AvbdBody& b = c->hBodies[i];
        P(BP_POSR, i) = make_float4(b.position[0], b.position[1], b.position[2], b.radius);
        P(BP_QUAT, i) = make_float4(b.orientation[0], b.orientation[1], b.orientation[2], b.orientation[3]);
        P(BP_LINV, i) = make_float4(b.linearVelocity[0], b.linearVelocity[1], b.linearVelocity[2], 0.f);
        P(BP_ANGV, i) = make_float4(b.angularVelocity[0], b.angularVelocity[1], b.angularVelocity[2], 0.f);
        P(BP_PLIN, i) = make_float4(b.prevLinearVelocity[0], b.prevLinearVelocity[1], b.prevLinearVelocity[2], 0.f);
        P(BP_PANG, i) = make_float4(b.prevAngularVelocity[0], b.prevAngularVelocity[1], b.prevAngularVelocity[2], 0.f);
        P(BP_IPOS, i) = make_float4(b.initialPosition[0], b.initialPosition[1], b.initialPosition[2], 0.f);
        P(BP_IQUAT, i) = make_float4(b.initialOrientation[0], b.initialOrientation[1], b.initialOrientation[2], b.initialOrientation[3]);
        P(BP_NPOS, i) = make_float4(b.inertialPosition[0], b.inertialPosition[1], b.inertialPosition[2], 0.f);
        P(BP_NQUAT, i) = make_float4(b.inertialOrientation[0], b.inertialOrientation[1], b.inertialOrientation[2], b.inertialOrientation[3]);
        P(BP_SIZE, i) = make_float4(b.size[0], b.size[1], b.size[2], b.friction);
        P(BP_MASS, i) = make_float4(b.mass, b.invMass, b.friction, b.radius);
        const float* I = &c->hI[9 * (size_t)i];
        const float* J = &c->hJ[9 * (size_t)i];
        for (int k = 0; k < 3; ++k) {
            P(BP_I0 + k, i) = make_float4(I[3 * k], I[3 * k + 1], I[3 * k + 2], 0.f);
            P(BP_J0 + k, i) = make_float4(J[3 * k], J[3 * k + 1], J[3 * k + 2], 0.f);
        }
    }
}

int upload_bodies_if_stale(AvbdContext* c)
{
    const int n = (int)c->hBodies.size();
    int rc = ensure_capacity(c, n);
    if (rc) return rc;
    if (!c->bodiesStale) return AVBD_OK;
    std::vector<float4> h;
    pack_planes(c, h);
    HIPCHK(c, hipMemcpyAsync(c->dPlanes.p, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    {   /* who solves which body */
        c->hOwner.resize((size_t)n, 0);
        std::vector<int> solved((size_t)std::max(n, 1), 0);
        c->hRankStart.assign((size_t)c->nranks + 1, 0);
        c->numSolved = 0;
        for (int i = 0; i < n; ++i) {
            const bool dyn = c->hBodies[(size_t)i].invMass > 0.0f;
            if (!dyn) c->hOwner[(size_t)i] = -1;
            else if (c->hOwner[(size_t)i] < 0 || c->hOwner[(size_t)i] >= c->nranks) c->hOwner[(size_t)i] = 0;
            solved[(size_t)i] = dyn && c->hOwner[(size_t)i] == c->rank;
            c->numSolved += solved[(size_t)i];
        }
        c->orderValid = false;
        ++c->sweptSerial;
        /* the exchange layout (who sits where in the all-gather) comes from the one host function the C ABI exports */
        c->hPerm.assign((size_t)std::max(n, 1), 0);
        c->hRow.assign((size_t)std::max(n, 1), -1);
        c->rowsStale = true;
        if (avbd_exchange_layout(n, c->hOwner.data(), c->nranks, c->hRankStart.data(), c->hPerm.data(), c->hRow.data(), &c->maxCount) != AVBD_OK)
            return c->fail(AVBD_E_INVALID, "exchange layout: owner index out of range");
        if (n) {
            if (c->dOwner.n < (size_t)c->bodyCap) HIPCHK(c, c->dOwner.alloc((size_t)c->bodyCap));
            HIPCHK(c, hipMemcpy(c->dOwner.p, c->hOwner.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpyAsync(c->dSolved.p, solved.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->dPerm.p, c->hPerm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
            {   /* row of each body this rank solves inside its all-gather block */
                std::vector<int> slot((size_t)n, 0);
                const int first = c->hRankStart[(size_t)c->rank];
                for (int k = first; k < c->hRankStart[(size_t)c->rank + 1]; ++k) slot[(size_t)c->hPerm[(size_t)k]] = k - first;
                if (c->dSendSlot.n < (size_t)c->bodyCap) HIPCHK(c, c->dSendSlot.alloc((size_t)c->bodyCap));
                HIPCHK(c, hipMemcpy(c->dSendSlot.p, slot.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
            }
        }
        if (c->nranks > 1 || c->forceSharded) {
            HIPCHK(c, c->dRankStart.alloc((size_t)c->nranks + 1));
            HIPCHK(c, hipMemcpyAsync(c->dRankStart.p, c->hRankStart.data(), ((size_t)c->nranks + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
            if (c->dSend.n < (size_t)2 * std::max(1, c->maxCount)) HIPCHK(c, c->dSend.alloc((size_t)2 * std::max(1, c->maxCount)));
            if (c->dRecv.n < (size_t)2 * std::max(1, c->maxCount) * c->nranks) HIPCHK(c, c->dRecv.alloc((size_t)2 * std::max(1, c->maxCount) * c->nranks));
        }
    }
    if (c->gridBroadphase && n >= c->gridFrom) {
        /* radius classes (radii never change after creation): cell edge = 2.002 x the largest radius that is at most 4x the
         * median; anything larger (the ground slab) is "big" and tested against every body */
        std::vector<float> rad((size_t)n);
        for (int i = 0; i < n; ++i) rad[(size_t)i] = c->hBodies[(size_t)i].radius;
        std::vector<float> sorted(rad);
        std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
        const float limit = 4.0f * sorted[(size_t)n / 2];
        std::vector<unsigned char> big((size_t)n, 0);
        std::vector<int> bigList;
        float rSmall = 0.0f;
        for (int i = 0; i < n; ++i) {
            if (rad[(size_t)i] > limit || !(rad[(size_t)i] == rad[(size_t)i])) { big[(size_t)i] = 1; bigList.push_back(i); }
            else rSmall = std::max(rSmall, rad[(size_t)i]);
        }
        c->nBig = (int)bigList.size();
        c->gridCell = std::max(1.0e-3f, 2.002f * rSmall);
        int buckets = 1024;
        while (buckets < 2 * n && buckets < (1 << 18)) buckets <<= 1;
        c->gridBuckets = buckets;
        HIPCHK(c, c->dIsBig.alloc((size_t)n)); HIPCHK(c, c->dBigList.alloc((size_t)std::max(1, c->nBig)));
        HIPCHK(c, c->dCellHead.alloc((size_t)buckets)); HIPCHK(c, c->dCellNext.alloc((size_t)n));
        HIPCHK(c, hipMemcpyAsync(c->dIsBig.p, big.data(), (size_t)n, hipMemcpyHostToDevice, c->stream));
        if (c->nBig) HIPCHK(c, hipMemcpyAsync(c->dBigList.p, bigList.data(), (size_t)c->nBig * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   /* the vectors above die with this scope */
    } else {
        c->gridBuckets = 0;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->bodiesStale = false;
    c->deviceAhead = false;
    return AVBD_OK;
}

int refresh_stats(AvbdContext* c, bool withLevels);

int download_bodies_to_host(AvbdContext* c)
{
    if (!c->statsFresh) { int rc = refresh_stats(c, false); if (rc) return rc; }
    if (!c->deviceAhead || c->bodiesStale) return AVBD_OK;
    const int n = (int)c->hBodies.size(), cap = c->bodyCap;
    if (!n) { c->deviceAhead = false; return AVBD_OK; }
    std::vector<float4> h((size_t)BP_NQUAT * cap + cap); /* planes 0..BP_NQUAT are the dynamic ones */
    HIPCHK(c, hipMemcpyAsync(h.data(), c->dPlanes.p, h.size() * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto P = [&](int plane, int i) -> const float4& { return h[(size_t)plane * cap + i]; };
    auto put3 = [](float* d, const float4& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
    auto put4 = [](float* d, const float4& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; };
    for (int i = 0; i < n; ++i) {
        AvbdBody& b = c->hBodies[i];
        put3(b.position, P(BP_POSR, i)); put4(b.orientation, P(BP_QUAT, i));
        put3(b.linearVelocity, P(BP_LINV, i)); put3(b.angularVelocity, P(BP_ANGV, i));
        put3(b.prevLinearVelocity, P(BP_PLIN, i)); put3(b.prevAngularVelocity, P(BP_PANG, i));
        put3(b.initialPosition, P(BP_IPOS, i)); put4(b.initialOrientation, P(BP_IQUAT, i));
        put3(b.inertialPosition, P(BP_NPOS, i)); put4(b.inertialOrientation, P(BP_NQUAT, i));
    }
    c->deviceAhead = false;
    return AVBD_OK;
}

int gforces_to_host(AvbdContext* c)
{
    if (!c->gDeviceAhead || c->hG.empty()) { c->gDeviceAhead = false; return AVBD_OK; }
    HIPCHK(c, hipMemcpyAsync(c->hG.data(), c->dG.p, c->hG.size() * sizeof(GForce), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->gDeviceAhead = false;
    return AVBD_OK;
}

int upload_gforces_if_stale(AvbdContext* c)
{
    int rc = ensure_topology_capacity(c);
    if (rc) return rc;
    if (!c->gStale) return AVBD_OK;
    if (!c->hG.empty()) {
        HIPCHK(c, hipMemcpyAsync(c->dG.p, c->hG.data(), c->hG.size() * sizeof(GForce), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->gStale = false;
    c->gDeviceAhead = false;
    c->adjDirty = true;
    return AVBD_OK;
}

/* profiling spans */
enum Family { F_BROAD = 0, F_NARROW, F_TOPO, F_PREDICT, F_PRIMAL, F_DUAL, F_VELOCITY };

hipEvent_t next_event(AvbdContext* c)
{
    if (c->evUsed == c->evPool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        c->evPool.push_back(e);
    }
    return c->evPool[c->evUsed++];
}

struct SpanGuard {
    AvbdContext* c; int fam; size_t e0 = 0;
    bool on;      /* (no event is recorded on the stream while a persistent sharded launch is in flight on it: see shard_launch_persistent) */
    SpanGuard(AvbdContext* ctx, int family) : c(ctx), fam(family), on(ctx->profiling && !ctx->spanPending)
    {
        if (on) { e0 = c->evUsed; (void)hipEventRecord(next_event(c), c->stream); }
    }
    ~SpanGuard()
    {
        if (on) { size_t e1 = c->evUsed; (void)hipEventRecord(next_event(c), c->stream); c->spans.push_back({fam, e0, e1}); }
    }
};

void collect_spans(AvbdContext* c)
{
    if (!c->profiling) return;
    (void)hipStreamSynchronize(c->stream);
    if (c->spanPending) {                                              /* the persistent sharded launch stamps its own span (10 ns ticks) */
        unsigned long long t[2] = {0, 0};
        if (hipMemcpy(t, c->dSpan.p, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess && t[0] && t[1] > t[0]) c->prof.ms_primal += (double)(t[1] - t[0]) * 1e-5;
        c->spanPending = false;
    }
    for (const auto& s : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->evPool[s.e0], c->evPool[s.e1]) != hipSuccess) continue;
        switch (s.family) {
        case F_BROAD: c->prof.ms_broadphase += ms; break;
        case F_NARROW: c->prof.ms_narrowphase += ms; break;
        case F_TOPO: c->prof.ms_topology += ms; break;
        case F_PREDICT: c->prof.ms_predict += ms; break;
        case F_PRIMAL: c->prof.ms_primal += ms; break;
        case F_DUAL: c->prof.ms_dual += ms; break;
        case F_VELOCITY: c->prof.ms_velocity += ms; break;
        }
    }
    c->spans.clear();
    c->evUsed = 0;
}

/* k_count_deg: adjacency degrees of the manifolds already in the current buffer (used when the
 * adjacency must be rebuilt outside a step: after host-side force creation or snapshot upload). */
__global__ void k_count_deg(BodyTable B, ManifoldTable M, const Counters* cnt, int* degForce, int* degInc)
{
    const int nm = cnt->numManifolds;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < nm; m += gridDim.x * blockDim.x) {
        int2 p = M.pair[m]; int nc = M.nc[m];
        atomicAdd(&degForce[p.x], 1); atomicAdd(&degForce[p.y], 1);
        if (B.solved[p.x]) atomicAdd(&degInc[p.x], nc);
        if (B.solved[p.y]) atomicAdd(&degInc[p.y], nc);
    }
}

__global__ void k_copy_count(Counters* cnt, int* nextCount) { *nextCount = cnt->numManifolds; }

int launch_topology(AvbdContext* c, bool fromStep)
{
    c->placeQueued = false;
    BodyTable B = body_table(c);
    const int n = B.n;
    ManifoldTable M = c->mbuf[c->cur].view();
    const int ng = (int)c->hG.size();
    SolverParams P = solver_params(c);
    if (!fromStep) {
        hipLaunchKernelGGL(k_copy_count, dim3(1), dim3(1), 0, c->stream, c->dCnt.p, c->dNext.p);
        hipLaunchKernelGGL(k_count_deg, dim3(grid_for(c->mCap, 256, 1024)), dim3(256), 0, c->stream, B, M, c->dCnt.p, c->degForce.p, c->degInc.p);
    }
    if (ng) hipLaunchKernelGGL(k_gforce_prepare, dim3(grid_for(ng, 128)), dim3(128), 0, c->stream, B, c->dG.p, ng, c->degForce.p, c->degInc.p, P, fromStep ? 1 : 0);
    /* (beyond kTopoEntries adjacency entries the one-workgroup kernel sorts in global memory: 2 744 cubes with 5 200 manifolds take
     * 0.22 ms there against 0.05 ms for the launches below — the last known manifold count decides) */
    if (fromStep && c->fusedTopology && n <= kTopoBodies && c->mode != 0 && 2ll * ((long long)c->stats.numManifolds + ng) <= kTopoEntries) {
        /* small scene: scans, scatter, per-body order, levels, predictor and flag reset as one workgroup in LDS */
        /* (the placement of the one-body-per-wave launch follows a graph that changes slowly: while the same bodies are swept the table of an
         * earlier step stays a valid one — every swept body has its wave — and is rebuilt every placeEvery-th step only: 5.9 of the kernel's 21 us) */
        const int placeW = c->iterBlocks * kIterWaves;
        const bool placing = fromStep && c->placeChains && c->orderByIndex && placeW <= 1024;
        const bool orderKept = c->orderByIndex && c->smallOrderSerial == c->sweptSerial;   /* (the same for the sweep order itself: 1.7 us) */
        bool placeKeep = false;
        if (placing) {
            placeKeep = c->placeSerial == c->sweptSerial && c->placeWaves == placeW && ++c->placeAge < c->placeEvery;
            if (!placeKeep) { c->placeSerial = c->sweptSerial; c->placeWaves = placeW; c->placeAge = 0; }
        }
        hipLaunchKernelGGL(k_topology_small, dim3(1 + (n + 1023) / 1024), dim3(1024), kTopoLdsBytes, c->stream, B, M, c->dG.p, ng, c->dCnt.p, c->dNext.p, c->degForce.p, c->degInc.p,
                           c->adjStart.p, c->incStart.p, c->adjRef.p, c->adjOther.p, c->adjKey.p, c->adjTmp.p, c->inc.p, c->level.p, c->lvlCount.p, c->lvlStart.p,
                           c->order.p, c->lvlCap, c->iterBlocks * kIterWaves, P, c->dFlags.p, 2 * c->bodyCap /* not the error word behind them: a failure stays visible until the next synchronising call */, c->orderByIndex ? 1 : 0,
                           (const int2*)c->dCrit.p, c->dTail.p, c->dPlaceSlot.p, placeW, placing ? c->dPlace.p : nullptr, c->dPlaceInfo.p, placeKeep ? 1 : 0,
                           std::max(1, std::min(c->placeEvery, 16)), orderKept ? c->numSolved : -1);
        c->smallOrderSerial = c->orderByIndex ? c->sweptSerial : ~0u;
        c->placeQueued = placing;
        c->topologyDidPredict = true;
        c->orderValid = false;
        HIPCHK(c, hipGetLastError());
        c->adjDirty = false;
        return AVBD_OK;
    }
    c->topologyDidPredict = false;
    c->smallOrderSerial = ~0u;                                          /* (the kernels below write c->order / level / lvlStart their own way) */
    /* the sweep by creation index only changes with the bodies: computed once, its per-step counters then come from k_scan */
    const bool byIndex = fromStep && c->mode != 0 && c->orderByIndex;
    const bool orderCached = byIndex && c->orderValid;
    hipLaunchKernelGGL(k_scan, dim3(std::max(1, (n + kTileBodies - 1) / kTileBodies)), dim3(1024), 0, c->stream, n, c->degForce.p, c->degInc.p, c->adjStart.p, c->incStart.p, c->dCnt.p, c->dNext.p,
                       c->dTileAgg.p, ++c->tileStamp, orderCached ? c->numSolved : -1, c->iterBlocks * kIterWaves, M.cap);
    /* (one workgroup per CU, grid-stride: the launch is bound by its scattered returning atomics — 380 000 of them at 64 000 cubes, 19 of its 37 us,
     * tools/ubench/fill_adj.hip — and fewer, longer-lived waves get them through faster: 37.6 -> 30.6 us there with 256 workgroups instead of 738) */
    hipLaunchKernelGGL(k_fill_adj, dim3(grid_for((long long)c->stats.numManifolds + c->stats.numCandidates + ng + 4096, 256, 256)), dim3(256), 0, c->stream,
                       M, c->dCnt.p, c->dG.p, ng, c->adjStart.p, c->degForce.p, c->adjRef.p, c->adjOther.p, c->adjKey.p, B.solved);
    hipLaunchKernelGGL(k_sort_expand, dim3(grid_for(n, 128, 1 << 20)), dim3(128), 0, c->stream, B, M, c->dG.p, c->adjStart.p, c->adjRef.p, c->adjOther.p,
                       c->adjKey.p, c->incStart.p, c->inc.p, byIndex ? nullptr : c->level.p, c->degForce.p, c->dCnt.p);
    if (fromStep) {
        if (byIndex) {
            if (!orderCached)
                hipLaunchKernelGGL(k_order_by_index, dim3(1), dim3(1024), 0, c->stream, B, c->level.p, c->lvlStart.p, c->order.p, c->dCnt.p, c->iterBlocks * kIterWaves);
            c->orderValid = true;
        } else {
            c->orderValid = false;
            hipLaunchKernelGGL(k_levels, dim3(1), dim3(1024), kLevelsLdsBytes, c->stream, B, c->adjStart.p, c->adjRef.p, c->adjOther.p, c->level.p, c->lvlCount.p,
                               c->lvlStart.p, c->order.p, c->dCnt.p, c->lvlCap, c->iterBlocks * kIterWaves);
        }
    }
    HIPCHK(c, hipGetLastError());
    c->adjDirty = false;
    return AVBD_OK;
}

/* Copy the device counters (and, for the level mode, the level offsets) to the host and turn device-side
 * failure words into error codes.  Synchronises the stream. */
int refresh_stats(AvbdContext* c, bool withLevels)
{
    HIPCHK(c, hipMemcpyAsync(&c->hRb->cnt, c->dCnt.p, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    if (withLevels)
        HIPCHK(c, hipMemcpyAsync(c->hRb->lvlStart, c->lvlStart.p, sizeof(int) * (std::min(kReadbackLevels, c->lvlCap) + 1), hipMemcpyDeviceToHost, c->stream));
    if (c->groupsLaunched) HIPCHK(c, hipMemcpyAsync(&c->hRb->plan, c->dPlan.p, sizeof(GroupPlan), hipMemcpyDeviceToHost, c->stream));
    if (c->fastLaunched) HIPCHK(c, hipMemcpyAsync(c->hRb->fastHdr, c->fPlan.p, sizeof(c->hRb->fastHdr), hipMemcpyDeviceToHost, c->stream));
    int iterError = 0;
    if (c->dFlags.p)
        HIPCHK(c, hipMemcpyAsync(&iterError, c->dFlags.p + 2 * (size_t)c->bodyCap, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const Counters& cnt = c->hRb->cnt;
    c->stats.numBodies = (int)c->hBodies.size(); c->stats.numManifolds = cnt.numManifolds; c->stats.numContacts = cnt.numContacts;
    c->stats.numIncidences = cnt.numIncidences; c->stats.numCandidates = cnt.numCandidates; c->stats.numLevels = cnt.numLevels;
    c->stats.numDynamic = cnt.dynamicBodies;
    c->stats.schedule = c->mode == 0 ? 0 : ((c->residentLaunched && (cnt.residentBad & 1) == 0 && ((cnt.residentBad & 2) == 0 || c->residentMultiLaunched)) ? (((cnt.residentBad & 2) || c->residentMultiLaunched) ? 3 : 2) : 1);
    if (c->groupsLaunched && c->hRb->plan.bad == 0) c->stats.schedule = 4;   /* group schedule (avbd_groups.h) */
    if (c->fastLaunched && c->hRb->fastHdr[2] == 0) c->stats.schedule = 5;    /* fast sweep (avbd_fast.hip) */
    if (c->fastLaunched && (c->hRb->fastHdr[2] & 256)) {
        int info[8] = {0};
        (void)hipMemcpy(info, (const char*)c->fPlan.p + offsetof(FastPlan, errInfo), sizeof(info), hipMemcpyDeviceToHost);
        char buf[320];
        std::snprintf(buf, sizeof(buf), "k_fast_sweep: a wait timed out (grid not resident?) [tile %d body %d partner %d iteration %d: contact record %08x, visit needed %d; partner epoch %d, needed %d]",
                      info[7], info[0], info[1], info[2], (unsigned)info[3], info[4], info[5], info[6]);
        return c->fail(AVBD_E_HIP, buf);
    }
    int nj = 0, ns = 0, ni = 0;
    for (const auto& g : c->hG) { nj += g.kind == FK_JOINT; ns += g.kind == FK_SPRING; ni += g.kind == FK_IGNORE; }   /* (host forces: stats.numBodies etc. only) */
    c->stats.numJoints = nj; c->stats.numSprings = ns; c->stats.numIgnores = ni;
    c->statsFresh = true;
    if (cnt.overflow & 32) return c->fail(AVBD_E_HIP, "race checker: the schedule of this step is not race-free (two bodies of one launch share a force, or the sweep order is not a linear extension of the dependencies)");
    if (cnt.overflow & 64) return c->fail(AVBD_E_CAPACITY, "narrow phase: a clipped contact polygon had more than eight vertices (the reference's buffers hold sixteen; a quad clipped against four planes cannot have more than eight): the step's manifolds are not the reference's");
    if (cnt.overflow) {
        char buf[320];
        std::snprintf(buf, sizeof(buf), "device pool overflow (mask %d): candidates=%d manifolds=%d cap=%d (option pool_per_body, %d now, sizes the pools when it is set before the bodies are added)", cnt.overflow, cnt.numCandidates, cnt.numManifolds, c->mCap, c->poolPerBody);
        return c->fail(AVBD_E_CAPACITY, buf);
    }
    if (iterError == 2 || iterError == 3) {
        (void)hipMemsetAsync(c->dFlags.p + 2 * (size_t)c->bodyCap, 0, 16 * sizeof(int), c->stream);
        return c->fail(iterError == 2 ? AVBD_E_INVALID : AVBD_E_CAPACITY, iterError == 2 ?
            "exact cuts between processes: the plan of a step did not fit this rank's share of the device (a body with more than 64 contacts, or more than 8 bodies per sub-group): that step was not iterated" :
            "exact cuts: more manifolds across the cuts than the cut tables hold (16 384 per pair of ranks): the records of some cut contacts shared a slot, the results are not the single-GPU sweep's");
    }
    if (iterError) {
        int why[16] = {0};
        (void)hipMemcpy(why, c->dFlags.p + 2 * (size_t)c->bodyCap, sizeof(why), hipMemcpyDeviceToHost);
        (void)hipMemsetAsync(c->dFlags.p + 2 * (size_t)c->bodyCap, 0, sizeof(why), c->stream);   /* reported once */
        char buf[320];
        if (why[1] && why[15] >= 100)   /* several bodies per sub-group: the first body of the first timed-out sub-group that still has tasks */
            std::snprintf(buf, sizeof(buf), "k_iterate: dependency wait timed out (grid not resident?) [rank %d wave %d class %d lane %d: body %d at iteration %d of %d watches record offset %d for stamp base%+d, last saw base%+d; timed out %08x%08x]",
                          c->rank, why[1] - 1, why[15] - 100, why[8], why[2], why[4], why[7], why[3], why[5], why[6], (unsigned)why[14], (unsigned)why[13]);
        else if (why[1] && why[15] == 50)   /* the resident schedule: the first wave to give up */
            std::snprintf(buf, sizeof(buf), "k_iterate: dependency wait timed out (grid not resident?) [resident schedule, rank %d wave %d (workgroup %d of %d, %s): body %d iteration %d partner %d; lanes missing pose %08x%08x visit %08x%08x dual %08x%08x; %d polls]",
                          c->rank, why[1] - 1, why[11], why[12], why[13] ? "several bodies per wave" : "one body per wave", why[2], why[4], why[3], (unsigned)why[6], (unsigned)why[5], (unsigned)why[8], (unsigned)why[7], (unsigned)why[10], (unsigned)why[9], why[14]);
        else if (why[1])   /* the group schedule says what its first wave to give up was waiting for */
            std::snprintf(buf, sizeof(buf), "k_iterate: dependency wait timed out (grid not resident?) [rank %d wave %d class %d: body %d iteration %d partner %d; lanes missing pose %08x%08x visit %08x%08x dual %08x%08x gather %08x%08x, timed out %08x%08x]",
                          c->rank, why[1] - 1, why[15], why[2], why[4], why[3], (unsigned)why[6], (unsigned)why[5], (unsigned)why[8], (unsigned)why[7], (unsigned)why[10], (unsigned)why[9], (unsigned)why[12], (unsigned)why[11], (unsigned)why[14], (unsigned)why[13]);
        else std::snprintf(buf, sizeof(buf), "k_iterate: dependency wait timed out (grid not resident?)");
        return c->fail(AVBD_E_HIP, buf);
    }
    return AVBD_OK;
}

#include "avbd_capi_step.inc"

#include "avbd_capi_shard.inc"

#include "avbd_capi_hostrows.inc"

#include "avbd_capi_exact.inc"

#include "avbd_capi_fast.inc"

int step_once(AvbdContext* c)
{
    int rc;
    if (!c->hostForceG.empty()) return step_with_host_rows(c);
    const bool exactProcesses = c->xwAttached && c->shardExact && c->nranks > 1 && c->mode != 0;
    c->fastLaunched = false;
    if (exactProcesses && (rc = exact_process_precheck(c))) return rc;  /* BEFORE the front end: a refused step leaves the state where it was */
    if ((rc = step_begin(c))) return rc;
    const int totalIt = c->params.iterations + (c->params.postStabilize ? 1 : 0);
    const bool needReadback = (c->mode == 0) || c->profiling;
    if (needReadback) {
        /* level mode: the host launches one kernel per level, so it needs the schedule (one sync per step) */
        if ((rc = refresh_stats(c, true))) return rc;
    } else {
        c->statsFresh = false;
    }
    if (c->nranks > 1 || c->forceSharded) {
        if (c->mode == 0) return c->fail(AVBD_E_INVALID, "sharded runs need the data-flow mode");
        if (exactProcesses) {                                          /* ranks are processes, exact cuts through their windows */
            rc = exact_process_step(c, needReadback);
            /* a failure in here comes after the front end: stepIndex, the manifold tables and the narrow-phase lists are half a step ahead */
            if (rc) c->xwPoisoned = true;
            return rc;
        }
        if (!c->comm) return c->fail(AVBD_E_COMM, "context is sharded (avbd_set_ownership) but has no communicator: call avbd_comm_init or drive it through avbd_group_step");
        bool persistent = false;
        c->shardRanPersistent = false;
        if (!fast_eligible(c) && (rc = shard_plan_accepted(c, persistent))) return rc;   /* (the fast sweep: one launch per iteration, the collective between the launches) */
        c->cutModeRan = persistent ? 2 : 1;
        if (persistent) {
            /* ONE launch for the whole loop; the exchange runs beside it on the second stream: wait until this rank has finished
             * iteration it (every rank does the same), all-gather the rows in place, release the tasks that read them */
            if ((rc = shard_launch_persistent(c))) return rc;
            const unsigned base = c->xSerial * 64u;
            for (int it = 0; it < totalIt; ++it) {
                if (c->shardGateKernel) hipLaunchKernelGGL(k_exchange_gate, dim3(1), dim3(64), 0, c->stream2, c->sigDone, base + (unsigned)it + 1u, c->dFlags.p + 2 * (size_t)c->bodyCap, (float4*)nullptr, (const float4*)nullptr, 0);
                else HIPCHK(c, hipStreamWaitValue32(c->stream2, c->sigDone, base + (unsigned)it + 1u, hipStreamWaitValueGte, 0xFFFFFFFFu));
                if ((rc = rccl_allgather_inplace(c, c->dGather[it & 1].p, c->stream2))) return rc;
                if (c->shardGateKernel) hipLaunchKernelGGL(k_exchange_publish, dim3(1), dim3(1), 0, c->stream2, c->sigGather, base + (unsigned)it + 1u);
                else HIPCHK(c, hipStreamWriteValue32(c->stream2, c->sigGather, base + (unsigned)it + 1u, 0));
            }
            if ((rc = shard_finish_persistent(c, totalIt))) return rc;
            if ((rc = step_end(c, needReadback))) return rc;
            return rccl_exchange_velocities(c);
        }
        for (int it = 0; it < totalIt; ++it) {
            if ((rc = step_iterate_any(c, it, it + 1))) return rc;
            if ((rc = step_pack(c))) return rc;
            if ((rc = rccl_allgather(c))) return rc;
            if ((rc = step_unpack(c))) return rc;
        }
        if ((rc = step_end(c, needReadback))) return rc;
        return rccl_exchange_velocities(c);
    } else if (c->mode == 0) {
        if ((rc = step_iterate_levels(c))) return rc;
    } else if (fast_eligible(c)) {
        if ((rc = step_iterate_fast(c, 0, totalIt))) return rc;
    } else {
        if ((rc = step_iterate(c, 0, totalIt))) return rc;
    }
    return step_end(c, needReadback);
}

void box_mass_properties(AvbdBody& b, float density, float* I9, float* J9)
{
    /* rigid.cpp:24-40 */
    b.mass = b.size[0] * b.size[1] * b.size[2] * density;
    b.invMass = (b.mass > 0.0f) ? 1.0f / b.mass : 0.0f;
    b.radius = sqrtf(b.size[0] * b.size[0] + b.size[1] * b.size[1] + b.size[2] * b.size[2]) * 0.5f;
    for (int k = 0; k < 9; ++k) { I9[k] = 0.0f; J9[k] = 0.0f; }
    if (b.invMass > 0.0f) {
        float Ixx = (1.0f / 12.0f) * b.mass * (b.size[1] * b.size[1] + b.size[2] * b.size[2]);
        float Iyy = (1.0f / 12.0f) * b.mass * (b.size[0] * b.size[0] + b.size[2] * b.size[2]);
        float Izz = (1.0f / 12.0f) * b.mass * (b.size[0] * b.size[0] + b.size[1] * b.size[1]);
        I9[0] = Ixx; I9[4] = Iyy; I9[8] = Izz;
        J9[0] = 1.0f / Ixx; J9[4] = 1.0f / Iyy; J9[8] = 1.0f / Izz;
    }
}

/* host-side quaternion helpers for Joint construction (joint.cpp:17-20, 46-52) — same
 * expression trees as avbd_math.h (that header is host/device) */
f3 H3(const float* p) { return mk3(p[0], p[1], p[2]); }
q4 HQ(const float* p) { return mkq(p[0], p[1], p[2], p[3]); }

} // namespace

static std::string g_createError = "null context";
static void exact_window_release(AvbdContext* c);
static void destroy_comm(void* comm);

/* ========================================================================================== */
extern "C" {

int avbd_abi_version(void) { return AVBD_ABI_VERSION; }

int avbd_default_params(AvbdParams* p)
{
    if (!p) return AVBD_E_INVALID;
    p->dt = 1.0f / 60.0f; p->gravity[0] = 0.0f; p->gravity[1] = -10.0f; p->gravity[2] = 0.0f;
    p->iterations = 10; p->alpha = 0.95f; p->beta = 100000.0f; p->gamma = 0.99f; p->postStabilize = 0;
    return AVBD_OK;
}

/* ---- contexts that share a device take turns ------------------------------------------------------------------------------------
 * The iteration loop of a step is ONE launch that needs every workgroup resident and spins until the others have published; a second
 * launch of that kind beside it — or a topology kernel of another context, whose tiles wait for earlier tiles — can leave both
 * half-resident for ever (two worlds of one process stepped without a synchronize in between: a dependency time-out in five runs
 * out of nine).  avbd_step / avbd_group_step calls of different contexts on one device are therefore ordered on the device with one
 * event per device: a call waits for the call that used the device last (another context's), and records when it has queued its
 * own work.  A process with one context per device never waits and never records. */
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_turnMu;
int g_turnLive[kMaxDevices] = {};              /* contexts alive per device */
hipEvent_t g_turnEv[kMaxDevices] = {};         /* end of the last call that used the device */
AvbdContext* g_turnLast[kMaxDevices] = {};     /* whose call that was */

void turn_begin(AvbdContext* c, AvbdContext* const* group = nullptr, int ngroup = 0)
{
    const int d = c->device;
    if (d < 0 || d >= kMaxDevices) return;
    std::lock_guard<std::mutex> lk(g_turnMu);
    if (!g_turnLast[d] || g_turnLast[d] == c || !g_turnEv[d]) return;
    for (int k = 0; k < ngroup; ++k) if (group[k] == g_turnLast[d]) return;   /* the contexts of one avbd_group_step run side by side by design */
    (void)hipStreamWaitEvent(c->stream, g_turnEv[d], 0);
}
void turn_end(AvbdContext* c)
{
    const int d = c->device;
    if (d < 0 || d >= kMaxDevices) return;
    std::lock_guard<std::mutex> lk(g_turnMu);
    /* (always, also while this is the only context of its device: a context created LATER must find the end of this call to wait for —
     * ADVICE round 3; one event record per call) */
    if (!g_turnEv[d] && hipEventCreateWithFlags(&g_turnEv[d], hipEventDisableTiming) != hipSuccess) { g_turnEv[d] = nullptr; return; }
    (void)hipEventRecord(g_turnEv[d], c->stream);
    g_turnLast[d] = c;
}
} // namespace

#include "avbd_capi_options.inc"

int avbd_create(int device, AvbdContext** out)
{
    if (!out) return AVBD_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return AVBD_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return AVBD_E_NODEVICE;
    AvbdContext* c = new AvbdContext();
    c->device = device;
    avbd_default_params(&c->params);
    if (const char* m = std::getenv("AVBD_MODE")) c->mode = (std::strcmp(m, "levels") == 0 || std::strcmp(m, "0") == 0) ? 0 : 1;   /* (before the allocations below) */
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_createError = "hipStreamCreate failed"; delete c; return AVBD_E_HIP; }
    if (hipHostMalloc((void**)&c->hRb, sizeof(Readback), hipHostMallocDefault) != hipSuccess) { delete c; return AVBD_E_HIP; }
    {   /* second stream + signal words of the persistent sharded step; without them sharded runs launch once per iteration */
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        /* The exchange stream gets the HIGHEST priority: HIP maps the streams of a process onto a few hardware queues (GPU_MAX_HW_QUEUES = 4) PER PRIORITY, and an
         * exchange stream that lands on the queue of a persistent launch — its own context's or another's — sits behind the kernel that waits for that very
         * exchange (the 4 x 16 000 stall of round 3; a 3-in-10 time-out of a test that built its ranks while five other contexts were alive).  With a priority of
         * their own, exchange streams only ever share queues with each other. */
        int prLeast = 0, prGreatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest) != hipSuccess) { (void)hipGetLastError(); prLeast = prGreatest = 0; }
        auto make_stream2 = [&]() {
            if (prGreatest != prLeast && hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prGreatest) == hipSuccess) return true;
            (void)hipGetLastError();
            return hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) == hipSuccess;
        };
        if (can && make_stream2() &&
            hipEventCreateWithFlags(&c->evBegin, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->evGathered, hipEventDisableTiming) == hipSuccess &&
            hipExtMallocWithFlags((void**)&c->sigDone, 8, hipMallocSignalMemory) == hipSuccess && hipExtMallocWithFlags((void**)&c->sigGather, 256, hipDeviceMallocUncached) == hipSuccess) {
            (void)hipMemsetAsync(c->sigDone, 0, 8, c->stream); (void)hipMemsetAsync(c->sigGather, 0, 8, c->stream);
            c->canWaitValue = true;
        } else (void)hipGetLastError();
    }
    std::memset(c->hRb, 0, sizeof(Readback));
    /* the Counters word, then the warning log of the step (kWarnCap records of 32 B) */
    static_assert(sizeof(WarnRec) == 32 && sizeof(Counters) == 64, "warning log layout");
    if (c->dCnt.alloc(1 + kWarnCap * sizeof(WarnRec) / sizeof(Counters)) != hipSuccess || c->dDiag.alloc(1) != hipSuccess || c->dNext.alloc(1) != hipSuccess ||
        c->dPlan.alloc(1) != hipSuccess) { delete c; return AVBD_E_HIP; }
    (void)hipMemsetAsync(c->dPlan.p, 0, sizeof(GroupPlan), c->stream);
    (void)hipMemsetAsync(c->dCnt.p, 0, sizeof(Counters), c->stream);
    (void)hipMemsetAsync(c->dDiag.p, 0, sizeof(DiagAccum), c->stream);
    (void)hipMemsetAsync(c->dNext.p, 0, sizeof(int), c->stream);
    if (ensure_capacity(c, 0) != AVBD_OK || ensure_topology_capacity(c) != AVBD_OK) { g_createError = "initial allocation failed: " + c->err; delete c; return AVBD_E_HIP; }
    {   /* k_iterate needs every workgroup resident.  One workgroup (4 waves = one per SIMD) per CU: a spinning
         * second wave on a SIMD steals issue slots from the wave that is on the critical path.  Padding the LDS
         * request past half a CU's 160 KiB pins that placement, which also makes residency certain
         * (MI355X_MICROARCH.md "Residency and cooperative launch"). */
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_createError = "hipGetDeviceProperties failed"; delete c; return AVBD_E_HIP; }
        c->iterBlocks = std::max(1, prop.multiProcessorCount);
        c->iterLdsPad = kIterWaves * kResProductBytesPerWave;   /* 84 672 B: > half a CU's LDS, and the resident schedule's product lines */
        apply_env_options(c);                                          /* AVBD_<OPTION>: debug overrides of avbd_set_option's keys, read once here */
        c->poolDirty = c->poolPerBody != 16;
        hipError_t fe = hipFuncSetAttribute((const void*)k_iterate<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_single<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_single<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (fe != hipSuccess) { g_createError = std::string("hipFuncSetAttribute(k_iterate, MaxDynamicSharedMemorySize): ") + hipGetErrorString(fe); (void)hipGetLastError(); c->iterLdsPad = 0; }
        fe = hipFuncSetAttribute((const void*)k_iterate<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kIterWaves * kResProductBytesPerWaveLite);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kIterWaves * kResProductBytesPerWaveLite);
        if (fe != hipSuccess) { g_createError = std::string("hipFuncSetAttribute(k_iterate<2>, MaxDynamicSharedMemorySize): ") + hipGetErrorString(fe); delete c; return AVBD_E_HIP; }
        fe = hipFuncSetAttribute((const void*)k_iterate_groups<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytes);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytesMulti);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytes);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytesMulti);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytes);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytesMulti);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups_exact<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytesMulti);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups_exact<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytesMulti);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups<false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytes);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups<true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytesMulti);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups<false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytes);
        if (fe == hipSuccess) fe = hipFuncSetAttribute((const void*)k_iterate_groups<true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrpLdsBytesMulti);
        if (fe != hipSuccess) { g_createError = std::string("hipFuncSetAttribute(k_iterate_groups, MaxDynamicSharedMemorySize): ") + hipGetErrorString(fe); delete c; return AVBD_E_HIP; }
        {   /* Every persistent launch spins on records that other workgroups of the same launch produce: ALL its workgroups must be
             * resident at once.  Ask the runtime how many workgroups of each kernel fit a CU with the exact dynamic LDS it is
             * launched with, clamp the grid sizes (and any AVBD_ITER_BLOCKS override) to that, and switch a schedule off when
             * its prerequisite does not hold on this device / this build (compiler register use, partitioned device). */
            const int cus = std::max(1, prop.multiProcessorCount);
            auto fits = [&](const void* fn, int dynLds) -> int {
                int perCu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, fn, 64 * kIterWaves, (size_t)dynLds) != hipSuccess) { (void)hipGetLastError(); return 0; }
                return perCu;
            };
            const int occ1 = std::min(std::min(fits((const void*)k_iterate<1, false>, c->iterLdsPad), fits((const void*)k_iterate<1, true>, c->iterLdsPad)),
                                      std::min(fits((const void*)k_iterate_single<false>, c->iterLdsPad), fits((const void*)k_iterate_single<true>, c->iterLdsPad)));
            const int occ2 = std::min(fits((const void*)k_iterate<2, false>, kIterWaves * kResProductBytesPerWaveLite), fits((const void*)k_iterate<2, true>, kIterWaves * kResProductBytesPerWaveLite));
            const int occG = std::min(std::min(std::min(fits((const void*)k_iterate_groups<false, 0>, kGrpLdsBytes), fits((const void*)k_iterate_groups<true, 0>, kGrpLdsBytesMulti)),
                                               std::min(fits((const void*)k_iterate_groups<false, 1>, kGrpLdsBytes), fits((const void*)k_iterate_groups<true, 1>, kGrpLdsBytesMulti))),
                                      std::min(fits((const void*)k_iterate_groups<false, 0, true>, kGrpLdsBytes), fits((const void*)k_iterate_groups<true, 0, true>, kGrpLdsBytesMulti)));
            /* the exact-cut forms spin on records of other ranks' launches as well: the same residency prerequisite */
            const int occX = std::min(std::min(std::min(fits((const void*)k_iterate_groups<false, 2>, kGrpLdsBytes), fits((const void*)k_iterate_groups<true, 2>, kGrpLdsBytesMulti)),
                                               std::min(fits((const void*)k_iterate_groups<false, 2, true>, kGrpLdsBytes), fits((const void*)k_iterate_groups<true, 2, true>, kGrpLdsBytesMulti))),
                                      std::min(fits((const void*)k_iterate_groups_exact<false>, kGrpLdsBytesMulti), fits((const void*)k_iterate_groups_exact<true>, kGrpLdsBytesMulti)));
            c->exactAllowed = occX * cus >= 2 * std::min(c->iterBlocks, std::max(1, occ1) * cus);
            c->capDataflow = occ1 >= 1;                                      /* not even one workgroup per CU: one launch per level, nothing spins */
            c->iterBlocks = std::min(c->iterBlocks, std::max(1, occ1) * cus);
            c->capLite = occ2 * cus >= 2 * c->iterBlocks;
            c->capGroups = occG * cus >= 2 * c->iterBlocks;
            c->iterBlocksMax = std::max(1, occ1) * cus;
            apply_capabilities(c);
        }
        c->numCU = std::max(1, prop.multiProcessorCount);
        c->capFast = fast_query(device, &c->fastPlanPerCU, &c->fastSweepPerCU) != 0;
        (void)hipGetLastError();
        fe = hipFuncSetAttribute((const void*)k_topology_small, hipFuncAttributeMaxDynamicSharedMemorySize, kTopoLdsBytes);
        if (fe != hipSuccess) { g_createError = std::string("hipFuncSetAttribute(k_topology_small, MaxDynamicSharedMemorySize): ") + hipGetErrorString(fe); delete c; return AVBD_E_HIP; }
        fe = hipFuncSetAttribute((const void*)k_levels, hipFuncAttributeMaxDynamicSharedMemorySize, kLevelsLdsBytes);
        if (fe != hipSuccess) { g_createError = std::string("hipFuncSetAttribute(k_levels, MaxDynamicSharedMemorySize): ") + hipGetErrorString(fe); delete c; return AVBD_E_HIP; }
    }
    if (device < kMaxDevices) { std::lock_guard<std::mutex> lk(g_turnMu); c->liveAtCreation = ++g_turnLive[device]; }
    *out = c;
    return AVBD_OK;
}

int avbd_destroy(AvbdContext* c)
{
    if (!c) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->device >= 0 && c->device < kMaxDevices) {
        std::lock_guard<std::mutex> lk(g_turnMu);
        if (!c->turnReleased) { --g_turnLive[c->device]; c->turnReleased = true; }
        if (g_turnLast[c->device] == c) g_turnLast[c->device] = nullptr;      /* (its work has finished: nothing to wait for) */
    }
    for (auto e : c->evPool) (void)hipEventDestroy(e);
    c->dPlanes.release(); c->mbuf[0].release(); c->mbuf[1].release(); c->dG.release();
    c->adjStart.release(); c->incStart.release(); c->degForce.release(); c->degInc.release(); c->level.release(); c->order.release();
    c->lvlStart.release(); c->lvlCount.release(); c->adjRef.release(); c->adjOther.release(); c->inc.release(); c->adjKey.release();
    c->cand.release(); c->dNarrowPerm[0].release(); c->dNarrowPerm[1].release(); c->dNarrowCnt.release(); c->dCnt.release(); c->dDiag.release(); c->dNext.release(); c->dFlags.release();
    c->dSolved.release(); c->dPerm.release(); c->dRankStart.release(); c->dSendSlot.release(); c->dSend.release(); c->dRecv.release(); c->dXchMismatch.release();
    if (c->iterDone) { (void)hipEventDestroy(c->iterDone); c->iterDone = nullptr; }
    c->dIsBig.release(); c->dBigList.release(); c->dCellHead.release(); c->dCellNext.release();
    c->dPlan.release(); c->dUnit.release(); c->dSlotOf.release(); c->dHostRows.release();
    c->dDone.release(); c->dGather[0].release(); c->dGather[1].release(); c->dRow.release(); c->dSpan.release();
    if (c->sigDone) (void)hipFree(c->sigDone); if (c->sigGather) (void)hipFree(c->sigGather);
    if (c->evBegin) (void)hipEventDestroy(c->evBegin); if (c->evGathered) (void)hipEventDestroy(c->evGathered);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    c->xPose.release(); c->xVisit.release(); c->xDual.release(); c->xPrivN.release(); c->xPrivPose.release(); c->xProbe.release(); c->adjTmp.release();
    c->fPlan.release(); c->fColor[0].release(); c->fColor[1].release(); c->fCbody.release(); c->fTiles.release(); c->fTileMask.release(); c->fSlotC.release(); c->fBodyC.release();
    c->fXpose.release(); c->fXS.release(); c->fSync.release();
    exact_window_release(c);
    if (c->comm) destroy_comm(c->comm);
    if (c->hRb) (void)hipHostFree(c->hRb);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return AVBD_OK;
}

const char* avbd_last_error(const AvbdContext* c) { return c ? c->err.c_str() : g_createError.c_str(); }

int avbd_set_params(AvbdContext* c, const AvbdParams* p)
{
    if (!c || !p) return AVBD_E_INVALID;
    c->params = *p;
    return AVBD_OK;
}

int avbd_get_params(const AvbdContext* c, AvbdParams* p)
{
    if (!c || !p) return AVBD_E_INVALID;
    *p = c->params;
    return AVBD_OK;
}

int avbd_clear(AvbdContext* c)
{
    if (!c) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->hBodies.clear(); c->hI.clear(); c->hJ.clear(); c->hG.clear(); c->hostForceG.clear(); c->hHostRows.clear();
    c->bodiesStale = true; c->deviceAhead = false; c->gStale = true; c->gDeviceAhead = false; c->adjDirty = true;
    c->stepIndex = 0; c->stepsSinceLoad = 0; c->hostKeyCounter = 1; c->numStatic = 0;
    c->stats = AvbdStepStats{};
    HIPCHK(c, hipMemsetAsync(c->dCnt.p, 0, sizeof(Counters), c->stream));
    HIPCHK(c, hipMemsetAsync(c->dDiag.p, 0, sizeof(DiagAccum), c->stream));
    return AVBD_OK;
}

int avbd_add_bodies(AvbdContext* c, int n, const float* size, const float* density, const float* friction, const float* position,
                    const float* orientation, const float* linearVelocity, const float* angularVelocity, int* first_index)
{
    if (!c || n < 0 || (n && (!size || !density || !friction || !position))) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = download_bodies_to_host(c);
    if (rc) return rc;
    const int first = (int)c->hBodies.size();
    if (first + n > (1 << 20) - 2) return c->fail(AVBD_E_INVALID, "more than 2^20-2 bodies");
    for (int i = 0; i < n; ++i) {
        AvbdBody b;
        std::memset(&b, 0, sizeof(b));
        for (int k = 0; k < 3; ++k) {
            b.size[k] = size[3 * i + k]; b.position[k] = position[3 * i + k];
            b.linearVelocity[k] = linearVelocity ? linearVelocity[3 * i + k] : 0.0f;
            b.angularVelocity[k] = angularVelocity ? angularVelocity[3 * i + k] : 0.0f;
            b.prevLinearVelocity[k] = b.linearVelocity[k]; b.prevAngularVelocity[k] = b.angularVelocity[k];
        }
        for (int k = 0; k < 4; ++k) b.orientation[k] = orientation ? orientation[4 * i + k] : (k == 3 ? 1.0f : 0.0f);
        b.initialOrientation[3] = 1.0f; b.inertialOrientation[3] = 1.0f;
        b.friction = friction[i];
        float I9[9], J9[9];
        box_mass_properties(b, density[i], I9, J9);
        if (!(b.invMass > 0.0f)) ++c->numStatic;
        c->hBodies.push_back(b);
        c->hI.insert(c->hI.end(), I9, I9 + 9);
        c->hJ.insert(c->hJ.end(), J9, J9 + 9);
    }
    if (first_index) *first_index = first;
    c->bodiesStale = true;
    c->adjDirty = true;
    return AVBD_OK;
}

static int add_gforce(AvbdContext* c, const GForce& g)
{
    int rc = gforces_to_host(c);
    if (rc) return rc;
    c->hG.push_back(g);
    c->hG.back().key = make_host_key(c->stepIndex, c->hostKeyCounter++);
    c->gStale = true;
    c->adjDirty = true;
    return AVBD_OK;
}

static void init_gforce(GForce& g, int kind, int a, int b)
{
    std::memset(&g, 0, sizeof(g));
    g.kind = kind; g.bodyA = a; g.bodyB = b;
    g.q0[3] = 1.0f;
    for (int i = 0; i < 6; ++i) { g.fmin[i] = -FLT_MAX; g.fmax[i] = FLT_MAX; } /* force.cpp:32-40 */
}

int avbd_add_joint(AvbdContext* c, int bodyA, int bodyB, const float rA[3], const float rB[3], float linearStiffness, float angularStiffness)
{
    if (!c || !rA) return AVBD_E_INVALID;
    const int n = (int)c->hBodies.size();
    if (bodyB < 0 || bodyB >= n || bodyA >= n || bodyA < -1) return c->fail(AVBD_E_INVALID, "joint body index out of range");
    (void)hipSetDevice(c->device);
    int rc = download_bodies_to_host(c);
    if (rc) return rc;
    GForce g; init_gforce(g, FK_JOINT, bodyA, bodyB);
    const AvbdBody& B = c->hBodies[bodyB];
    if (bodyA >= 0) {               /* joint.cpp:11-39 */
        if (!rB) return AVBD_E_INVALID;
        for (int k = 0; k < 3; ++k) { g.rA[k] = rA[k]; g.rB[k] = rB[k]; }
        q4 q = qconj(HQ(c->hBodies[bodyA].orientation)) * HQ(B.orientation);
        g.q0[0] = q.x; g.q0[1] = q.y; g.q0[2] = q.z; g.q0[3] = q.w;
    } else {                        /* joint.cpp:42-61 */
        for (int k = 0; k < 3; ++k) g.rA[k] = rA[k];
        f3 r = mtranspose(quat_to_mat(HQ(B.orientation))) * (H3(rA) - H3(B.position));
        g.rB[0] = r.x; g.rB[1] = r.y; g.rB[2] = r.z;
        for (int k = 0; k < 4; ++k) g.q0[k] = B.orientation[k];
    }
    for (int i = 0; i < 3; ++i) { g.stiffness[i] = linearStiffness; g.lambda[i] = 0; g.penalty[i] = kPenaltyMin; }
    for (int i = 3; i < 6; ++i) { g.stiffness[i] = angularStiffness; g.lambda[i] = 0; g.penalty[i] = kPenaltyMin; }
    g.pad = c->jointAngularFix ? 1 : 0;
    return add_gforce(c, g);
}

int avbd_set_joint_angular_sign(AvbdContext* c, int corrected)
{
    if (!c) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = gforces_to_host(c);                                      /* the rows the device has advanced */
    if (rc) return rc;
    c->jointAngularFix = corrected != 0;
    for (GForce& g : c->hG) if (g.kind == FK_JOINT) g.pad = c->jointAngularFix ? 1 : 0;
    c->gStale = true;
    return AVBD_OK;
}

int avbd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int avbd_set_shard_mode(AvbdContext* c, int mode)
{
    if (!c || mode < 0 || mode > 2) return AVBD_E_INVALID;
    c->shardExact = mode != 0;
    c->shardExactStrict = mode != 2;
    return AVBD_OK;
}

int avbd_set_option(AvbdContext* c, const char* key, int value)
{
    if (!c || !key) return AVBD_E_INVALID;
    int v = value;
    if (!std::strcmp(key, "narrow_lanes") && v != 0 && v != 32 && v != 64) return c->fail(AVBD_E_INVALID, "avbd_set_option: 'narrow_lanes' is 0 (by the body count), 32 or 64");
    if (!std::strcmp(key, "iter_blocks")) v = std::max(1, std::min(v, c->iterBlocksMax > 0 ? c->iterBlocksMax : v));   /* every workgroup of the loop must be resident */
    if (!std::strcmp(key, "pool_per_body")) {
        /* (validated BEFORE anything is touched, and a value the pools already cover changes nothing — they never shrink: ADVICE round 5) */
        if (v < 4 || v > 4096) return c->fail(AVBD_E_INVALID, "avbd_set_option: 'pool_per_body' is 4 .. 4096");
        if (v > c->poolPerBody) {
            /* the pools are re-allocated with the body tables at the next step: the host copy of the bodies becomes the authoritative one first (what adding a body does) */
            (void)hipSetDevice(c->device);
            if (int rc = download_bodies_to_host(c)) return rc;
            c->bodiesStale = true; c->poolDirty = true;
        }
    }
    if (!std::strcmp(key, "mode") && c->stepIndex != 0 && v != c->mode) return c->fail(AVBD_E_INVALID, "avbd_set_option: 'mode' is chosen before the first step");
    const int rc = option_access(c, key, &v, true);
    if (rc) return c->fail(rc, "avbd_set_option: unknown key");
    apply_capabilities(c);                                             /* a schedule whose workgroups cannot all be resident here stays off whatever was asked for */
    return AVBD_OK;
}
int avbd_get_option(AvbdContext* c, const char* key, int* value)
{
    if (!c || !key || !value) return AVBD_E_INVALID;
    return option_access(c, key, value, false);
}
int avbd_option_count(void) { return (int)(sizeof(kOptionKeys) / sizeof(kOptionKeys[0])); }
const char* avbd_option_key(int i) { return (i >= 0 && i < avbd_option_count()) ? kOptionKeys[i] : nullptr; }

/* ---- exact cuts between ranks that are processes ---- */
static void exact_window_detach(AvbdContext* c)
{
    for (int q = 0; q < kMaxExactRanks; ++q) {
        if (c->xwPeer[q] && c->xwPeer[q] != c->xwOwn) (void)hipIpcCloseMemHandle(c->xwPeer[q]);
        c->xwPeer[q] = nullptr;
    }
    c->xwAttached = false;
}
static void exact_window_release(AvbdContext* c)
{
    exact_window_detach(c);
    if (c->xwOwn) { (void)hipFree(c->xwOwn); c->xwOwn = nullptr; }
}

int avbd_exact_window_export(AvbdContext* c, unsigned char handle[64], unsigned long long* bytes)
{
    if (!c || !handle) return AVBD_E_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI carries a hipIpcMemHandle_t as 64 opaque bytes");
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(AVBD_E_NODEVICE, "hipSetDevice failed");
    if (c->nranks < 2 || c->nranks > kMaxExactRanks) return c->fail(AVBD_E_INVALID, "avbd_exact_window_export: set the ownership first (avbd_set_ownership, 2..8 ranks)");
    int rc;
    if ((rc = upload_bodies_if_stale(c))) return rc;
    if ((rc = upload_gforces_if_stale(c))) return rc;
    const size_t slots = exact_window_slots(c, c->nranks), f4 = exact_window_float4(c, c->nranks);
    if (slots * 32 >= (1ull << 31) || (size_t)3 * c->bodyCap * 16 >= (1ull << 31)) return c->fail(AVBD_E_INVALID, "avbd_exact_window_export: the window exceeds 32-bit record offsets");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    exact_window_release(c);
    /* uncached device memory: the peers may sit on other devices (tools/ubench/ipc_pingpong.hip: a hop through it costs 0.58 us against 0.53 us
     * through plain memory; every access to a record is a system-scope one anyway) */
    HIPCHK(c, hipExtMallocWithFlags((void**)&c->xwOwn, f4 * sizeof(float4), hipDeviceMallocUncached));
    HIPCHK(c, hipMemsetAsync(c->xwOwn, 0, f4 * sizeof(float4), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipIpcMemHandle_t h;
    HIPCHK(c, hipIpcGetMemHandle(&h, c->xwOwn));
    std::memcpy(handle, &h, 64);
    if (bytes) *bytes = (unsigned long long)(f4 * sizeof(float4));
    c->xwBodyCap = c->bodyCap; c->xwMCap = c->mCap; c->xwGCap = c->gCap; c->xwRanks = c->nranks; c->xwSerial = 0;
    return AVBD_OK;
}

int avbd_exact_window_attach(AvbdContext* c, int nranks, const unsigned char* handles, int ranksOnThisDevice)
{
    if (!c || !handles || ranksOnThisDevice < 1) return AVBD_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(AVBD_E_NODEVICE, "hipSetDevice failed");
    if (!c->xwOwn || nranks != c->nranks || nranks != c->xwRanks) return c->fail(AVBD_E_INVALID, "avbd_exact_window_attach: export this rank's window first (same number of ranks)");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    exact_window_detach(c);                                            /* a second attach without a new export: the old mappings are closed, not overwritten */
    for (int q = 0; q < nranks; ++q) {
        if (q == c->rank) { c->xwPeer[q] = c->xwOwn; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)q * 64, 64);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { (void)hipGetLastError(); exact_window_detach(c); return c->fail(AVBD_E_HIP, "hipIpcOpenMemHandle (a peer's window; windows of the same process cannot be opened — use avbd_group_step there)", e); }
        c->xwPeer[q] = (float4*)p;
    }
    /* ranks that share a device share its resident waves: every rank's loop kernel must be resident while the others' are */
    const int all = 2 * c->iterBlocks * kGrpWavesPerBlock;
    c->shardWaves = ranksOnThisDevice > 1 ? std::max(kGrpWavesPerBlock, (all / ranksOnThisDevice) / kGrpWavesPerBlock * kGrpWavesPerBlock) : 0;
    c->shardExact = true; c->shardExactStrict = true;
    c->xwSharedDevice = ranksOnThisDevice > 1;
    c->xwAttached = true;
    c->xwPoisoned = false;
    return AVBD_OK;
}

int avbd_set_exchange_floats(AvbdContext* c, int floatsPerBody)
{
    if (!c || (floatsPerBody != 8 && floatsPerBody != 13)) return AVBD_E_INVALID;
    c->exchangeFloats = floatsPerBody;
    return AVBD_OK;
}

int avbd_exchange_mismatches(AvbdContext* c, unsigned* words, unsigned* rowsCompared)
{
    if (!c || !words) return AVBD_E_INVALID;
    unsigned h[2] = {0, 0};
    if (c->dXchMismatch.p) {
        (void)hipSetDevice(c->device);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(h, c->dXchMismatch.p, sizeof(h), hipMemcpyDeviceToHost));
    }
    *words = h[0];
    if (rowsCompared) *rowsCompared = h[1];
    return AVBD_OK;
}

int avbd_add_spring(AvbdContext* c, int bodyA, int bodyB, const float rA[3], const float rB[3], float stiffness, float rest)
{
    if (!c || !rA || !rB) return AVBD_E_INVALID;
    const int n = (int)c->hBodies.size();
    if (bodyA < 0 || bodyA >= n || bodyB < 0 || bodyB >= n) return c->fail(AVBD_E_INVALID, "spring body index out of range");
    (void)hipSetDevice(c->device);
    int rc = download_bodies_to_host(c);
    if (rc) return rc;
    GForce g; init_gforce(g, FK_SPRING, bodyA, bodyB);
    for (int k = 0; k < 3; ++k) { g.rA[k] = rA[k]; g.rB[k] = rB[k]; }
    g.restLength = rest;
    g.stiffness[0] = stiffness;     /* spring.cpp:16 */
    if (g.restLength < 0) {         /* spring.cpp:19-23 */
        const AvbdBody& A = c->hBodies[bodyA]; const AvbdBody& B = c->hBodies[bodyB];
        f3 pA = H3(A.position) + qrotate(HQ(A.orientation), H3(rA));
        f3 pB = H3(B.position) + qrotate(HQ(B.orientation), H3(rB));
        g.restLength = len3(pA - pB);
    }
    g.lambda[0] = 0.0f; g.penalty[0] = kPenaltyMin;
    return add_gforce(c, g);
}

int avbd_add_ignore(AvbdContext* c, int bodyA, int bodyB)
{
    if (!c) return AVBD_E_INVALID;
    const int n = (int)c->hBodies.size();
    if (bodyA < 0 || bodyA >= n || bodyB < 0 || bodyB >= n) return c->fail(AVBD_E_INVALID, "ignore body index out of range");
    GForce g; init_gforce(g, FK_IGNORE, bodyA, bodyB);
    return add_gforce(c, g);
}

int avbd_add_host_force(AvbdContext* c, int bodyA, int bodyB, int* host_force)
{
    if (!c || !host_force) return AVBD_E_INVALID;
    const int n = (int)c->hBodies.size();
    if (bodyB < 0 || bodyB >= n || bodyA >= n || bodyA < -1) return c->fail(AVBD_E_INVALID, "host force body index out of range");
    (void)hipSetDevice(c->device);
    GForce g; init_gforce(g, FK_HOST, bodyA, bodyB);
    const int ordinal = (int)c->hostForceG.size();
    g.pad = ordinal;                                                   /* rows << 24 is filled in at every step begin */
    int rc = add_gforce(c, g);
    if (rc) return rc;
    c->hostForceG.push_back((int)c->hG.size() - 1);
    HostRowsDev h; std::memset(&h, 0, sizeof(h));
    c->hHostRows.push_back(h);
    *host_force = ordinal;
    return AVBD_OK;
}

int avbd_set_host_force_callback(AvbdContext* c, AvbdHostForceFn fn, void* user)
{
    if (!c) return AVBD_E_INVALID;
    c->hostFn = fn; c->hostUser = user;
    return AVBD_OK;
}

int avbd_add_force(AvbdContext* c, const AvbdForce* r)
{
    if (!c || !r) return AVBD_E_INVALID;
    const int n = (int)c->hBodies.size();
    if (r->kind != AVBD_FORCE_JOINT && r->kind != AVBD_FORCE_SPRING && r->kind != AVBD_FORCE_IGNORE)
        return c->fail(AVBD_E_INVALID, "avbd_add_force: only Joint / Spring / IgnoreCollision can be host-created");
    if (r->bodyB < 0 || r->bodyB >= n || r->bodyA >= n || r->bodyA < -1) return c->fail(AVBD_E_INVALID, "force body index out of range");
    (void)hipSetDevice(c->device);
    GForce g; init_gforce(g, r->kind, r->bodyA, r->bodyB);
    for (int i = 0; i < 6; ++i) {
        g.C[i] = r->C[i]; g.fmin[i] = r->fmin[i]; g.fmax[i] = r->fmax[i]; g.lambda[i] = r->lambda[i];
        g.penalty[i] = r->penalty[i]; g.stiffness[i] = r->stiffness[i]; g.motor[i] = r->motor[i];
    }
    for (int k = 0; k < 3; ++k) { g.rA[k] = r->rA[k]; g.rB[k] = r->rB[k]; }
    for (int k = 0; k < 4; ++k) g.q0[k] = r->q0[k];
    g.restLength = r->restLength;
    return add_gforce(c, g);
}

/* ---- snapshots: the complete solver state as one little-endian file --------------------------------
 * header { magic "AVBDSNP1", sizeof(AvbdBody), sizeof(AvbdForce), nBodies, nForces, stepIndex, AvbdParams } then the
 * AvbdBody and AvbdForce arrays exactly as avbd_download_bodies / avbd_download_forces return them (force list newest
 * first).  A context that loads a snapshot continues bit for bit like the one that saved it. */
namespace {
struct SnapshotHeader { char magic[8]; int bodyBytes, forceBytes, nBodies, nForces, stepIndex, reserved; AvbdParams params; };
}

#include "avbd_capi_state.inc"

int avbd_step(AvbdContext* c, int nsteps)
{
    if (!c || nsteps < 0) return AVBD_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(AVBD_E_NODEVICE, "hipSetDevice failed");
    turn_begin(c);
    int rc = AVBD_OK;
    for (int s = 0; s < nsteps && !rc; ++s) rc = step_once(c);
    turn_end(c);
    return rc;
}

int avbd_synchronize(AvbdContext* c)
{
    if (!c) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    if (!c->statsFresh) return refresh_stats(c, false);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AVBD_OK;
}

int avbd_step_index(const AvbdContext* c) { return c ? c->stepIndex : AVBD_E_INVALID; }

int avbd_get_diagnostics(AvbdContext* c, AvbdDiagnostics* out)
{
    if (!c || !out) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    DiagAccum d;
    HIPCHK(c, hipMemcpyAsync(&d, c->dDiag.p, sizeof(d), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto f = [](unsigned u) { float v; std::memcpy(&v, &u, 4); return v; };
    out->maxPenetration = f(d.maxPenetration); out->maxConstraintViolation = f(d.maxViolation);
    out->maxLinearSpeed = f(d.maxLinearSpeed); out->maxAngularSpeed = f(d.maxAngularSpeed); out->maxNormalImpulse = f(d.maxNormalImpulse);
    out->activeContacts = d.activeContacts; out->activeManifolds = d.activeManifolds; out->dynamicBodies = d.dynamicBodies;
    return AVBD_OK;
}

int avbd_get_warnings(AvbdContext* c, AvbdWarning* out, int capacity)
{
    if (!c || capacity < 0 || (capacity && !out)) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    Counters cnt;
    HIPCHK(c, hipMemcpy(&cnt, c->dCnt.p, sizeof(Counters), hipMemcpyDeviceToHost));
    const int have = std::min(cnt.warnings, kWarnCap);
    if (have <= 0) return 0;
    std::vector<WarnRec> log((size_t)have);
    HIPCHK(c, hipMemcpy(log.data(), c->dCnt.p + 1, sizeof(WarnRec) * (size_t)have, hipMemcpyDeviceToHost));
    std::sort(log.begin(), log.end(), [](const WarnRec& a, const WarnRec& b) { return a.key < b.key; });
    const int n = std::min(have, capacity);
    for (int i = 0; i < n; ++i) { out[i].body = log[(size_t)i].body; out[i].kind = log[(size_t)i].kind; out[i].value[0] = log[(size_t)i].x; out[i].value[1] = log[(size_t)i].y; out[i].value[2] = log[(size_t)i].z; }
    return cnt.warnings;
}

int avbd_get_step_stats(AvbdContext* c, AvbdStepStats* out)
{
    if (!c || !out) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    if (!c->statsFresh) { int rc = refresh_stats(c, false); if (rc) return rc; }
    *out = c->stats;
    out->warnings = c->hRb ? c->hRb->cnt.warnings : 0;
    out->cutMode = c->nranks > 1 ? c->cutModeRan : 0;
    return AVBD_OK;
}

int avbd_num_bodies(const AvbdContext* c) { return c ? (int)c->hBodies.size() : AVBD_E_INVALID; }

int avbd_download_bodies(AvbdContext* c, AvbdBody* out, int capacity)
{
    if (!c || (!out && capacity > 0)) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = download_bodies_to_host(c);
    if (rc) return rc;
    const int n = std::min(capacity, (int)c->hBodies.size());
    if (n > 0) std::memcpy(out, c->hBodies.data(), (size_t)n * sizeof(AvbdBody));
    return n;
}

int avbd_download_state13(AvbdContext* c, float* out, int capacity)
{
    if (!c || (!out && capacity > 0)) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = download_bodies_to_host(c);
    if (rc) return rc;
    const int n = std::min(capacity, (int)c->hBodies.size());
    for (int i = 0; i < n; ++i) {
        const AvbdBody& b = c->hBodies[i];
        float* o = out + 13 * (size_t)i;
        std::memcpy(o, b.position, 12); std::memcpy(o + 3, b.orientation, 16);
        std::memcpy(o + 7, b.linearVelocity, 12); std::memcpy(o + 10, b.angularVelocity, 12);
    }
    return n;
}

int avbd_upload_state13(AvbdContext* c, const float* in, int n)
{
    if (!c || !in || n < 0 || n > (int)c->hBodies.size()) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = download_bodies_to_host(c);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        AvbdBody& b = c->hBodies[i];
        const float* o = in + 13 * (size_t)i;
        std::memcpy(b.position, o, 12); std::memcpy(b.orientation, o + 3, 16);
        std::memcpy(b.linearVelocity, o + 7, 12); std::memcpy(b.angularVelocity, o + 10, 12);
    }
    c->bodiesStale = true;
    return AVBD_OK;
}

int avbd_upload_bodies(AvbdContext* c, const AvbdBody* in, int n)
{
    if (!c || !in || n < 0 || n > (int)c->hBodies.size()) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = download_bodies_to_host(c);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        AvbdBody& b = c->hBodies[i];
        const AvbdBody& s = in[i];
        std::memcpy(b.position, s.position, sizeof(float) * 33); /* all dynamic members; static ones keep the ctor values */
    }
    c->bodiesStale = true;
    return AVBD_OK;
}

/* ---- forces --------------------------------------------------------------------------- */
namespace {
struct HostManifolds {
    int n = 0;
    std::vector<int2> pair; std::vector<uint64_t> key; std::vector<int> nc; std::vector<float> fric;
    std::vector<float4> cRA, cRB, cN, cLam, cPen, cC, cFmin, cFmax; std::vector<int2> cFS;
};

int fetch_manifolds(AvbdContext* c, HostManifolds& h)
{
    Counters cnt;
    HIPCHK(c, hipMemcpyAsync(&cnt, c->dCnt.p, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n = cnt.numManifolds;
    h.n = n;
    if (!n) return AVBD_OK;
    ManifoldBuf& b = c->mbuf[c->cur];
    h.pair.resize(n); h.key.resize(n); h.nc.resize(n); h.fric.resize(n);
    size_t k = (size_t)n * 4;
    h.cRA.resize(k); h.cRB.resize(k); h.cN.resize(k); h.cLam.resize(k); h.cPen.resize(k); h.cC.resize(k); h.cFmin.resize(k); h.cFmax.resize(k); h.cFS.resize(k);
    HIPCHK(c, hipMemcpy(h.pair.data(), b.pair.p, n * sizeof(int2), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.key.data(), b.key.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.nc.data(), b.nc.p, n * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.fric.data(), b.fric.p, n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.cRA.data(), b.cRA.p, k * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.cRB.data(), b.cRB.p, k * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.cN.data(), b.cN.p, k * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.cLam.data(), b.cLam.p, k * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.cPen.data(), b.cPen.p, k * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.cC.data(), b.cC.p, k * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.cFmin.data(), b.cFmin.p, k * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.cFmax.data(), b.cFmax.p, k * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h.cFS.data(), b.cFS.p, k * sizeof(int2), hipMemcpyDeviceToHost));
    return AVBD_OK;
}

struct ForceRef { uint64_t key; int kind; int index; };

int sorted_force_refs(AvbdContext* c, HostManifolds& hm, std::vector<ForceRef>& refs)
{
    int rc = fetch_manifolds(c, hm);
    if (rc) return rc;
    if ((rc = gforces_to_host(c))) return rc;
    refs.clear();
    for (int m = 0; m < hm.n; ++m) refs.push_back({hm.key[m], FK_MANIFOLD, m});
    for (size_t g = 0; g < c->hG.size(); ++g) refs.push_back({c->hG[g].key, c->hG[g].kind, (int)g});
    std::sort(refs.begin(), refs.end(), [](const ForceRef& a, const ForceRef& b) { return a.key > b.key; });
    return AVBD_OK;
}
} // namespace

int avbd_num_forces(AvbdContext* c)
{
    if (!c) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    Counters cnt;
    HIPCHK(c, hipMemcpyAsync(&cnt, c->dCnt.p, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return cnt.numManifolds + (int)c->hG.size();
}

int avbd_download_forces(AvbdContext* c, AvbdForce* out, int capacity)
{
    if (!c || (!out && capacity > 0)) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    HostManifolds hm; std::vector<ForceRef> refs;
    int rc = sorted_force_refs(c, hm, refs);
    if (rc) return rc;
    const int n = std::min(capacity, (int)refs.size());
    for (int i = 0; i < n; ++i) {
        AvbdForce& o = out[i];
        std::memset(&o, 0, sizeof(o));
        for (int r = 0; r < 12; ++r) { o.fmin[r] = -FLT_MAX; o.fmax[r] = FLT_MAX; }
        o.kind = refs[i].kind;
        if (refs[i].kind == FK_MANIFOLD) {
            const int m = refs[i].index;
            o.bodyA = hm.pair[m].x; o.bodyB = hm.pair[m].y; o.numContacts = hm.nc[m]; o.rows = 3 * hm.nc[m];
            o.combinedFriction = hm.fric[m];
            for (int ci = 0; ci < hm.nc[m]; ++ci) {
                const size_t s = (size_t)m * 4 + ci;
                AvbdContact& d = o.contacts[ci];
                d.feature = hm.cFS[s].x; d.stick = hm.cFS[s].y;
                d.rA[0] = hm.cRA[s].x; d.rA[1] = hm.cRA[s].y; d.rA[2] = hm.cRA[s].z; d.C0_n = hm.cRA[s].w;
                d.rB[0] = hm.cRB[s].x; d.rB[1] = hm.cRB[s].y; d.rB[2] = hm.cRB[s].z; d.C0_t[0] = hm.cRB[s].w;
                d.normal[0] = hm.cN[s].x; d.normal[1] = hm.cN[s].y; d.normal[2] = hm.cN[s].z; d.C0_t[1] = hm.cN[s].w; d.C0_t[2] = 0.0f;
                d.penetration = hm.cLam[s].w;
                const float lam[3] = {hm.cLam[s].x, hm.cLam[s].y, hm.cLam[s].z}, pen[3] = {hm.cPen[s].x, hm.cPen[s].y, hm.cPen[s].z};
                const float C[3] = {hm.cC[s].x, hm.cC[s].y, hm.cC[s].z}, lo[3] = {hm.cFmin[s].x, hm.cFmin[s].y, hm.cFmin[s].z};
                const float hi[3] = {hm.cFmax[s].x, hm.cFmax[s].y, hm.cFmax[s].z};
                for (int k = 0; k < 3; ++k) {
                    const int r = ci * 3 + k;
                    o.lambda[r] = lam[k]; o.penalty[r] = pen[k]; o.C[r] = C[k]; o.fmin[r] = lo[k]; o.fmax[r] = hi[k];
                    o.stiffness[r] = FLT_MAX; o.motor[r] = 0.0f;
                }
            }
        } else {
            const GForce& g = c->hG[refs[i].index];
            o.bodyA = g.bodyA; o.bodyB = g.bodyB; o.rows = g.kind == FK_HOST ? host_force_rows(g) : force_row_count(g.kind, 0);
            for (int r = 0; r < 6; ++r) {
                o.C[r] = g.C[r]; o.fmin[r] = g.fmin[r]; o.fmax[r] = g.fmax[r]; o.lambda[r] = g.lambda[r];
                o.penalty[r] = g.penalty[r]; o.stiffness[r] = g.stiffness[r]; o.motor[r] = g.motor[r];
            }
            if (g.kind != FK_IGNORE) for (int k = 0; k < 3; ++k) { o.rA[k] = g.rA[k]; o.rB[k] = g.rB[k]; }
            if (g.kind == FK_JOINT) for (int k = 0; k < 4; ++k) o.q0[k] = g.q0[k];
            if (g.kind == FK_SPRING) o.restLength = g.restLength;
        }
    }
    return n;
}

int avbd_download_force(AvbdContext* c, int list_index, AvbdForce* out)
{
    if (!c || !out || list_index < 0) return AVBD_E_INVALID;
    const int n = avbd_num_forces(c);
    if (n < 0) return n;
    if (list_index >= n) return c->fail(AVBD_E_INVALID, "avbd_download_force: index out of range");
    std::vector<AvbdForce> all((size_t)list_index + 1);
    const int rc = avbd_download_forces(c, all.data(), list_index + 1);
    if (rc < 0) return rc;
    *out = all[(size_t)list_index];
    return AVBD_OK;
}

/* Re-balanced ownership ("spatial slabs re-balanced every K steps") for the contexts of ONE process.  Ownership decides which
 * manifolds a rank holds — those that touch a body it sweeps — so changing it after the first step means moving manifolds (contacts, lambda,
 * penalty, their place in the force lists) between ranks: every rank's forces are downloaded with their creation keys, merged into the one list
 * the single-GPU run would hold (a manifold that lives on two ranks is taken from the rank that swept its owner, the earlier-created dynamic
 * endpoint, which ran its last dual), and every rank is rebuilt from the bodies (identical on all ranks after a step) and the part of that list
 * that touches a body it sweeps under the NEW ownership.  With exact cuts the step after a re-balance continues the single-GPU sweep bit for bit
 * (tests/test_gpu_exact_cuts.py::test_rebalanced_ownership_...), i.e. it equals a run that was started with the new ownership.  Host-side and
 * synchronous: meant for every few hundred steps.  (Ranks that are processes would exchange their lists the same way through the caller's
 * transport; not built.) */
/* ---- re-balanced ownership (SURVEY.md section 8e "spatial slabs re-balanced every K steps") ------------------------------------------------------
 * Two halves, so that ranks that are PROCESSES can do it too (the host side gathers the exports, e.g. with torch.distributed.all_gather_object):
 *   avbd_rebalance_export   this rank's force list, newest first, with the creation keys
 *   avbd_rebalance_import   all ranks' exports (any order) + the new ownership: one copy per force — the one of the rank that swept the contact's
 *                           owner under the OLD ownership —, the part of that one list this rank needs, bodies and step index kept
 * avbd_group_rebalance is the two in a row for contexts of one process. */
struct RebalanceItem { uint64_t key; AvbdForce f; int rank; };

static int rebalance_collect(AvbdContext* c, std::vector<RebalanceItem>& out)
{
    int rc;
    (void)hipSetDevice(c->device);
    if (!c->hostForceG.empty()) return c->fail(AVBD_E_INVALID, "rebalance: scenes with user-defined Force subclasses cannot be sharded");
    HostManifolds hm; std::vector<ForceRef> refs;
    if ((rc = sorted_force_refs(c, hm, refs))) return rc;
    std::vector<AvbdForce> fs(std::max<size_t>(refs.size(), 1));
    const int got = avbd_download_forces(c, fs.data(), (int)refs.size());
    if (got < 0) return got;
    if (got != (int)refs.size()) return c->fail(AVBD_E_INVALID, "rebalance: the force list changed under the download");
    for (int i = 0; i < got; ++i) out.push_back({refs[(size_t)i].key, fs[(size_t)i], c->rank});
    return AVBD_OK;
}

/* everything rebalance_apply can refuse before it touches the context (avbd_group_rebalance runs this on EVERY rank before the first one is
 * rebuilt: a refusal on rank r > 0 used to leave ranks 0..r-1 with the new ownership and the rest with the old — ADVICE round 4) */
static int rebalance_validate(AvbdContext* c, const int* owner, int nbodies, const std::vector<RebalanceItem>& all)
{
    const int n = c->nranks;
    if ((int)c->hBodies.size() != nbodies) return c->fail(AVBD_E_INVALID, "rebalance: owner[] must have one entry per body");
    if ((int)c->hOwner.size() != nbodies) return c->fail(AVBD_E_INVALID, "rebalance: the context has no ownership yet (avbd_set_ownership before the first step)");
    if (!c->hostForceG.empty()) return c->fail(AVBD_E_INVALID, "rebalance: scenes with user-defined Force subclasses cannot be sharded");
    for (int i = 0; i < nbodies; ++i) if (owner[i] >= n) return c->fail(AVBD_E_INVALID, "rebalance: owner index out of range");
    for (const RebalanceItem& it : all) if (it.rank < 0 || it.rank >= n) return c->fail(AVBD_E_INVALID, "rebalance: exporting rank out of range");
    return AVBD_OK;
}

static int rebalance_apply(AvbdContext* c, const int* owner, int nbodies, std::vector<RebalanceItem> all)
{
    int rc;
    (void)hipSetDevice(c->device);
    if ((rc = rebalance_validate(c, owner, nbodies, all))) return rc;
    /* one copy per force: the one of the rank that swept its owner (old ownership) */
    const std::vector<int> oldOwner = c->hOwner;
    auto prefer = [&](const AvbdForce& f) -> int {
        const int a = f.bodyA, b = f.bodyB;
        const bool da = a >= 0 && a < nbodies && oldOwner[(size_t)a] >= 0, db = b >= 0 && b < nbodies && oldOwner[(size_t)b] >= 0;
        const int own = (da && db) ? std::min(a, b) : (da ? a : (db ? b : -1));
        return own >= 0 ? oldOwner[(size_t)own] : 0;
    };
    /* (by key, then by exporting rank: every rank — every process — sees the same list whatever order the exports arrived in) */
    std::stable_sort(all.begin(), all.end(), [](const RebalanceItem& x, const RebalanceItem& y) { return x.key != y.key ? x.key > y.key : x.rank < y.rank; });
    std::vector<AvbdForce> merged;
    for (size_t i = 0; i < all.size();) {
        size_t j = i, pick = i;
        const int want = prefer(all[i].f);
        for (; j < all.size() && all[j].key == all[i].key; ++j) {
            if (all[j].f.bodyA != all[i].f.bodyA || all[j].f.bodyB != all[i].f.bodyB || all[j].f.kind != all[i].f.kind)
                return c->fail(AVBD_E_INVALID, "rebalance: two different forces carry one creation key (were the ranks built from the same scene?)");
            if (all[j].rank == want) pick = j;
        }
        merged.push_back(all[pick].f);
        i = j;
    }
    /* the merged list's own keys: position in the ONE list, the same on every rank that gets the force (manifolds created later have step >= 1
     * in their key and stay newer) */
    std::vector<AvbdForce> mine; std::vector<uint64_t> mineKeys;
    for (size_t i = 0; i < merged.size(); ++i) {
        const AvbdForce& f = merged[i];
        const bool ta = f.bodyA >= 0 && f.bodyA < nbodies && owner[f.bodyA] == c->rank, tb = f.bodyB >= 0 && f.bodyB < nbodies && owner[f.bodyB] == c->rank;
        if (f.kind != AVBD_FORCE_MANIFOLD || ta || tb) { mine.push_back(f); mineKeys.push_back(make_host_key(0, (uint64_t)(merged.size() - i))); }
    }
    /* the bodies: every rank holds the same state after a step */
    std::vector<AvbdBody> bodies((size_t)std::max(nbodies, 1));
    if (nbodies && (rc = avbd_download_bodies(c, bodies.data(), nbodies)) < 0) return rc;
    const AvbdParams params = c->params;
    const int stepIndex = c->stepIndex;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->hOwner.assign(owner, owner + nbodies);                         /* (before the rebuild: the upload marks the bodies this rank sweeps) */
    if ((rc = restore_state(c, params, stepIndex, bodies.data(), nbodies, mine.data(), (int)mine.size(), mineKeys.data()))) return rc;
    c->hOwner.assign(owner, owner + nbodies);
    c->bodiesStale = true; c->adjDirty = true;
    return AVBD_OK;
}

int avbd_rebalance_export(AvbdContext* c, AvbdForce* forces, unsigned long long* keys, int capacity)
{
    if (!c || capacity < 0 || (capacity > 0 && (!forces || !keys))) return AVBD_E_INVALID;
    std::vector<RebalanceItem> items;
    int rc = rebalance_collect(c, items);
    if (rc) return rc;
    if (capacity == 0) return (int)items.size();                      /* (a first call for the count) */
    if ((int)items.size() > capacity) return c->fail(AVBD_E_CAPACITY, "avbd_rebalance_export: the force list is longer than the caller's buffers");
    for (size_t i = 0; i < items.size(); ++i) { forces[i] = items[i].f; keys[i] = items[i].key; }
    return (int)items.size();
}

int avbd_rebalance_import(AvbdContext* c, const int* owner, int nbodies, const AvbdForce* forces, const unsigned long long* keys, const int* fromRank, int count)
{
    if (!c || !owner || nbodies < 0 || count < 0 || (count > 0 && (!forces || !keys || !fromRank))) return AVBD_E_INVALID;
    std::vector<RebalanceItem> all((size_t)count);
    for (int i = 0; i < count; ++i) all[(size_t)i] = {keys[i], forces[i], fromRank[i]};
    return rebalance_apply(c, owner, nbodies, std::move(all));
}

int avbd_group_rebalance(AvbdContext** ctxs, int n, const int* owner, int nbodies)
{
    if (!ctxs || n < 1 || !owner) return AVBD_E_INVALID;
    for (int r = 0; r < n; ++r) if (!ctxs[r]) return AVBD_E_INVALID;
    int rc;
    for (int r = 0; r < n; ++r) {
        AvbdContext* c = ctxs[r];
        if (c->nranks != n || c->rank != r) return c->fail(AVBD_E_INVALID, "avbd_group_rebalance: the contexts are not the ranks 0..n-1 of one sharded scene");
        if (c->xwAttached) return c->fail(AVBD_E_INVALID, "avbd_group_rebalance: ranks that are processes exchange their lists themselves (avbd_rebalance_export / _import)");
    }
    std::vector<RebalanceItem> all;
    for (int r = 0; r < n; ++r) if ((rc = rebalance_collect(ctxs[r], all))) return rc;
    for (int r = 0; r < n; ++r) if ((rc = rebalance_validate(ctxs[r], owner, nbodies, all))) return rc;   /* all ranks first, then apply */
    for (int r = 0; r < n; ++r) if ((rc = rebalance_apply(ctxs[r], owner, nbodies, all))) return rc;
    return AVBD_OK;
}

static int upload_forces_keyed(AvbdContext* c, const AvbdForce* in, int n, const uint64_t* keys);
int avbd_upload_forces(AvbdContext* c, const AvbdForce* in, int n) { return upload_forces_keyed(c, in, n, nullptr); }
/* `keys` (optional, descending): the creation keys the forces keep — a re-balance gives every rank a part of ONE list, and the same force must
 * carry the same key on every rank that holds it (the next re-balance merges the ranks' lists by key) */
static int upload_forces_keyed(AvbdContext* c, const AvbdForce* in, int n, const uint64_t* keys)
{
    if (!c || (!in && n > 0) || n < 0) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = validate_forces(c, in, n, (int)c->hBodies.size()))) return rc;
    if ((rc = upload_bodies_if_stale(c))) return rc;
    int nm = 0;
    for (int i = 0; i < n; ++i) nm += in[i].kind == AVBD_FORCE_MANIFOLD;
    if (nm > c->mCap) return c->fail(AVBD_E_CAPACITY, "manifold pool too small for upload");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HostManifolds h;
    h.n = nm;
    h.pair.resize(nm); h.key.resize(nm); h.nc.resize(nm); h.fric.resize(nm);
    size_t k4 = (size_t)nm * 4;
    std::vector<float4> hU1(k4, make_float4(0, 0, 0, 0)), hU2(k4, make_float4(0, 0, 0, 0));
    h.cRA.assign(k4, make_float4(0, 0, 0, 0)); h.cRB = h.cRA; h.cN = h.cRA; h.cLam = h.cRA; h.cPen = h.cRA; h.cC = h.cRA; h.cFmin = h.cRA; h.cFmax = h.cRA;
    h.cFS.assign(k4, make_int2(0, 0));
    c->hG.clear();
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const AvbdForce& f = in[i];
        const uint64_t key = keys ? keys[i] : make_host_key(0, (uint64_t)(n - i)); /* newest first → descending keys */
        if (f.kind == AVBD_FORCE_MANIFOLD) {
            h.pair[m] = make_int2(f.bodyA, f.bodyB); h.key[m] = key; h.nc[m] = f.numContacts; h.fric[m] = f.combinedFriction;
            for (int ci = 0; ci < f.numContacts; ++ci) {
                const size_t s = (size_t)m * 4 + ci;
                const AvbdContact& d = f.contacts[ci];
                h.cRA[s] = make_float4(d.rA[0], d.rA[1], d.rA[2], d.C0_n);
                h.cRB[s] = make_float4(d.rB[0], d.rB[1], d.rB[2], d.C0_t[0]);
                h.cN[s] = make_float4(d.normal[0], d.normal[1], d.normal[2], d.C0_t[1]);
                {   /* tangent memo of the stored normal (contact_basis_memo invariant) */
                    f3 t1, t2;
                    contact_tangents(mk3(d.normal[0], d.normal[1], d.normal[2]), t1, t2);
                    hU1[s] = make_float4(t1.x, t1.y, t1.z, 0.f); hU2[s] = make_float4(t2.x, t2.y, t2.z, 0.f);
                }
                h.cLam[s] = make_float4(f.lambda[ci * 3], f.lambda[ci * 3 + 1], f.lambda[ci * 3 + 2], d.penetration);
                h.cPen[s] = make_float4(f.penalty[ci * 3], f.penalty[ci * 3 + 1], f.penalty[ci * 3 + 2], 0.f);
                h.cC[s] = make_float4(f.C[ci * 3], f.C[ci * 3 + 1], f.C[ci * 3 + 2], 0.f);
                h.cFmin[s] = make_float4(f.fmin[ci * 3], f.fmin[ci * 3 + 1], f.fmin[ci * 3 + 2], 0.f);
                h.cFmax[s] = make_float4(f.fmax[ci * 3], f.fmax[ci * 3 + 1], f.fmax[ci * 3 + 2], 0.f);
                h.cFS[s] = make_int2(d.feature, d.stick);
            }
            ++m;
        } else {
            GForce g; init_gforce(g, f.kind, f.bodyA, f.bodyB);
            g.key = key;
            for (int r = 0; r < 6; ++r) {
                g.C[r] = f.C[r]; g.fmin[r] = f.fmin[r]; g.fmax[r] = f.fmax[r]; g.lambda[r] = f.lambda[r];
                g.penalty[r] = f.penalty[r]; g.stiffness[r] = f.stiffness[r]; g.motor[r] = f.motor[r];
            }
            for (int k = 0; k < 3; ++k) { g.rA[k] = f.rA[k]; g.rB[k] = f.rB[k]; }
            for (int k = 0; k < 4; ++k) g.q0[k] = f.q0[k];
            g.restLength = f.restLength;
            c->hG.push_back(g);
        }
    }
    c->hostKeyCounter = (uint64_t)n + 1;
    if (keys && n > 0) c->hostKeyCounter = std::max<uint64_t>(c->hostKeyCounter, (keys[0] & ((1ull << 40) - 1)) + 1);
    ManifoldBuf& b = c->mbuf[c->cur];
    c->narrowOrderValid = false;                                       /* (the slots are rewritten: k_narrowphase walks them in slot order once) */
    if (nm) {
        HIPCHK(c, hipMemcpy(b.pair.p, h.pair.data(), nm * sizeof(int2), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.key.p, h.key.data(), nm * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.nc.p, h.nc.data(), nm * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.fric.p, h.fric.data(), nm * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cRA.p, h.cRA.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cRB.p, h.cRB.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cN.p, h.cN.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cU1.p, hU1.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cU2.p, hU2.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cLam.p, h.cLam.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cPen.p, h.cPen.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cC.p, h.cC.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cFmin.p, h.cFmin.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cFmax.p, h.cFmax.data(), k4 * sizeof(float4), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b.cFS.p, h.cFS.data(), k4 * sizeof(int2), hipMemcpyHostToDevice));
    }
    Counters cnt;
    std::memset(&cnt, 0, sizeof(cnt));
    cnt.numManifolds = nm;
    HIPCHK(c, hipMemcpy(c->dCnt.p, &cnt, sizeof(cnt), hipMemcpyHostToDevice));
    c->stats.numManifolds = nm;
    c->gStale = true; c->gDeviceAhead = false; c->adjDirty = true;
    return AVBD_OK;
}

/* ---- row API on the device ------------------------------------------------------------- */
namespace {
#include "avbd_capi_rowapi.inc"

bool load_rccl()
{
    if (g_rccl.lib) return true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather) return false;
    g_rccl.lib = h;
    return true;
}

int rccl_allgather(AvbdContext* c)
{
    /* one all-gather of {position+radius, orientation} rows per iteration: 2 float4 = 8 floats per body */
    const size_t count = (size_t)8 * std::max(1, c->maxCount);
    int r = g_rccl.AllGather(c->dSend.p, c->dRecv.p, count, /* ncclFloat32 */ 7, c->comm, c->stream);
    if (r != 0) return c->fail(AVBD_E_COMM, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "ncclAllGather failed");
    return AVBD_OK;
}

int rccl_exchange_velocities(AvbdContext* c)
{
    /* the velocity half of the 13-float exchange (group_exchange_velocities has the contract): same row count as the poses */
    if (c->exchangeFloats != 13 || c->nranks < 2) return AVBD_OK;
    int rc;
    if ((rc = step_pack_velocities(c))) return rc;
    if ((rc = rccl_allgather(c))) return rc;
    return step_unpack_velocities(c);
}

int rccl_allgather_inplace(AvbdContext* c, float4* buffer, hipStream_t stream)
{
    /* in place: this rank's block of `buffer` is its contribution (sendbuff = recvbuff + rank * count) */
    const size_t count = (size_t)8 * std::max(1, c->maxCount);
    int r = g_rccl.AllGather(buffer + (size_t)2 * std::max(1, c->maxCount) * c->rank, buffer, count, /* ncclFloat32 */ 7, c->comm, stream);
    if (r != 0) return c->fail(AVBD_E_COMM, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "ncclAllGather failed");
    return AVBD_OK;
}
} // namespace

static void destroy_comm(void* comm) { if (g_rccl.CommDestroy) g_rccl.CommDestroy(comm); }

extern "C" {

int avbd_partition_by_axis(int n, const float* position, const float* invMass, int nranks, int axis, int* owner)
{
    /* equal-count slabs of the dynamic bodies along one world axis; static bodies are replicated (-1) */
    if (n < 0 || !position || !invMass || !owner || nranks < 1 || axis < 0 || axis > 2) return AVBD_E_INVALID;
    std::vector<int> dyn;
    for (int i = 0; i < n; ++i) { owner[i] = -1; if (invMass[i] > 0.0f) dyn.push_back(i); }
    std::stable_sort(dyn.begin(), dyn.end(), [&](int a, int b) { return position[3 * (size_t)a + axis] < position[3 * (size_t)b + axis]; });
    const size_t m = dyn.size();
    for (size_t k = 0; k < m; ++k) owner[dyn[k]] = (int)std::min<size_t>((size_t)nranks - 1, k * (size_t)nranks / std::max<size_t>(m, 1));
    return AVBD_OK;
}

int avbd_exchange_layout(int n, const int* owner, int nranks, int* rankStart, int* perm, int* row, int* maxCount)
{
    /* Host-only (no device needed): the layout of the per-iteration pose exchange.  Bodies with owner r (0 <= r < nranks) are
     * rank r's; negative owners are replicated (static) bodies.  rankStart[nranks + 1]: prefix counts; perm[n]: swept bodies
     * grouped by rank, creation order inside a rank; row[n]: r * max(1, maxCount) + position inside rank r's block — the row of
     * a body in every all-gather buffer (equal-sized blocks, one per rank) — or -1; maxCount: largest block. */
    if (n < 0 || nranks < 1 || (n && !owner) || !rankStart || (n && (!perm || !row)) || !maxCount) return AVBD_E_INVALID;
    for (int r = 0; r <= nranks; ++r) rankStart[r] = 0;
    for (int i = 0; i < n; ++i) {
        if (owner[i] >= nranks) return AVBD_E_INVALID;
        if (owner[i] >= 0) ++rankStart[owner[i] + 1];
    }
    int most = 0;
    for (int r = 0; r < nranks; ++r) { most = std::max(most, rankStart[r + 1]); rankStart[r + 1] += rankStart[r]; }
    *maxCount = most;
    std::vector<int> cur(rankStart, rankStart + nranks);
    const int block = std::max(1, most);
    for (int i = 0; i < n; ++i) {
        row[i] = -1;
        if (owner[i] < 0) continue;
        const int k = cur[(size_t)owner[i]]++;
        perm[k] = i;
        row[i] = owner[i] * block + (k - rankStart[owner[i]]);
    }
    return AVBD_OK;
}

int avbd_set_ownership(AvbdContext* c, int nranks, int rank, const int* owner, int n)
{
    if (!c || nranks < 1 || rank < 0 || rank >= nranks || n != (int)c->hBodies.size() || (!owner && n > 0)) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    int rc = download_bodies_to_host(c);
    if (rc) return rc;
    /* before this context has stepped on its present content: a fresh scene, or one restored from a snapshot (avbd_load_snapshot: stepIndex is the snapshot's,
     * the whole force list is here and every rank keeps what touches its slab from the first sharded step on).  Later: avbd_group_rebalance / avbd_rebalance_import */
    if (c->stepsSinceLoad != 0) return c->fail(AVBD_E_INVALID, "ownership is static: set it before the first step of a scene or right after avbd_load_snapshot (later: avbd_group_rebalance)");
    c->nranks = nranks; c->rank = rank;
    c->hOwner.assign(owner, owner + n);
    c->bodiesStale = true; c->adjDirty = true;
    /* A context that already holds manifolds (a snapshot was loaded, or state uploaded): keep what a rank of a run started from scratch would hold — the manifolds that touch a
     * body it sweeps, every other force — with keys from the position in the ONE list every rank was handed (what avbd_group_rebalance leaves; a manifold between two bodies of
     * other ranks stalled the all-gather form's sweep). */
    const int nf = nranks > 1 ? avbd_num_forces(c) : 0;
    if (nf > 0) {
        std::vector<AvbdForce> all((size_t)nf), mine; std::vector<uint64_t> mineKeys;
        std::vector<AvbdBody> bodies((size_t)std::max(n, 1));
        if ((rc = avbd_download_forces(c, all.data(), nf)) < 0) return rc;
        for (int i = 0; i < nf; ++i) {
            const AvbdForce& f = all[(size_t)i];
            const bool ta = f.bodyA >= 0 && f.bodyA < n && owner[f.bodyA] == rank, tb = f.bodyB >= 0 && f.bodyB < n && owner[f.bodyB] == rank;
            if (f.kind != AVBD_FORCE_MANIFOLD || ta || tb) { mine.push_back(f); mineKeys.push_back(make_host_key(0, (uint64_t)(nf - i))); }
        }
        if ((int)mine.size() != nf) {
            if (n && (rc = avbd_download_bodies(c, bodies.data(), n)) < 0) return rc;
            const AvbdParams params = c->params;
            const int stepIndex = c->stepIndex;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if ((rc = restore_state(c, params, stepIndex, bodies.data(), n, mine.data(), (int)mine.size(), mineKeys.data()))) return rc;
            c->hOwner.assign(owner, owner + n);
            c->bodiesStale = true; c->adjDirty = true;
        }
    }
    return AVBD_OK;
}

int avbd_comm_unique_id(char id[128])
{
    if (!id) return AVBD_E_INVALID;
    if (!load_rccl()) return AVBD_E_COMM;
    Id128 u; std::memset(&u, 0, sizeof(u));
    if (g_rccl.GetUniqueId(&u) != 0) return AVBD_E_COMM;
    std::memcpy(id, u.b, 128);
    return AVBD_OK;
}

int avbd_comm_init(AvbdContext* c, int nranks, int rank, const char id[128])
{
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return AVBD_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(AVBD_E_NODEVICE, "hipSetDevice failed");
    if (!load_rccl()) return c->fail(AVBD_E_COMM, "librccl.so could not be loaded");
    Id128 u; std::memcpy(u.b, id, 128);
    void* comm = nullptr;
    /* The all-gather of a persistent sharded step (AVBD_SHARD_PERSISTENT=1) runs BESIDE the launch that waits for it, on the workgroup slots that
     * launch leaves free (kShardFreeSlots): its kernel must not need more workgroups than that.  RCCL reads NCCL_MAX_NCHANNELS once per PROCESS, when
     * its first communicator is created — so the cap is the caller's to set, before that moment (bench.py does; ADVICE round 3: a library that
     * calls setenv / unsetenv around ncclCommInitRank is neither thread-safe nor effective once torch has created a communicator).  A persistent
     * step whose collective finds no room ends in the bounded time-out of the waves, never in a hang. */
    int r = g_rccl.CommInitRank(&comm, nranks, u, rank);
    if (r != 0) return c->fail(AVBD_E_COMM, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "ncclCommInitRank failed");
    c->comm = comm;
    return AVBD_OK;
}

#include "avbd_capi_group.inc"

/* ---- measurement ------------------------------------------------------------------------ */
/* debug: per-wave time breakdown of the last k_iterate launch (only in -DAVBD_DF_STATS builds) */
int avbd_debug_topology_times(AvbdContext* c, unsigned long long* out7)
{
    if (!c || !out7) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out7, c->dFlags.p + 2 * (size_t)c->bodyCap + 16 + 1024 * 32 + 20 * 1024 * 8, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return AVBD_OK;
}

int avbd_debug_front_times(AvbdContext* c, unsigned long long* out4)
{
    if (!c || !out4) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ManifoldBuf& m = c->mbuf[c->cur];
    HIPCHK(c, hipMemcpy(out4, reinterpret_cast<unsigned long long*>(m.cC.p + (size_t)m.cap * 4 - 8), 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return AVBD_OK;
}

int avbd_debug_front_blocks(AvbdContext* c, unsigned long long* out, int blocks)
{
    /* -DAVBD_DF_STATS builds: sixteen words per workgroup 0 .. blocks-1 of the last k_front_small (10 ns ticks): [0] first instant, [15] last instant; of the first
     * pair of its first group [1] after the loads, [2] after Manifold::initialize, [3] at its end, and inside initialize [4] SAT axes tested, [5] axis chosen, [6] incident face built,
     * [7] clipped, [8] contacts chosen, [9] collide done, [10] features matched; zeros in the product build */
    if (!c || !out || blocks < 1) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memset(out, 0, (size_t)blocks * 16 * sizeof(unsigned long long));
#ifdef AVBD_DF_STATS
    HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(gFrontBlockStamps), (size_t)std::min(blocks, kFrontStampBlocks) * 16 * sizeof(unsigned long long)));
#endif
    return AVBD_OK;
}

int avbd_debug_task_times(AvbdContext* c, unsigned* out, int iterations)
{
    if (!c || !out) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->dFlags.p + 2 * (size_t)c->bodyCap + 16 + 1024 * 32, (size_t)iterations * 1024 * 8 * sizeof(unsigned), hipMemcpyDeviceToHost));
    return AVBD_OK;
}

int avbd_debug_group_task_times(AvbdContext* c, unsigned* out, int iterations)
{
    /* -DAVBD_DF_STATS builds: per task (iteration-major, body index) {inputs seen, pose published, task end, wave} in 10 ns ticks */
    if (!c || !out || iterations < 1) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t words = (size_t)iterations * c->hBodies.size() * 4;
    if (!c->xPrivN.p || c->privNCap * sizeof(float4) < words * sizeof(unsigned)) return c->fail(AVBD_E_INVALID, "no task time buffer");
    HIPCHK(c, hipMemcpy(out, c->xPrivN.p, words * sizeof(unsigned), hipMemcpyDeviceToHost));
    return AVBD_OK;
}

/* debug: after a sharded / group launch was abandoned (dependency time-out), what every body's sub-group was waiting for:
 * out[4 * body] = {iteration, partner or -1, records missing (1 pose, 2 visit, 4 dual, 8 gather), wave} (tools/debug_sharded.py) */
extern "C" int avbd_debug_wait_graph(AvbdContext* c, int* out)
{
    if (!c || !out || !c->xPrivPose.p) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    std::vector<float4> h((size_t)2 * c->hBodies.size());
    if (hipMemcpy(h.data(), c->xPrivPose.p, h.size() * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) return AVBD_E_HIP;
    for (size_t b = 0; b < c->hBodies.size(); ++b) std::memcpy(out + 8 * b, &h[2 * b], 32);
    return AVBD_OK;
}

/* debug: the counters and words of the persistent sharded step: out[0..63] done[k], [64] signal word, [65] gather word, [66] stamp base, [67] owned bodies */
extern "C" int avbd_debug_shard_words(AvbdContext* c, unsigned* out)
{
    if (!c || !out || !c->dDone.p || !c->sigDone || !c->sigGather) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (hipMemcpy(out, c->dDone.p, 64 * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return AVBD_E_HIP;
    if (hipMemcpy(out + 64, c->sigDone, 4, hipMemcpyDeviceToHost) != hipSuccess) return AVBD_E_HIP;
    if (hipMemcpy(out + 65, c->sigGather, 4, hipMemcpyDeviceToHost) != hipSuccess) return AVBD_E_HIP;
    out[66] = c->xSerial * 64u; out[67] = (unsigned)(c->hRankStart[(size_t)c->rank + 1] - c->hRankStart[(size_t)c->rank]);
    return AVBD_OK;
}

/* debug: stamps (minus the launch's base) of the three pose records of every body: out[3 * body + k] */
extern "C" int avbd_debug_pose_stamps(AvbdContext* c, int* out)
{
    if (!c || !out || !c->xPose.p) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    std::vector<float4> h((size_t)3 * c->hBodies.size());
    if (hipMemcpy(h.data(), c->xPose.p, h.size() * sizeof(float4), hipMemcpyDeviceToHost) != hipSuccess) return AVBD_E_HIP;
    for (size_t i = 0; i < h.size(); ++i) { unsigned wv; std::memcpy(&wv, &h[i].w, 4); out[i] = (int)(wv - c->xSerial * 64u); }
    return AVBD_OK;
}

extern "C" int avbd_debug_shard_words_nosync(AvbdContext* c, unsigned* out)
{
    if (!c || !out || !c->dDone.p || !c->sigDone || !c->sigGather) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    if (hipMemcpy(out, c->dDone.p, 64 * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return AVBD_E_HIP;
    if (hipMemcpy(out + 64, c->sigDone, 4, hipMemcpyDeviceToHost) != hipSuccess) return AVBD_E_HIP;
    if (hipMemcpy(out + 65, c->sigGather, 4, hipMemcpyDeviceToHost) != hipSuccess) return AVBD_E_HIP;
    out[66] = c->xSerial * 64u; out[67] = 0;
    return AVBD_OK;
}

/* debug (AVBD_DEBUG_KEEP_EXCHANGE_WAITING runs): let the exchange streams that wait for this rank go on */
extern "C" int avbd_debug_release_exchange(AvbdContext* c)
{
    if (!c || !c->sigDone) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    const unsigned big = c->xSerial * 64u + 63u;
    return hipMemcpy(c->sigDone, &big, 4, hipMemcpyHostToDevice) == hipSuccess ? AVBD_OK : AVBD_E_HIP;
}

/* debug: what k_place_chains decided for the last step {nd, groups, grouped bodies, valid} and, optionally, the table (1024 ints) */
extern "C" int avbd_debug_place_info(AvbdContext* c, int* info4, int* place1024)
{
    if (!c || !info4) return AVBD_E_INVALID;
    if (!c->dPlaceInfo.p) { info4[0] = info4[1] = info4[2] = info4[3] = 0; return AVBD_OK; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(info4, c->dPlaceInfo.p, sizeof(PlaceInfo), hipMemcpyDeviceToHost));
    if (place1024) HIPCHK(c, hipMemcpy(place1024, c->dPlace.p, 1024 * sizeof(int), hipMemcpyDeviceToHost));
    return AVBD_OK;
}

int avbd_debug_iterate_stats(AvbdContext* c, unsigned long long* out, int waves)
{
    if (!c || !out) return AVBD_E_INVALID;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->dFlags.p + 2 * (size_t)c->bodyCap + 16, (size_t)waves * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return AVBD_OK;
}

int avbd_fast_plan_info(AvbdContext* c, int out[8])
{
    if (!c || !out) return AVBD_E_INVALID;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!c->fPlan.p) return AVBD_OK;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->fPlan.p, 8 * sizeof(int), hipMemcpyDeviceToHost));
    return AVBD_OK;
}
int avbd_fast_colors(AvbdContext* c, int* out, int capacity)
{
    if (!c || !out) return AVBD_E_INVALID;
    const int n = (int)c->hBodies.size();
    if (capacity < n) return c->fail(AVBD_E_INVALID, "avbd_fast_colors: capacity below the number of bodies");
    if (!c->fColor[c->fColorCur].p || !c->fastLaunched) return c->fail(AVBD_E_INVALID, "avbd_fast_colors: the last step was not a fast step");
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->fColor[c->fColorCur].p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return n;
}
int avbd_profile_enable(AvbdContext* c, int enable) { if (!c) return AVBD_E_INVALID; c->profiling = enable != 0; return AVBD_OK; }
int avbd_profile_reset(AvbdContext* c) { if (!c) return AVBD_E_INVALID; c->prof = AvbdProfile{}; return AVBD_OK; }
int avbd_profile_get(AvbdContext* c, AvbdProfile* out) { if (!c || !out) return AVBD_E_INVALID; *out = c->prof; return AVBD_OK; }
void* avbd_stream(AvbdContext* c) { return c ? (void*)c->stream : nullptr; }

} // extern "C"
