/*
 * solver.h — drop-in C++17 host API of the AVBD solver, backed by libavbd_hip.so (MI355X).
 *
 * Mirrors the public interface of the reference's source/solver.h (Rigid :47-82, Force :85-109,
 * Manifold :112-143, Solver :146-181; constants :25-36) so that scene code and the headless
 * driver written against the reference compile and behave the same.  The objects here are HOST
 * MIRRORS: authoritative simulation state lives in HBM inside an AvbdContext (include/avbd_hip.h)
 * and Solver::step() runs entirely on the GPU — there is no CPU solve path in this library.
 *
 *   new Rigid(...)            → queued, created on the device by the next step()/flush()
 *   body->position = ...      → detected by comparison with the last downloaded value and uploaded
 *   solver->step()            → avbd_step(ctx, 1) + download of body state and diagnostics
 *   force->computeConstraint  → evaluated on the device through avbd_force_compute_constraint
 *
 * Differences from the reference that a caller can observe are listed in INTEGRATION.md
 * (manifold mirrors are materialised on demand by Solver::syncForces(); a user-defined Force
 * subclass keeps its virtuals on the host: the device fetches its rows before every visit —
 * exact, slow — see INTEGRATION.md).
 */
#pragma once

#include "maths.h"
#include "../../include/avbd_hip.h"

#define MAX_CONSTRAINT_ROWS 12           /* solver.h:25 */
#define PENALTY_MIN 20000.0f             /* solver.h:26 */
#define PENALTY_MAX 1000000000.0f        /* solver.h:27 */
#define COLLISION_MARGIN 0.02f           /* solver.h:28 */
#define STICK_THRESH 0.02f               /* solver.h:29 */
constexpr float PENETRATION_SLOP = 0.005f; /* solver.h:36 */

struct Rigid;
struct Force;
struct Manifold;
struct Joint;
struct Spring;
struct IgnoreCollision;
struct Solver;

struct Rigid {
    Solver* solver;
    Force* forces;   /* head of this body's force list (user forces + synced manifolds), newest first */
    Rigid* next;     /* solver->bodies list, newest first (rigid.cpp:20-21) */

    int id;
    static int next_id;

    vec3 position;
    quat orientation;
    vec3 linearVelocity, angularVelocity;
    vec3 prevLinearVelocity, prevAngularVelocity;
    vec3 initialPosition;
    quat initialOrientation;
    vec3 inertialPosition;
    quat inertialOrientation;

    vec3 size;
    float mass, invMass;
    mat3 inertiaTensor;
    mat3 invInertiaTensor;
    float friction;
    float radius;

    int index;       /* creation index inside the device context (build-only field) */
    float density;   /* kept so the device can redo rigid.cpp:24-40 bit for bit */

    Rigid(Solver* solver, const vec3& size, float density, float friction, const vec3& pos, const quat& orient = quat(),
          const vec3& linVel = vec3(), const vec3& angVel = vec3());
    ~Rigid();

    mat3 getInvInertiaTensorWorld() const;   /* rigid.cpp:51-54 */
    mat3 getInertiaTensorWorld() const;      /* rigid.cpp:56-59 */
    bool isConstrainedTo(Rigid* other) const;/* rigid.cpp:61-69 (needs Solver::syncForces() to see manifolds) */
    void draw() const {}
};

struct Force {
    Solver* solver;
    Rigid* bodyA;
    Rigid* bodyB;
    Force *nextA, *nextB, *next;

    float C[MAX_CONSTRAINT_ROWS];
    float fmin[MAX_CONSTRAINT_ROWS], fmax[MAX_CONSTRAINT_ROWS];
    float lambda[MAX_CONSTRAINT_ROWS];
    float penalty[MAX_CONSTRAINT_ROWS];
    float motor[MAX_CONSTRAINT_ROWS];
    float stiffness[MAX_CONSTRAINT_ROWS];
    float fracture[MAX_CONSTRAINT_ROWS];

    int deviceKind;  /* AVBD_FORCE_* of a built-in kind; -1: a user subclass not yet registered; AVBD_FORCE_HOST once it is (build-only field) */
    int listIndex;   /* position in the device's newest-first force list after syncForces(), else -1 */
    int serial;      /* creation serial inside the solver (build-only field: orders built-in and user-defined forces alike) */

    Force(Solver* solver, Rigid* bodyA, Rigid* bodyB);
    virtual ~Force();

    virtual int getRowCount() const = 0;
    virtual bool initialize() = 0;
    virtual void computeConstraint(float alpha) = 0;
    virtual void computeDerivatives(vec3& J_linear, vec3& J_angular, const Rigid* body, int row) const = 0;
    virtual void draw() const {}
    virtual bool isManifold() const { return false; }

protected:
    void deviceComputeConstraint(float alpha);
    void deviceComputeDerivatives(vec3& Jl, vec3& Ja, const Rigid* body, int row) const;
};

struct Manifold : Force {
    union FeaturePair {
        struct { unsigned char in_A, out_A, in_B, out_B; } e;
        int value;
    };
    struct Contact {
        FeaturePair feature;
        vec3 rA, rB;
        vec3 normal;
        float penetration;
        float C0_n;
        vec3 C0_t;
        bool stick;
    };
    Contact contacts[4];
    int numContacts;
    float combinedFriction;

    Manifold(Solver* solver, Rigid* bodyA, Rigid* bodyB);
    int getRowCount() const override { return numContacts * 3; }
    bool initialize() override { return numContacts > 0; }
    void computeConstraint(float alpha) override { deviceComputeConstraint(alpha); }
    void computeDerivatives(vec3& J_linear, vec3& J_angular, const Rigid* body, int row) const override
    {
        deviceComputeDerivatives(J_linear, J_angular, body, row);
    }
    bool isManifold() const override { return true; }
    /* Manifold::collide (collision.cpp:420-489) evaluated on the device for one pair */
    static int collide(Rigid* bodyA, Rigid* bodyB, Contact* contacts, bool flip);
};

struct Solver {
    float dt;
    vec3 gravity;
    int iterations;
    float alpha, beta, gamma;
    bool postStabilize;
    Rigid* bodies;
    Force* forces;

    struct Diagnostics {
        float maxPenetration;
        float maxConstraintViolation;
        float maxLinearSpeed;
        float maxAngularSpeed;
        float maxNormalImpulse;
        int activeContacts;
        int activeManifolds;
        int dynamicBodies;
    };

    bool enableDiagnostics;
    int logFrequency;
    int stepIndex;
    Diagnostics lastDiagnostics;

    Solver();                              /* creates its own AvbdContext on device 0 (AVBD_DEVICE overrides) */
    explicit Solver(AvbdContext* borrowed);/* drives an existing context; does not destroy it */
    ~Solver();

    Rigid* pick(const vec3& origin, const vec3& dir, vec3& local); /* solver.cpp:145-228 on the device (avbd_pick) */
    void clear();
    void defaultParams();
    void step();
    void draw() {}

    /* ---- build-only extensions (not in the reference) ---- */
    bool hostSync;          /* true (default): upload edited bodies before and download all bodies after every step() */
    bool mirrorManifolds;   /* true: rebuild host Manifold mirrors after every step() (default false: call syncForces()) */
    void stepMany(int n);   /* n device steps back to back, one download at the end */
    void flush();           /* create queued bodies/forces on the device now */
    void pull();            /* download body state + diagnostics into the mirrors */
    void syncForces();      /* rebuild solver->forces / body->forces mirrors from the device (newest first) */
    /* checkpoint / resume (absent in the reference): the whole device state as one file.  loadSnapshot
     * replaces the scene: bodies get fresh Rigid mirrors (same creation order, hence the same ids relative to each other),
     * contact manifolds and joints/springs live on the device (syncForces() mirrors the manifolds on demand). */
    void saveSnapshot(const char* path);
    void loadSnapshot(const char* path);
    /* One scene over several GPUs (BASELINE configs[3..4]; absent in the reference): `n` Solvers that hold the SAME scene, one per
     * rank — shardAcross cuts the dynamic bodies into equal slabs along `axis` (avbd_partition_by_axis) and gives solver r slab r;
     * stepGroup then steps all of them in lock-step (avbd_group_step) and downloads the state into every Solver's mirrors.
     * exactCuts: results bit-identical to one GPU (avbd_set_shard_mode(ctx, 1)); otherwise one all-gather per iteration. */
    static void shardAcross(Solver** solvers, int n, int axis, bool exactCuts);
    static void stepGroup(Solver** solvers, int n, int steps);
    /* ownership re-balanced after steps have run (slabs re-balanced every K steps): equal slabs along `axis` from the CURRENT
     * positions, the manifolds move with the bodies (avbd_group_rebalance); with exact cuts the run continues the single-GPU sweep bit for bit */
    static void rebalance(Solver** solvers, int n, int axis);
    AvbdContext* context() const { return ctx; }
    int bodyCount() const { return numBodies; }
    Rigid* bodyByIndex(int i) const;

private:
    friend struct Rigid;
    friend struct Force;
    friend struct Manifold;
    friend struct Joint;
    friend struct Spring;
    friend struct IgnoreCollision;
    AvbdContext* ctx;
    bool ownsCtx;
    int numBodies;
    int flushedBodies;
    struct Impl;
    Impl* impl;
    void init(AvbdContext* borrowed);
    void releaseMirrors();
    void pushParams();
    void pushEditedBodies();
    void refreshMirror(Force* f, const AvbdForce& record);
    static int hostForceCallback(void* user, int hostForce, int phase, float alpha, int body, const float* poseA, const float* poseB, AvbdHostRows* rows);
    void fail(const char* what, int rc) const;
};
