/*
 * solver_host.cpp — implementation of the drop-in host API (host/solver.h, joint.h, spring.h,
 * ignorecollision.h) on top of the C ABI of include/avbd_hip.h.  No physics is computed here:
 * construction-time set-up values (box mass properties rigid.cpp:24-40, joint frames
 * joint.cpp:17-20/46-52, spring rest length spring.cpp:19-23) are evaluated with the shared
 * expression trees of csrc/avbd_math.h; everything else is a call into the device library.
 */
#include "solver.h"
#include "joint.h"
#include "spring.h"
#include "ignorecollision.h"
#include "scenes.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <utility>
#include <vector>

int Rigid::next_id = 1; /* never reset, like rigid.cpp:10 */

struct Solver::Impl {
    std::vector<Rigid*> byIndex;        /* creation order */
    std::vector<Force*> userForces;     /* Joint/Spring/IgnoreCollision in creation order */
    size_t flushedForces = 0;
    std::vector<AvbdBody> shadow;       /* body state as last seen by the device */
    std::vector<Manifold*> manifolds;   /* mirrors created by syncForces() */
    std::vector<Force*> hostForces;     /* user-defined subclasses by device ordinal (avbd_add_host_force) */
    int nextSerial = 0;
    bool clearing = false;
    bool syncing = false;               /* syncForces() is replacing the manifold mirrors */
};

/* ------------------------------------------------------------------ Rigid ----------------- */
Rigid::Rigid(Solver* s, const vec3& size_, float density_, float friction_, const vec3& pos, const quat& orient,
             const vec3& linVel, const vec3& angVel)
    : solver(s), forces(nullptr), next(nullptr), id(next_id++), position(pos), orientation(orient), linearVelocity(linVel),
      angularVelocity(angVel), prevLinearVelocity(linVel), prevAngularVelocity(angVel), size(size_), friction(friction_),
      density(density_)
{
    next = s->bodies;
    s->bodies = this;
    /* rigid.cpp:24-40 */
    mass = size.x * size.y * size.z * density;
    invMass = (mass > 0.0f) ? 1.0f / mass : 0.0f;
    radius = length(size) * 0.5f;
    if (invMass > 0.0f) {
        float Ixx = (1.0f / 12.0f) * mass * (size.y * size.y + size.z * size.z);
        float Iyy = (1.0f / 12.0f) * mass * (size.x * size.x + size.z * size.z);
        float Izz = (1.0f / 12.0f) * mass * (size.x * size.x + size.y * size.y);
        inertiaTensor = mat3(vec3(Ixx, 0, 0), vec3(0, Iyy, 0), vec3(0, 0, Izz));
        invInertiaTensor = mat3(vec3(1.0f / Ixx, 0, 0), vec3(0, 1.0f / Iyy, 0), vec3(0, 0, 1.0f / Izz));
    } else {
        inertiaTensor = mat3(vec3(), vec3(), vec3());
        invInertiaTensor = mat3(vec3(), vec3(), vec3());
    }
    index = s->numBodies++;
    s->impl->byIndex.push_back(this);
}

Rigid::~Rigid()
{
    Solver* s = solver;
    const bool onDevice = !s->impl->clearing && index < s->flushedBodies;
    if (onDevice) {
        /* `delete body` of a body that lives on the device (rigid.cpp:43-49): the device state is rebuilt without it and without
         * the forces attached to it.  Joint / Spring mirrors attached to this body stay alive as plain host objects (the
         * reference leaves them dangling); they no longer exist for the solver. */
        s->flush();
        int rc = avbd_remove_body(s->ctx, index);
        if (rc) s->fail("avbd_remove_body", rc);
        auto& uf = s->impl->userForces;
        for (size_t i = 0; i < uf.size();) {
            if (uf[i]->bodyA == this || uf[i]->bodyB == this) {
                if (i < s->impl->flushedForces) --s->impl->flushedForces;
                uf[i]->deviceKind = -2;                                  /* gone from the device */
                uf.erase(uf.begin() + (long)i);
            } else ++i;
        }
        for (Force* f : uf) f->listIndex = -1;
        for (Manifold* m : s->impl->manifolds) m->listIndex = -1;
        s->impl->shadow.erase(s->impl->shadow.begin() + index);
        --s->flushedBodies;
    }
    /* forces that still point at this body lose the pointer (a dangling pointer in the reference) */
    for (Force* f = s->forces; f; f = f->next) { if (f->bodyA == this) f->bodyA = nullptr; else if (f->bodyB == this) f->bodyB = nullptr; }
    Rigid** p = &s->bodies;
    while (*p && *p != this) p = &(*p)->next;
    if (*p) *p = next;
    if (!s->impl->clearing) { /* drop it from the creation list */
        auto& v = s->impl->byIndex;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == this) { v.erase(v.begin() + (long)i); break; }
        for (size_t i = 0; i < v.size(); ++i) v[i]->index = (int)i;
        s->numBodies = (int)v.size();
        if (onDevice) s->syncForces();                                    /* manifold mirrors and list links of the smaller scene */
    }
}

/* ------------------------------------------------------------------ Force ----------------- */
Force::Force(Solver* s, Rigid* a, Rigid* b) : solver(s), bodyA(a), bodyB(b), nextA(nullptr), nextB(nullptr), next(nullptr), deviceKind(-1), listIndex(-1), serial(0)
{
    serial = s->impl->nextSerial++;
    next = s->forces; s->forces = this;
    if (a) { nextA = a->forces; a->forces = this; }
    if (b) { nextB = b->forces; b->forces = this; }
    for (int i = 0; i < MAX_CONSTRAINT_ROWS; ++i) { /* force.cpp:32-40 */
        stiffness[i] = 0.0f; lambda[i] = 0.0f; penalty[i] = 0.0f; motor[i] = 0.0f;
        fmin[i] = -FLT_MAX; fmax[i] = FLT_MAX; C[i] = 0.0f; fracture[i] = 0.0f;
    }
}

static void unlinkForce(Force* f)
{
    Solver* s = f->solver;
    Force** p = &s->forces;
    while (*p && *p != f) p = &(*p)->next;
    if (*p) *p = f->next;
    Rigid* ends[2] = {f->bodyA, f->bodyB};
    for (int e = 0; e < 2; ++e) {
        Rigid* b = ends[e];
        if (!b) continue;
        p = &b->forces;
        while (*p && *p != f) p = ((*p)->bodyA == b) ? &(*p)->nextA : &(*p)->nextB;
        if (*p) *p = (f->bodyA == b) ? f->nextA : f->nextB;
    }
}

Force::~Force()
{
    Solver* s = solver;
    /* (the stored kind, not isManifold(): inside a base-class destructor the virtual call resolves to Force::isManifold) */
    const bool manifold = deviceKind == AVBD_FORCE_MANIFOLD;
    if (!s->impl->clearing && deviceKind == AVBD_FORCE_HOST) {
        for (Force*& h : s->impl->hostForces)
            if (h == this) s->fail("deleting a user-defined Force that is registered with the device is not supported (clear() the solver instead)", AVBD_E_INVALID);
    }
    if (!s->impl->clearing && !s->impl->syncing && deviceKind != -2) {
        auto& v = s->impl->userForces;
        size_t k = v.size();
        if (!manifold) for (size_t i = 0; i < v.size(); ++i) if (v[i] == this) { k = i; break; }
        const bool userOnDevice = k < s->impl->flushedForces;
        if (manifold) {
            /* whatever happens below, syncForces() must never see this pointer again */
            auto& mm = s->impl->manifolds;
            for (size_t i = 0; i < mm.size(); ++i) if ((Force*)mm[i] == this) { mm.erase(mm.begin() + (long)i); break; }
        }
        if (userOnDevice || (manifold && listIndex >= 0)) {
            /* `delete force` of a force that lives on the device (force.cpp:31-69): rebuilt without it */
            if (listIndex < 0) s->syncForces();                         /* (a user force: positions of the device list) */
            int rc = avbd_remove_force(s->ctx, listIndex);
            if (rc) s->fail("avbd_remove_force", rc);
            if (userOnDevice) --s->impl->flushedForces;
            for (Force* f : v) f->listIndex = -1;
            for (Manifold* m : s->impl->manifolds) m->listIndex = -1;
        }
        if (k < v.size()) v.erase(v.begin() + (long)k);
    }
    unlinkForce(this);
}

void Force::deviceComputeConstraint(float alpha)
{
    Solver* s = solver;
    s->flush();
    if (listIndex < 0) s->syncForces();           /* mirrors are matched and updated IN PLACE: `this` survives */
    if (listIndex < 0) s->fail("computeConstraint on a force that no longer exists on the device", AVBD_E_INVALID);
    int rc = avbd_force_compute_constraint(s->ctx, listIndex, alpha, C, fmin, fmax, lambda);
    if (rc) s->fail("avbd_force_compute_constraint", rc);
    /* the reference's in-place mutations (manifold.cpp:185-186, 224-231, 241, 243: normal, cone-clamped lambda, stick,
     * penetration) come back through this one record; no mirror is created or destroyed here */
    AvbdForce r;
    if ((rc = avbd_download_force(s->ctx, listIndex, &r))) s->fail("avbd_download_force", rc);
    s->refreshMirror(this, r);
}

void Force::deviceComputeDerivatives(vec3& Jl, vec3& Ja, const Rigid* body, int row) const
{
    Solver* s = solver;
    s->flush();
    if (listIndex < 0) s->syncForces();
    float jl[3], ja[3];
    int rc = avbd_force_compute_derivatives(s->ctx, listIndex, body ? body->index : -1, row, jl, ja);
    if (rc) s->fail("avbd_force_compute_derivatives", rc);
    Jl = vec3(jl[0], jl[1], jl[2]);
    Ja = vec3(ja[0], ja[1], ja[2]);
}

Manifold::Manifold(Solver* s, Rigid* a, Rigid* b) : Force(s, a, b), numContacts(0), combinedFriction(0.0f)
{
    deviceKind = AVBD_FORCE_MANIFOLD;
    for (int i = 0; i < 4; ++i) { contacts[i].C0_n = 0.0f; contacts[i].C0_t = vec3(); contacts[i].stick = false; contacts[i].feature.value = 0; contacts[i].penetration = 0.0f; }
}

int Manifold::collide(Rigid* a, Rigid* b, Contact* out, bool flip)
{
    Solver* s = a->solver;
    AvbdContact c[4];
    int count = 0;
    int rc = avbd_kat_collide(s->ctx, 1, &a->position.x, &a->orientation.x, &a->size.x, &b->position.x, &b->orientation.x, &b->size.x, c, &count);
    if (rc) s->fail("avbd_kat_collide", rc);
    for (int i = 0; i < count; ++i) {
        out[i].feature.value = c[i].feature;
        out[i].rA = vec3(c[i].rA[0], c[i].rA[1], c[i].rA[2]); out[i].rB = vec3(c[i].rB[0], c[i].rB[1], c[i].rB[2]);
        out[i].normal = vec3(c[i].normal[0], c[i].normal[1], c[i].normal[2]);
        out[i].penetration = c[i].penetration; out[i].C0_n = 0.0f; out[i].C0_t = vec3(); out[i].stick = false;
        if (flip) { vec3 t = out[i].rA; out[i].rA = out[i].rB; out[i].rB = t; out[i].normal = -out[i].normal; } /* collision.cpp:479-486 */
    }
    return count;
}

/* ------------------------------------------------------------------ Joint / Spring / Ignore */
Joint::Joint(Solver* s, Rigid* a, Rigid* b, const vec3& localAnchorA, const vec3& localAnchorB, float linearStiffness,
             float angularStiffness_, float motor_, float fracture_)
    : Force(s, a, b), rA(localAnchorA), rB(localAnchorB)
{
    deviceKind = AVBD_FORCE_JOINT;
    quat qa = a ? a->orientation : quat();
    initialRelativeOrientation = conjugate(qa) * b->orientation;          /* joint.cpp:17-20 */
    for (int i = 0; i < 3; ++i) { stiffness[i] = linearStiffness; lambda[i] = 0; penalty[i] = PENALTY_MIN; }
    for (int i = 3; i < 6; ++i) { stiffness[i] = angularStiffness_; lambda[i] = 0; penalty[i] = PENALTY_MIN; }
    angularStiffness = angularStiffness_; angularMotor = motor_; angularFracture = fracture_;
    s->impl->userForces.push_back(this);
}

Joint::Joint(Solver* s, Rigid* b, const vec3& worldAnchor, float linearStiffness, float angularStiffness_, float motor_, float fracture_)
    : Force(s, nullptr, b)
{
    deviceKind = AVBD_FORCE_JOINT;
    rA = worldAnchor;                                                     /* joint.cpp:46-52 */
    rB = transpose(mat3_from_quat(b->orientation)) * (worldAnchor - b->position);
    initialRelativeOrientation = b->orientation;
    for (int i = 0; i < 3; ++i) { stiffness[i] = linearStiffness; lambda[i] = 0; penalty[i] = PENALTY_MIN; }
    for (int i = 3; i < 6; ++i) { stiffness[i] = angularStiffness_; lambda[i] = 0; penalty[i] = PENALTY_MIN; }
    angularStiffness = angularStiffness_; angularMotor = motor_; angularFracture = fracture_;
    s->impl->userForces.push_back(this);
}

Spring::Spring(Solver* s, Rigid* a, Rigid* b, const vec3& localAnchorA, const vec3& localAnchorB, float k, float rest)
    : Force(s, a, b), springStiffness(k), restLength(rest), rA(localAnchorA), rB(localAnchorB)
{
    deviceKind = AVBD_FORCE_SPRING;
    stiffness[0] = springStiffness;                                       /* spring.cpp:16-30 */
    if (restLength < 0) {
        vec3 pA = a->position + rotate(a->orientation, rA);
        vec3 pB = b->position + rotate(b->orientation, rB);
        restLength = length(pA - pB);
    }
    lambda[0] = 0.0f; penalty[0] = PENALTY_MIN; fmin[0] = -FLT_MAX; fmax[0] = FLT_MAX;
    s->impl->userForces.push_back(this);
}

IgnoreCollision::IgnoreCollision(Solver* s, Rigid* a, Rigid* b) : Force(s, a, b)
{
    deviceKind = AVBD_FORCE_IGNORE;
    s->impl->userForces.push_back(this);
}

/* ------------------------------------------------------------------ Solver ---------------- */
void Solver::fail(const char* what, int rc) const
{
    std::fprintf(stderr, "[avbd] fatal: %s (code %d): %s\n", what, rc, ctx ? avbd_last_error(ctx) : "no device context");
    std::fflush(stderr);
    std::abort();
}

void Solver::init(AvbdContext* borrowed)
{
    bodies = nullptr; forces = nullptr; enableDiagnostics = false; logFrequency = 60; stepIndex = 0;
    std::memset(&lastDiagnostics, 0, sizeof(lastDiagnostics));
    hostSync = true; mirrorManifolds = false; numBodies = 0; flushedBodies = 0;
    impl = new Impl();
    ctx = borrowed; ownsCtx = false;
    if (!ctx) {
        int device = 0;
        if (const char* d = std::getenv("AVBD_DEVICE")) device = std::atoi(d);
        int rc = avbd_create(device, &ctx);
        if (rc) { ctx = nullptr; fail("avbd_create: no MI355X device / HIP runtime available (this build has no CPU solve path)", rc); }
        ownsCtx = true;
    }
    defaultParams();
}

Solver::Solver() { init(nullptr); }
Solver::Solver(AvbdContext* borrowed) { init(borrowed); }

void Solver::releaseMirrors()
{
    impl->clearing = true;
    while (forces) delete forces;
    while (bodies) delete bodies;
    impl->clearing = false;
    impl->byIndex.clear(); impl->userForces.clear(); impl->hostForces.clear(); impl->flushedForces = 0; impl->shadow.clear(); impl->manifolds.clear();
    bodies = nullptr; forces = nullptr; numBodies = 0; flushedBodies = 0;
}

Solver::~Solver()
{
    /* the reference's destructor calls clear() (solver.cpp:140-143); the device scene of a borrowed
     * context outlives its mirror, an owned context is destroyed with everything in it */
    releaseMirrors();
    if (ownsCtx && ctx) avbd_destroy(ctx);
    delete impl;
}

Rigid* Solver::pick(const vec3& origin, const vec3& dir, vec3& local)      /* solver.cpp:145-228, evaluated on the device */
{
    flush();
    if (hostSync) pushEditedBodies();
    const float o[3] = {origin.x, origin.y, origin.z}, d[3] = {dir.x, dir.y, dir.z};
    float l[3] = {0.0f, 0.0f, 0.0f};
    int body = -1;
    int rc = avbd_pick(ctx, 1, o, d, &body, l);
    if (rc) fail("avbd_pick", rc);
    if (body < 0 || body >= (int)impl->byIndex.size()) return nullptr;
    local = vec3(l[0], l[1], l[2]);
    return impl->byIndex[(size_t)body];
}

void Solver::clear()                                                       /* solver.cpp:230-238 */
{
    releaseMirrors();
    stepIndex = 0;
    std::memset(&lastDiagnostics, 0, sizeof(lastDiagnostics));
    int rc = avbd_clear(ctx);
    if (rc) fail("avbd_clear", rc);
}

void Solver::defaultParams()                                               /* solver.cpp:240-253 */
{
    AvbdParams p;
    avbd_default_params(&p);
    dt = p.dt; gravity = vec3(p.gravity[0], p.gravity[1], p.gravity[2]); iterations = p.iterations;
    alpha = p.alpha; beta = p.beta; gamma = p.gamma; postStabilize = p.postStabilize != 0;
    if (logFrequency <= 0) logFrequency = 60;
    std::memset(&lastDiagnostics, 0, sizeof(lastDiagnostics));
}

Rigid* Solver::bodyByIndex(int i) const { return (i >= 0 && i < (int)impl->byIndex.size()) ? impl->byIndex[(size_t)i] : nullptr; }

static void packBody(const Rigid* b, AvbdBody& o)
{
    std::memcpy(o.position, &b->position.x, 12); std::memcpy(o.orientation, &b->orientation.x, 16);
    std::memcpy(o.linearVelocity, &b->linearVelocity.x, 12); std::memcpy(o.angularVelocity, &b->angularVelocity.x, 12);
    std::memcpy(o.prevLinearVelocity, &b->prevLinearVelocity.x, 12); std::memcpy(o.prevAngularVelocity, &b->prevAngularVelocity.x, 12);
    std::memcpy(o.initialPosition, &b->initialPosition.x, 12); std::memcpy(o.initialOrientation, &b->initialOrientation.x, 16);
    std::memcpy(o.inertialPosition, &b->inertialPosition.x, 12); std::memcpy(o.inertialOrientation, &b->inertialOrientation.x, 16);
    std::memcpy(o.size, &b->size.x, 12);
    o.mass = b->mass; o.invMass = b->invMass; o.friction = b->friction; o.radius = b->radius;
}

static void unpackBody(Rigid* b, const AvbdBody& o)
{
    std::memcpy(&b->position.x, o.position, 12); std::memcpy(&b->orientation.x, o.orientation, 16);
    std::memcpy(&b->linearVelocity.x, o.linearVelocity, 12); std::memcpy(&b->angularVelocity.x, o.angularVelocity, 12);
    std::memcpy(&b->prevLinearVelocity.x, o.prevLinearVelocity, 12); std::memcpy(&b->prevAngularVelocity.x, o.prevAngularVelocity, 12);
    std::memcpy(&b->initialPosition.x, o.initialPosition, 12); std::memcpy(&b->initialOrientation.x, o.initialOrientation, 16);
    std::memcpy(&b->inertialPosition.x, o.inertialPosition, 12); std::memcpy(&b->inertialOrientation.x, o.inertialOrientation, 16);
}

void Solver::flush()
{
    const int n = numBodies;
    if (n > flushedBodies) {
        const int k = n - flushedBodies;
        std::vector<float> size(3 * k), dens(k), fric(k), pos(3 * k), rot(4 * k), lv(3 * k), av(3 * k);
        for (int i = 0; i < k; ++i) {
            const Rigid* b = impl->byIndex[(size_t)(flushedBodies + i)];
            std::memcpy(&size[3 * i], &b->size.x, 12); dens[i] = b->density; fric[i] = b->friction;
            std::memcpy(&pos[3 * i], &b->position.x, 12); std::memcpy(&rot[4 * i], &b->orientation.x, 16);
            std::memcpy(&lv[3 * i], &b->linearVelocity.x, 12); std::memcpy(&av[3 * i], &b->angularVelocity.x, 12);
        }
        int first = -1;
        int rc = avbd_add_bodies(ctx, k, size.data(), dens.data(), fric.data(), pos.data(), rot.data(), lv.data(), av.data(), &first);
        if (rc) fail("avbd_add_bodies", rc);
        impl->shadow.resize((size_t)n);
        for (int i = flushedBodies; i < n; ++i) packBody(impl->byIndex[(size_t)i], impl->shadow[(size_t)i]);
        flushedBodies = n;
    }
    {   /* A Force that registered only with the base constructor is a user subclass: it joins the device's force list in creation
         * order like the built-in kinds (the order decides the per-body accumulation order), but its virtuals stay here — the
         * library asks for its rows before every visit (hostForceCallback below). */
        std::vector<Force*> fresh;
        for (Force* f = forces; f; f = f->next) if (f->deviceKind == -1) fresh.push_back(f);
        for (Force* f : fresh) { f->deviceKind = AVBD_FORCE_HOST; impl->userForces.push_back(f); }
        if (!fresh.empty()) {
            std::stable_sort(impl->userForces.begin() + (long)impl->flushedForces, impl->userForces.end(), [](const Force* a, const Force* b) { return a->serial < b->serial; });
            int rc = avbd_set_host_force_callback(ctx, &Solver::hostForceCallback, this);
            if (rc) fail("avbd_set_host_force_callback", rc);
        }
    }
    for (; impl->flushedForces < impl->userForces.size(); ++impl->flushedForces) {
        Force* f = impl->userForces[impl->flushedForces];
        if (f->deviceKind == AVBD_FORCE_HOST) {
            if (!f->bodyB) fail("a user-defined Force needs bodyB", AVBD_E_INVALID);
            int ord = -1;
            int rc = avbd_add_host_force(ctx, f->bodyA ? f->bodyA->index : -1, f->bodyB->index, &ord);
            if (rc) fail("avbd_add_host_force", rc);
            if ((size_t)ord != impl->hostForces.size()) fail("host force ordinal out of step", AVBD_E_INVALID);
            impl->hostForces.push_back(f);
            continue;
        }
        AvbdForce r;
        std::memset(&r, 0, sizeof(r));
        r.kind = f->deviceKind; r.bodyA = f->bodyA ? f->bodyA->index : -1; r.bodyB = f->bodyB ? f->bodyB->index : -1;
        r.rows = f->getRowCount();
        for (int i = 0; i < MAX_CONSTRAINT_ROWS; ++i) {
            r.C[i] = f->C[i]; r.fmin[i] = f->fmin[i]; r.fmax[i] = f->fmax[i]; r.lambda[i] = f->lambda[i];
            r.penalty[i] = f->penalty[i]; r.motor[i] = f->motor[i]; r.stiffness[i] = f->stiffness[i];
        }
        r.q0[3] = 1.0f;
        if (f->deviceKind == AVBD_FORCE_JOINT) {
            const Joint* j = static_cast<const Joint*>(f);
            std::memcpy(r.rA, &j->rA.x, 12); std::memcpy(r.rB, &j->rB.x, 12); std::memcpy(r.q0, &j->initialRelativeOrientation.x, 16);
        } else if (f->deviceKind == AVBD_FORCE_SPRING) {
            const Spring* sp = static_cast<const Spring*>(f);
            std::memcpy(r.rA, &sp->rA.x, 12); std::memcpy(r.rB, &sp->rB.x, 12); r.restLength = sp->restLength;
        }
        int rc = avbd_add_force(ctx, &r);
        if (rc) fail("avbd_add_force", rc);
    }
}

/* The library's window into a user-defined Force (include/avbd_hip.h: AvbdHostForceFn): it hands over the CURRENT poses of the
 * two endpoints, the mirror objects take them, and the subclass's own virtuals run exactly where Solver::step would call them. */
int Solver::hostForceCallback(void* user, int hostForce, int phase, float alphaNow, int body, const float* poseA, const float* poseB, AvbdHostRows* rows)
{
    Solver* s = static_cast<Solver*>(user);
    if (hostForce < 0 || (size_t)hostForce >= s->impl->hostForces.size()) return 1;
    Force* f = s->impl->hostForces[(size_t)hostForce];
    if (!f) return 1;
    auto take = [](Rigid* b, const float* p) {
        if (!b || !p) return;
        b->position = vec3(p[0], p[1], p[2]); b->orientation = quat(p[3], p[4], p[5], p[6]);
    };
    take(f->bodyA, poseA); take(f->bodyB, poseB);
    auto copyOut = [&]() {
        for (int i = 0; i < MAX_CONSTRAINT_ROWS; ++i) {
            rows->C[i] = f->C[i]; rows->fmin[i] = f->fmin[i]; rows->fmax[i] = f->fmax[i]; rows->lambda[i] = f->lambda[i];
            rows->penalty[i] = f->penalty[i]; rows->stiffness[i] = f->stiffness[i]; rows->motor[i] = f->motor[i];
        }
    };
    auto derivatives = [&](const Rigid* b, float (*Jl)[3], float (*Ja)[3]) {
        for (int row = 0; row < rows->rows; ++row) {
            vec3 jl, ja;
            f->computeDerivatives(jl, ja, b, row);
            Jl[row][0] = jl.x; Jl[row][1] = jl.y; Jl[row][2] = jl.z; Ja[row][0] = ja.x; Ja[row][1] = ja.y; Ja[row][2] = ja.z;
        }
    };
    switch (phase) {
    case AVBD_HOST_INITIALIZE:
        if (!f->initialize()) {
            /* solver.cpp:274-279: the solver deletes it.  rows = -1 tells the library, which drops the ordinal as this side does. */
            rows->rows = -1;
            Impl* im = s->impl;
            im->hostForces.erase(im->hostForces.begin() + hostForce);
            for (size_t k = 0; k < im->userForces.size(); ++k)
                if (im->userForces[k] == f) { im->userForces.erase(im->userForces.begin() + (long)k); if (k < im->flushedForces) --im->flushedForces; break; }
            for (Force* u : im->userForces) u->listIndex = -1;             /* the device's force list has changed */
            for (Manifold* m : im->manifolds) m->listIndex = -1;
            const bool was = im->syncing;
            im->syncing = true;                                            /* ~Force: the device side is taken care of, only unlink */
            delete f;
            im->syncing = was;
            return 0;
        }
        rows->rows = f->getRowCount();
        if (rows->rows > MAX_CONSTRAINT_ROWS) return 1;
        copyOut();
        return 0;
    case AVBD_HOST_STORE:
        for (int i = 0; i < MAX_CONSTRAINT_ROWS; ++i) { f->lambda[i] = rows->lambda[i]; f->penalty[i] = rows->penalty[i]; }
        return 0;
    case AVBD_HOST_PRIMAL: {
        f->computeConstraint(alphaNow);
        rows->rows = f->getRowCount();
        if (rows->rows > MAX_CONSTRAINT_ROWS) return 1;
        Rigid* b = s->bodyByIndex(body);
        if (b == f->bodyA) derivatives(b, rows->JlA, rows->JaA); else derivatives(b, rows->JlB, rows->JaB);
        copyOut();
        return 0;
    }
    case AVBD_HOST_DUAL:
        f->computeConstraint(alphaNow);
        rows->rows = f->getRowCount();
        if (rows->rows > MAX_CONSTRAINT_ROWS) return 1;
        if (f->bodyA) derivatives(f->bodyA, rows->JlA, rows->JaA);
        derivatives(f->bodyB, rows->JlB, rows->JaB);
        copyOut();
        return 0;
    }
    return 1;
}

void Solver::pushParams()
{
    AvbdParams p;
    p.dt = dt; p.gravity[0] = gravity.x; p.gravity[1] = gravity.y; p.gravity[2] = gravity.z; p.iterations = iterations;
    p.alpha = alpha; p.beta = beta; p.gamma = gamma; p.postStabilize = postStabilize ? 1 : 0;
    int rc = avbd_set_params(ctx, &p);
    if (rc) fail("avbd_set_params", rc);
}

void Solver::pushEditedBodies()
{
    const size_t n = impl->byIndex.size();
    bool edited = false;
    std::vector<AvbdBody> now(n);
    for (size_t i = 0; i < n; ++i) {
        packBody(impl->byIndex[i], now[i]);
        if (!edited && std::memcmp(&now[i], &impl->shadow[i], sizeof(float) * 33) != 0) edited = true;
    }
    if (!edited) return;
    int rc = avbd_upload_bodies(ctx, now.data(), (int)n);
    if (rc) fail("avbd_upload_bodies", rc);
    impl->shadow = now;
}

void Solver::pull()
{
    const size_t n = impl->byIndex.size();
    impl->shadow.resize(n);
    if (n) {
        int rc = avbd_download_bodies(ctx, impl->shadow.data(), (int)n);
        if (rc < 0) fail("avbd_download_bodies", rc);
        for (size_t i = 0; i < n; ++i) unpackBody(impl->byIndex[i], impl->shadow[i]);
    }
    {   /* what solver.cpp:51-66 prints while it steps: the kernels log it, the host prints it (same text, same order) */
        AvbdWarning w[128];
        int nw = avbd_get_warnings(ctx, w, 128);
        if (nw < 0) fail("avbd_get_warnings", nw);
        static const char* const label[6] = {"linear velocity", "angular velocity", "predicted position", "predicted orientation", "position", "orientation"};
        for (int i = 0; i < nw && i < 128; ++i) {
            const int id = (w[i].body >= 0 && (size_t)w[i].body < impl->byIndex.size()) ? impl->byIndex[(size_t)w[i].body]->id : w[i].body;
            if (w[i].kind == 3 || w[i].kind == 5)
                std::printf("[Physics] Warning: body %d produced non-finite %s; resetting to identity.\n", id, label[w[i].kind]);
            else
                std::printf("[Physics] Warning: body %d produced non-finite %s (%.3f, %.3f, %.3f); resetting to zero.\n", id, label[w[i].kind % 6],
                            w[i].value[0], w[i].value[1], w[i].value[2]);
        }
    }
    AvbdDiagnostics d;
    int rc = avbd_get_diagnostics(ctx, &d);
    if (rc) fail("avbd_get_diagnostics", rc);
    lastDiagnostics.maxPenetration = d.maxPenetration; lastDiagnostics.maxConstraintViolation = d.maxConstraintViolation;
    lastDiagnostics.maxLinearSpeed = d.maxLinearSpeed; lastDiagnostics.maxAngularSpeed = d.maxAngularSpeed;
    lastDiagnostics.maxNormalImpulse = d.maxNormalImpulse; lastDiagnostics.activeContacts = d.activeContacts;
    lastDiagnostics.activeManifolds = d.activeManifolds; lastDiagnostics.dynamicBodies = d.dynamicBodies;
    stepIndex = avbd_step_index(ctx);
}

void Solver::saveSnapshot(const char* path)
{
    flush();
    pushParams();
    if (hostSync) pushEditedBodies();
    int rc = avbd_save_snapshot(ctx, path);
    if (rc) fail("avbd_save_snapshot", rc);
}

void Solver::loadSnapshot(const char* path)
{
    clear();
    int rc = avbd_load_snapshot(ctx, path);
    if (rc) fail("avbd_load_snapshot", rc);
    const int n = avbd_num_bodies(ctx);
    std::vector<AvbdBody> st((size_t)(n > 0 ? n : 1));
    if (n > 0 && (rc = avbd_download_bodies(ctx, st.data(), n)) < 0) fail("avbd_download_bodies", rc);
    for (int i = 0; i < n; ++i) {
        const AvbdBody& o = st[(size_t)i];
        const float vol = o.size[0] * o.size[1] * o.size[2];
        /* the mirror's density is informational (mass = volume * density need not round-trip); the mass properties
         * themselves are taken from the record */
        Rigid* b = new Rigid(this, vec3(o.size[0], o.size[1], o.size[2]), vol > 0.0f ? o.mass / vol : 0.0f, o.friction,
                             vec3(o.position[0], o.position[1], o.position[2]));
        b->mass = o.mass; b->invMass = o.invMass; b->radius = o.radius;
        if (o.invMass > 0.0f) {
            float Ixx = (1.0f / 12.0f) * o.mass * (o.size[1] * o.size[1] + o.size[2] * o.size[2]);
            float Iyy = (1.0f / 12.0f) * o.mass * (o.size[0] * o.size[0] + o.size[2] * o.size[2]);
            float Izz = (1.0f / 12.0f) * o.mass * (o.size[0] * o.size[0] + o.size[1] * o.size[1]);
            b->inertiaTensor = mat3(vec3(Ixx, 0, 0), vec3(0, Iyy, 0), vec3(0, 0, Izz));
            b->invInertiaTensor = mat3(vec3(1.0f / Ixx, 0, 0), vec3(0, 1.0f / Iyy, 0), vec3(0, 0, 1.0f / Izz));
        }
        unpackBody(b, o);
    }
    flushedBodies = n;                                                  /* they already live on the device */
    impl->shadow = st;
    impl->shadow.resize((size_t)n);
    AvbdParams p;
    if ((rc = avbd_get_params(ctx, &p))) fail("avbd_get_params", rc);
    dt = p.dt; gravity = vec3(p.gravity[0], p.gravity[1], p.gravity[2]); iterations = p.iterations; alpha = p.alpha; beta = p.beta; gamma = p.gamma;
    postStabilize = p.postStabilize != 0;
    stepIndex = avbd_step_index(ctx);
}

void Solver::stepMany(int n)
{
    flush();
    pushParams();
    if (hostSync) pushEditedBodies();
    for (Manifold* m : impl->manifolds) m->listIndex = -1;
    for (Force* f : impl->userForces) f->listIndex = -1;
    int rc = avbd_step(ctx, n);
    if (rc) fail("avbd_step", rc);
    stepIndex += n;
    if (hostSync) pull();
    if (mirrorManifolds) syncForces();
}

void Solver::shardAcross(Solver** solvers, int n, int axis, bool exactCuts)
{
    if (!solvers || n < 1) return;
    Solver* s0 = solvers[0];
    for (int r = 0; r < n; ++r) { solvers[r]->flush(); solvers[r]->pushParams(); }
    const int nb = s0->numBodies;
    std::vector<AvbdBody> hb((size_t)std::max(nb, 1));
    if (nb) { int rc = avbd_download_bodies(s0->ctx, hb.data(), nb); if (rc < 0) s0->fail("avbd_download_bodies", rc); }
    std::vector<float> pos((size_t)3 * nb), inv((size_t)nb);
    for (int i = 0; i < nb; ++i) { pos[3 * i] = hb[i].position[0]; pos[3 * i + 1] = hb[i].position[1]; pos[3 * i + 2] = hb[i].position[2]; inv[i] = hb[i].invMass; }
    std::vector<int> owner((size_t)std::max(nb, 1), -1);
    int rc = avbd_partition_by_axis(nb, pos.data(), inv.data(), n, axis, owner.data());
    if (rc) s0->fail("avbd_partition_by_axis", rc);
    for (int r = 0; r < n; ++r) {
        if (solvers[r]->numBodies != nb) solvers[r]->fail("shardAcross: the solvers do not hold the same scene", AVBD_E_INVALID);
        if ((rc = avbd_set_ownership(solvers[r]->ctx, n, r, owner.data(), nb))) solvers[r]->fail("avbd_set_ownership", rc);
        if ((rc = avbd_set_shard_mode(solvers[r]->ctx, exactCuts ? 1 : 0))) solvers[r]->fail("avbd_set_shard_mode", rc);
    }
}

void Solver::rebalance(Solver** solvers, int n, int axis)
{
    if (!solvers || n < 1) return;
    Solver* s0 = solvers[0];
    const int nb = s0->numBodies;
    std::vector<AvbdBody> hb((size_t)std::max(nb, 1));
    if (nb) { int rc = avbd_download_bodies(s0->ctx, hb.data(), nb); if (rc < 0) s0->fail("avbd_download_bodies", rc); }
    std::vector<float> pos((size_t)3 * nb), inv((size_t)nb);
    for (int i = 0; i < nb; ++i) { pos[3 * i] = hb[i].position[0]; pos[3 * i + 1] = hb[i].position[1]; pos[3 * i + 2] = hb[i].position[2]; inv[i] = hb[i].invMass; }
    std::vector<int> owner((size_t)std::max(nb, 1), -1);
    int rc = avbd_partition_by_axis(nb, pos.data(), inv.data(), n, axis, owner.data());
    if (rc) s0->fail("avbd_partition_by_axis", rc);
    std::vector<AvbdContext*> ctxs((size_t)n);
    for (int r = 0; r < n; ++r) ctxs[(size_t)r] = solvers[r]->ctx;
    if ((rc = avbd_group_rebalance(ctxs.data(), n, owner.data(), nb))) s0->fail("avbd_group_rebalance", rc);
    for (int r = 0; r < n; ++r) {                                      /* the manifold mirrors refer to list positions of the old tables */
        for (Manifold* m : solvers[r]->impl->manifolds) m->listIndex = -1;
    }
}

void Solver::stepGroup(Solver** solvers, int n, int steps)
{
    if (!solvers || n < 1) return;
    std::vector<AvbdContext*> ctxs((size_t)n);
    for (int r = 0; r < n; ++r) {
        Solver* s = solvers[r];
        s->flush(); s->pushParams();
        if (s->hostSync) s->pushEditedBodies();
        for (Manifold* m : s->impl->manifolds) m->listIndex = -1;
        for (Force* f : s->impl->userForces) f->listIndex = -1;
        ctxs[(size_t)r] = s->ctx;
    }
    int rc = avbd_group_step(ctxs.data(), n, steps);
    if (rc) solvers[0]->fail("avbd_group_step", rc);
    for (int r = 0; r < n; ++r) { solvers[r]->stepIndex += steps; if (solvers[r]->hostSync) solvers[r]->pull(); }
}

void Solver::step()                                                        /* solver.cpp:255-514 */
{
    stepMany(1);
    if (enableDiagnostics) {                                               /* solver.cpp:499-513 */
        if (!hostSync) pull();
        int frequency = logFrequency > 0 ? logFrequency : 1;
        if (stepIndex % frequency == 0) {
            const Diagnostics& s = lastDiagnostics;
            std::printf("[Physics] step %d | manifolds: %d | contacts: %d | dyn bodies: %d | maxPen: %.6f | maxDrift: %.6f | maxLin: %.3f | maxAng: %.3f | maxLambda: %.3f\n",
                        stepIndex, s.activeManifolds, s.activeContacts, s.dynamicBodies, s.maxPenetration, s.maxConstraintViolation,
                        s.maxLinearSpeed, s.maxAngularSpeed, s.maxNormalImpulse);
        }
    }
}

void Solver::refreshMirror(Force* f, const AvbdForce& r)
{
    if (r.kind == AVBD_FORCE_HOST) return;                                /* a user-defined Force: the object IS the state */
    if (r.kind == AVBD_FORCE_MANIFOLD) {
        Manifold* m = static_cast<Manifold*>(f);
        m->numContacts = r.numContacts; m->combinedFriction = r.combinedFriction;
        for (int c = 0; c < r.numContacts; ++c) {
            const AvbdContact& s = r.contacts[c];
            Manifold::Contact& d = m->contacts[c];
            d.feature.value = s.feature; d.rA = vec3(s.rA[0], s.rA[1], s.rA[2]); d.rB = vec3(s.rB[0], s.rB[1], s.rB[2]);
            d.normal = vec3(s.normal[0], s.normal[1], s.normal[2]); d.penetration = s.penetration; d.C0_n = s.C0_n;
            d.C0_t = vec3(s.C0_t[0], s.C0_t[1], s.C0_t[2]); d.stick = s.stick != 0;
        }
    }
    for (int k = 0; k < MAX_CONSTRAINT_ROWS; ++k) {
        f->C[k] = r.C[k]; f->fmin[k] = r.fmin[k]; f->fmax[k] = r.fmax[k]; f->lambda[k] = r.lambda[k];
        f->penalty[k] = r.penalty[k]; f->motor[k] = r.motor[k]; f->stiffness[k] = r.stiffness[k];
    }
}

void Solver::syncForces()
{
    flush();
    int n = avbd_num_forces(ctx);
    if (n < 0) fail("avbd_num_forces", n);
    std::vector<AvbdForce> rec((size_t)n);
    if (n) {
        int rc = avbd_download_forces(ctx, rec.data(), n);
        if (rc < 0) fail("avbd_download_forces", rc);
    }
    /* Manifold mirrors are matched by their pair of bodies and UPDATED IN PLACE (a pointer the caller holds stays valid as
     * long as the manifold exists on the device, like in the reference, where step() deletes only separated manifolds);
     * every list link is rebuilt oldest -> newest with head insertion; mirrors of manifolds that are gone are deleted. */
    std::map<std::pair<const Rigid*, const Rigid*>, Manifold*> old;
    for (Manifold* m : impl->manifolds) old[std::make_pair((const Rigid*)m->bodyA, (const Rigid*)m->bodyB)] = m;
    impl->manifolds.clear();
    forces = nullptr;
    for (Rigid* b : impl->byIndex) b->forces = nullptr;
    size_t userLeft = impl->userForces.size();
    for (int i = n - 1; i >= 0; --i) {
        const AvbdForce& r = rec[(size_t)i];
        Force* f = nullptr;
        if (r.kind == AVBD_FORCE_MANIFOLD) {
            Rigid* a = bodyByIndex(r.bodyA); Rigid* b = bodyByIndex(r.bodyB);
            auto it = old.find(std::make_pair((const Rigid*)a, (const Rigid*)b));
            Manifold* m;
            if (it != old.end()) {
                m = it->second; old.erase(it);
                m->next = forces; forces = m;
                m->nextA = nullptr; m->nextB = nullptr;
                if (a) { m->nextA = a->forces; a->forces = m; }
                if (b) { m->nextB = b->forces; b->forces = m; }
            } else {
                m = new Manifold(this, a, b);                             /* the constructor links it at the three list heads */
            }
            impl->manifolds.push_back(m);
            f = m;
        } else {
            /* device generic forces are in creation order: the oldest unmatched user force is next */
            size_t k = impl->userForces.size() - userLeft;
            if (userLeft == 0) fail("device reports more user forces than the host created", AVBD_E_INVALID);
            f = impl->userForces[k];
            --userLeft;
            f->next = forces; forces = f;
            f->nextA = nullptr; f->nextB = nullptr;
            if (f->bodyA) { f->nextA = f->bodyA->forces; f->bodyA->forces = f; }
            if (f->bodyB) { f->nextB = f->bodyB->forces; f->bodyB->forces = f; }
        }
        f->listIndex = i;
        refreshMirror(f, r);
    }
    impl->syncing = true;                                                  /* ~Force: no device call, nothing left to unlink */
    for (auto& kv : old) delete kv.second;
    impl->syncing = false;
}

/* ------------------------------------------------------------------ C ABI glue ------------ */
extern "C" int avbd_build_scene(AvbdContext* ctx, const char* name)
{
    if (!ctx || !name) return AVBD_E_INVALID;
    Solver s(ctx);
    AvbdParams before;
    avbd_get_params(ctx, &before);
    s.dt = before.dt; s.gravity = vec3(before.gravity[0], before.gravity[1], before.gravity[2]); s.iterations = before.iterations;
    s.alpha = before.alpha; s.beta = before.beta; s.gamma = before.gamma; s.postStabilize = before.postStabilize != 0;
    int idx = buildSceneWithSuffix(&s, name);
    if (idx < 0) return AVBD_E_INVALID;
    s.flush();
    AvbdParams p;
    p.dt = s.dt; p.gravity[0] = s.gravity.x; p.gravity[1] = s.gravity.y; p.gravity[2] = s.gravity.z; p.iterations = s.iterations;
    p.alpha = s.alpha; p.beta = s.beta; p.gamma = s.gamma; p.postStabilize = s.postStabilize ? 1 : 0;
    avbd_set_params(ctx, &p);
    return idx; /* ~Solver drops the mirror objects; the device scene stays in the borrowed context */
}
