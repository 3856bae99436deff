/*
 * oracle_capi.cpp — orc_* C entry points over the CPU restatement (TEST INFRASTRUCTURE).
 * Same shapes as the ref_* entry points of ref_shim.cpp so one Python harness drives both.
 */
#include "avbd_oracle.h"
#include "harness_api.h"

#include <cstdio>
#include <cstring>

using namespace orc;

static V3 v3p(const float* p) { V3 r = {p[0], p[1], p[2]}; return r; }
static Q4 q4p(const float* p) { Q4 r = {p[0], p[1], p[2], p[3]}; return r; }
static void put3(float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
static void put4(float* d, Q4 q) { d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; }

extern "C" {

void* orc_create() { return new World(); }
void orc_destroy(void* h) { delete (World*)h; }

void orc_get_params(void* h, HarnessParams* p)
{
    World* w = (World*)h;
    p->dt = w->dt; put3(p->gravity, w->gravity); p->iterations = w->iterations;
    p->alpha = w->alpha; p->beta = w->beta; p->gamma = w->gamma; p->postStabilize = w->postStabilize ? 1 : 0;
}

void orc_set_params(void* h, const HarnessParams* p)
{
    World* w = (World*)h;
    w->dt = p->dt; w->gravity = v3p(p->gravity); w->iterations = p->iterations;
    w->alpha = p->alpha; w->beta = p->beta; w->gamma = p->gamma; w->postStabilize = p->postStabilize != 0;
}

int orc_scene(void* h, const char* name) { return buildScene(*(World*)h, name); }

int orc_add_body(void* h, const float* size, float density, float friction, const float* pos, const float* q,
                 const float* lin, const float* ang)
{
    return ((World*)h)->addBody(v3p(size), density, friction, v3p(pos), q4p(q), v3p(lin), v3p(ang));
}

int orc_add_joint(void* h, int a, int b, const float* rA, const float* rB, float linStiff, float angStiff)
{
    ((World*)h)->addJoint(a, b, v3p(rA), v3p(rB), linStiff, angStiff);
    return 0;
}

int orc_add_spring(void* h, int a, int b, const float* rA, const float* rB, float stiffness, float rest)
{
    ((World*)h)->addSpring(a, b, v3p(rA), v3p(rB), stiffness, rest);
    return 0;
}

int orc_add_ignore(void* h, int a, int b) { ((World*)h)->addIgnore(a, b); return 0; }

void orc_step(void* h, int n)
{
    World* w = (World*)h;
    for (int i = 0; i < n; ++i) w->step();
}

int orc_step_index(void* h) { return ((World*)h)->stepIndex; }
int orc_num_bodies(void* h) { return (int)((World*)h)->bodies.size(); }

void orc_get_bodies(void* h, HarnessBody* out)
{
    World* w = (World*)h;
    for (size_t i = 0; i < w->bodies.size(); ++i) {
        const Body& b = w->bodies[i];
        HarnessBody& o = out[i];
        put3(o.pos, b.pos); put4(o.quat, b.rot); put3(o.linVel, b.linVel); put3(o.angVel, b.angVel);
        put3(o.prevLinVel, b.prevLinVel); put3(o.prevAngVel, b.prevAngVel);
        put3(o.initialPos, b.initPos); put4(o.initialQuat, b.initRot);
        put3(o.inertialPos, b.inerPos); put4(o.inertialQuat, b.inerRot);
        put3(o.size, b.size);
        o.mass = b.mass; o.invMass = b.invMass; o.friction = b.friction; o.radius = b.radius;
    }
}

/* Teacher forcing: overwrite the dynamic state of every body (pos, quat, velocities, prev velocities). */
void orc_set_bodies(void* h, const HarnessBody* in)
{
    World* w = (World*)h;
    for (size_t i = 0; i < w->bodies.size(); ++i) {
        Body& b = w->bodies[i];
        const HarnessBody& o = in[i];
        b.pos = v3p(o.pos); b.rot = q4p(o.quat); b.linVel = v3p(o.linVel); b.angVel = v3p(o.angVel);
        b.prevLinVel = v3p(o.prevLinVel); b.prevAngVel = v3p(o.prevAngVel);
    }
}

int orc_num_forces(void* h) { return (int)((World*)h)->forceList.size(); }

static void packForce(const World* w, const Force& f, HarnessForce& o)
{
    std::memset(&o, 0, sizeof(o));
    o.kind = f.kind; o.bodyA = f.bodyA; o.bodyB = f.bodyB; o.rows = w->rowCount(f);
    for (int r = 0; r < kMaxRows; ++r) {
        o.C[r] = f.C[r]; o.fmin[r] = f.fmin[r]; o.fmax[r] = f.fmax[r]; o.lambda[r] = f.lambda[r];
        o.penalty[r] = f.penalty[r]; o.motor[r] = f.motor[r]; o.stiffness[r] = f.stiffness[r];
    }
    if (f.kind == MANIFOLD) {
        o.numContacts = f.numContacts; o.combinedFriction = f.combinedFriction;
        for (int i = 0; i < f.numContacts; ++i) {
            const Contact& c = f.contacts[i];
            HarnessContact& d = o.contacts[i];
            d.feature = c.feature; put3(d.rA, c.rA); put3(d.rB, c.rB); put3(d.normal, c.normal);
            d.penetration = c.penetration; d.C0_n = c.C0_n; put3(d.C0_t, c.C0_t); d.stick = c.stick ? 1 : 0;
        }
    } else if (f.kind == JOINT) {
        put3(o.rA, f.rA); put3(o.rB, f.rB); put4(o.q0, f.q0);
    } else if (f.kind == SPRING) {
        put3(o.rA, f.rA); put3(o.rB, f.rB); o.restLength = f.restLength;
    }
}

/* Forces in the reference's solver-list order (newest first). */
void orc_get_forces(void* h, HarnessForce* out)
{
    World* w = (World*)h;
    int k = 0;
    for (size_t i = w->forceList.size(); i-- > 0; ++k) packForce(w, w->pool[w->forceList[i]], out[k]);
}

/* Teacher forcing: replace all forces by the given list (newest first, as orc_get_forces returns it). */
void orc_set_forces(void* h, const HarnessForce* in, int n)
{
    World* w = (World*)h;
    while (!w->forceList.empty()) w->deleteForce(w->forceList.back());
    for (int k = n - 1; k >= 0; --k) {
        const HarnessForce& o = in[k];
        int id = w->newForce(o.kind, o.bodyA, o.bodyB);
        Force& f = w->pool[id];
        for (int r = 0; r < kMaxRows; ++r) {
            f.C[r] = o.C[r]; f.fmin[r] = o.fmin[r]; f.fmax[r] = o.fmax[r]; f.lambda[r] = o.lambda[r];
            f.penalty[r] = o.penalty[r]; f.motor[r] = o.motor[r]; f.stiffness[r] = o.stiffness[r];
        }
        f.numContacts = o.numContacts; f.combinedFriction = o.combinedFriction;
        for (int i = 0; i < o.numContacts; ++i) {
            const HarnessContact& d = o.contacts[i];
            Contact& c = f.contacts[i];
            c.feature = d.feature; c.rA = v3p(d.rA); c.rB = v3p(d.rB); c.normal = v3p(d.normal);
            c.penetration = d.penetration; c.C0_n = d.C0_n; c.C0_t = v3p(d.C0_t); c.stick = d.stick != 0;
        }
        f.rA = v3p(o.rA); f.rB = v3p(o.rB); f.q0 = q4p(o.q0); f.restLength = o.restLength;
    }
}

void orc_set_step_index(void* h, int s) { ((World*)h)->stepIndex = s; }

/* row API (solver.h:102-108) on the list_index-th force of the solver list, newest first — same entry points as ref_shim.cpp */
int orc_force_compute_constraint(void* h, int list_index, float alpha)
{
    World* w = (World*)h;
    const int n = (int)w->forceList.size();
    if (list_index < 0 || list_index >= n) return -1;
    Force& f = w->pool[w->forceList[(size_t)(n - 1 - list_index)]];
    w->computeConstraint(f, alpha);
    return w->rowCount(f);
}

int orc_force_compute_derivatives(void* h, int list_index, int body, int row, float* Jl, float* Ja)
{
    World* w = (World*)h;
    const int n = (int)w->forceList.size();
    if (list_index < 0 || list_index >= n) return -1;
    const Force& f = w->pool[w->forceList[(size_t)(n - 1 - list_index)]];
    V3 jl, ja;
    w->computeDerivatives(f, jl, ja, body, row);
    Jl[0] = jl.x; Jl[1] = jl.y; Jl[2] = jl.z; Ja[0] = ja.x; Ja[1] = ja.y; Ja[2] = ja.z;
    return 0;
}

void orc_get_diag(void* h, HarnessDiag* d)
{
    const Diag& s = ((World*)h)->last;
    d->maxPenetration = s.maxPenetration; d->maxConstraintViolation = s.maxConstraintViolation;
    d->maxLinearSpeed = s.maxLinearSpeed; d->maxAngularSpeed = s.maxAngularSpeed;
    d->maxNormalImpulse = s.maxNormalImpulse; d->activeContacts = s.activeContacts;
    d->activeManifolds = s.activeManifolds; d->dynamicBodies = s.dynamicBodies;
}

void orc_set_diag_logging(void* h, int enable, int freq)
{
    World* w = (World*)h;
    w->enableDiagnostics = enable != 0; w->logFrequency = freq;
}

int orc_collide(const float* posA, const float* qA, const float* sizeA, const float* posB, const float* qB,
                const float* sizeB, HarnessContact* out)
{
    Body a, b;
    V3 z = {0, 0, 0};
    initBoxBody(a, v3p(sizeA), 1.0f, 0.5f, v3p(posA), q4p(qA), z, z);
    initBoxBody(b, v3p(sizeB), 1.0f, 0.5f, v3p(posB), q4p(qB), z, z);
    Contact c[4];
    int n = collide(a, b, c);
    for (int i = 0; i < n; ++i) {
        out[i].feature = c[i].feature; put3(out[i].rA, c[i].rA); put3(out[i].rB, c[i].rB); put3(out[i].normal, c[i].normal);
        out[i].penetration = c[i].penetration; out[i].C0_n = c[i].C0_n; put3(out[i].C0_t, c[i].C0_t); out[i].stick = c[i].stick ? 1 : 0;
    }
    return n;
}

int orc_pick(void* h, const float* origin, const float* dir, float* local)
{
    V3 l = {0.0f, 0.0f, 0.0f};
    V3 o = {origin[0], origin[1], origin[2]}, d = {dir[0], dir[1], dir[2]};
    int r = ((World*)h)->pick(o, d, l);
    local[0] = l.x; local[1] = l.y; local[2] = l.z;
    return r;
}

void orc_solve3(const float* A, const float* b, float* x)
{
    M3 M; M.c[0] = v3p(A); M.c[1] = v3p(A + 3); M.c[2] = v3p(A + 6);
    put3(x, solve3(M, v3p(b)));
}

void orc_solve6(const float* A, const float* b, float* x)
{
    M3 m[4];
    for (int k = 0; k < 4; ++k) { m[k].c[0] = v3p(A + 9 * k); m[k].c[1] = v3p(A + 9 * k + 3); m[k].c[2] = v3p(A + 9 * k + 6); }
    V3 xl, xa;
    solve6(m[0], m[1], m[2], m[3], v3p(b), v3p(b + 3), xl, xa);
    put3(x, xl); put3(x + 3, xa);
}

void orc_world_inertia(const float* size, float density, const float* q, float* I9, float* invI9)
{
    Body b; V3 z = {0, 0, 0};
    initBoxBody(b, v3p(size), density, 0.5f, z, q4p(q), z, z);
    M3 I = worldInertia(b), J = worldInvInertia(b);
    for (int c = 0; c < 3; ++c) { put3(I9 + 3 * c, I.c[c]); put3(invI9 + 3 * c, J.c[c]); }
}

/* Trace block in the wire format of main.cpp:228-244 (bodies newest first). */
void orc_print_step(void* h, int step)
{
    World* w = (World*)h;
    std::printf("Step %d:\n", step);
    for (size_t i = w->bodies.size(); i-- > 0;) {
        const Body& b = w->bodies[i];
        std::printf("  Body %d: Pos(%.4f, %.4f, %.4f)  ", b.id, b.pos.x, b.pos.y, b.pos.z);
        std::printf("Rot(%.4f, %.4f, %.4f, %.4f)  ", b.rot.x, b.rot.y, b.rot.z, b.rot.w);
        std::printf("LinVel(%.4f, %.4f, %.4f)  ", b.linVel.x, b.linVel.y, b.linVel.z);
        std::printf("AngVel(%.4f, %.4f, %.4f)\n", b.angVel.x, b.angVel.y, b.angVel.z);
    }
    const Diag& d = w->last;
    std::printf("  Diagnostics: manifolds=%d contacts=%d dynBodies=%d maxPen=%.6f maxDrift=%.6f maxLin=%.3f maxAng=%.3f maxLambda=%.3f\n",
                d.activeManifolds, d.activeContacts, d.dynamicBodies, d.maxPenetration, d.maxConstraintViolation,
                d.maxLinearSpeed, d.maxAngularSpeed, d.maxNormalImpulse);
}

} // extern "C"
