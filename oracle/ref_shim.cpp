/*
 * ref_shim.cpp — C API + headless driver around the UNMODIFIED reference sources.
 *
 * TEST INFRASTRUCTURE ONLY (container + GPU box as prebuilt oracle/_ref/ artefacts).
 * This file contains none of the reference's code: it #includes the reference translation
 * unit solver.cpp from where it lies (/root/reference/source) so that the anonymous-namespace
 * helpers solve6x6 / rowPenaltyGain (solver.cpp:68-125) are reachable for known-answer tests,
 * and links the other reference .cpp files compiled from /root/reference/source directly
 * (see oracle/Makefile, target _ref).  main.cpp cannot be built here (SDL2/imgui submodules
 * are empty), so the headless branch main.cpp:189-248 is reproduced in behaviour by
 * ref_print_step()/main() below: same flags, same printf wire format.
 */
#include "solver.cpp" /* resolved through -I/root/reference/source */
#include "joint.h"
#include "spring.h"
#include "ignorecollision.h"
#include "scenes.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "harness_api.h"

namespace {

struct RefWorld {
    Solver* solver = nullptr;
    std::vector<Rigid*> bodies; /* creation order */
};

static void rebuildBodyIndex(RefWorld* w)
{
    w->bodies.clear();
    for (Rigid* b = w->solver->bodies; b; b = b->next) w->bodies.push_back(b);
    /* list order is newest-first (rigid.cpp:20-21) → reverse to creation order */
    for (size_t i = 0, j = w->bodies.size(); i + 1 < j; ++i) { --j; std::swap(w->bodies[i], w->bodies[j]); }
}

static int indexOf(const RefWorld* w, const Rigid* b)
{
    if (!b) return -1;
    for (size_t i = 0; i < w->bodies.size(); ++i)
        if (w->bodies[i] == b) return (int)i;
    return -2;
}

static vec3 v3(const float* p) { return vec3(p[0], p[1], p[2]); }
static void put3(float* d, const vec3& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
static void put4(float* d, const quat& q) { d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; }

/* StressN(n): the build's generalisation of sceneStress1000 (scenes.h:87-132) with
 * NX=NY=NZ=n, expressed only through public reference constructors.  `packed` is the
 * build-defined contact-rich variant: spacingY=1.02, startY=0.52. */
static void buildStressN(Solver* s, int n, bool packed)
{
    s->clear();
    sceneGround(s);
    s->iterations = 20;
    s->beta = 30000.0f;
    s->gamma = 0.995f;
    const float spacingXZ = 1.15f;
    const float spacingY = packed ? 1.02f : 2.0f;
    const float startY = packed ? 0.52f : 20.0f;
    const float jitterXZ = 0.04f;
    const float jitterY = packed ? 0.0f : 0.25f;
    auto h01 = [](unsigned int x) {
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        return (x & 0x00FFFFFFU) / 16777215.0f;
    };
    for (int y = 0; y < n; ++y)
        for (int z = 0; z < n; ++z)
            for (int x = 0; x < n; ++x) {
                unsigned int seed = (unsigned int)(x + n * (z + n * y) + 1);
                float jx = (h01(seed * 9781U) * 2.0f - 1.0f) * jitterXZ;
                float jz = (h01(seed * 6271U) * 2.0f - 1.0f) * jitterXZ;
                float jy = h01(seed * 3343U) * jitterY;
                float px = (x - (n - 1) * 0.5f) * spacingXZ + jx;
                float py = startY + y * spacingY + jy;
                float pz = (z - (n - 1) * 0.5f) * spacingXZ + jz;
                new Rigid(s, {1.0f, 1.0f, 1.0f}, 1.0f, 0.5f, {px, py, pz}, quat(), {0, 0, 0}, {0, 0, 0});
            }
}

} // namespace

extern "C" {

void* ref_create()
{
    RefWorld* w = new RefWorld();
    w->solver = new Solver();
    return w;
}

void ref_destroy(void* h)
{
    RefWorld* w = (RefWorld*)h;
    delete w->solver;
    delete w;
}

void ref_get_params(void* h, HarnessParams* p)
{
    Solver* s = ((RefWorld*)h)->solver;
    p->dt = s->dt; put3(p->gravity, s->gravity); p->iterations = s->iterations;
    p->alpha = s->alpha; p->beta = s->beta; p->gamma = s->gamma; p->postStabilize = s->postStabilize ? 1 : 0;
}

void ref_set_params(void* h, const HarnessParams* p)
{
    Solver* s = ((RefWorld*)h)->solver;
    s->dt = p->dt; s->gravity = v3(p->gravity); s->iterations = p->iterations;
    s->alpha = p->alpha; s->beta = p->beta; s->gamma = p->gamma; s->postStabilize = p->postStabilize != 0;
}

/* Build one of the reference's own scenes (scenes.h:186-212) by exact name, or
 * "StressN:<n>" / "Packed:<n>".  Returns the scene index (>=0), 100 for StressN, -1 unknown. */
int ref_scene(void* h, const char* name)
{
    RefWorld* w = (RefWorld*)h;
    int idx = -1;
    if (std::strncmp(name, "StressN:", 8) == 0) { buildStressN(w->solver, std::atoi(name + 8), false); idx = 100; }
    else if (std::strncmp(name, "Packed:", 7) == 0) { buildStressN(w->solver, std::atoi(name + 7), true); idx = 101; }
    else {
        for (int i = 0; i < sceneCount; ++i)
            if (std::strcmp(sceneNames[i], name) == 0) { idx = i; break; }
        if (idx < 0) return -1;
        scenes[idx](w->solver);
    }
    rebuildBodyIndex(w);
    return idx;
}

int ref_add_body(void* h, const float* size, float density, float friction, const float* pos,
                 const float* q, const float* lin, const float* ang)
{
    RefWorld* w = (RefWorld*)h;
    Rigid* b = new Rigid(w->solver, v3(size), density, friction, v3(pos), quat(q[0], q[1], q[2], q[3]), v3(lin), v3(ang));
    w->bodies.push_back(b);
    return (int)w->bodies.size() - 1;
}

int ref_add_joint(void* h, int a, int b, const float* rA, const float* rB, float linStiff, float angStiff)
{
    RefWorld* w = (RefWorld*)h;
    if (a < 0) new Joint(w->solver, w->bodies[b], v3(rA), linStiff, angStiff);
    else new Joint(w->solver, w->bodies[a], w->bodies[b], v3(rA), v3(rB), linStiff, angStiff);
    return 0;
}

int ref_add_spring(void* h, int a, int b, const float* rA, const float* rB, float stiffness, float rest)
{
    RefWorld* w = (RefWorld*)h;
    new Spring(w->solver, w->bodies[a], w->bodies[b], v3(rA), v3(rB), stiffness, rest);
    return 0;
}

int ref_add_ignore(void* h, int a, int b)
{
    RefWorld* w = (RefWorld*)h;
    new IgnoreCollision(w->solver, w->bodies[a], w->bodies[b]);
    return 0;
}

void ref_step(void* h, int n)
{
    Solver* s = ((RefWorld*)h)->solver;
    for (int i = 0; i < n; ++i) s->step();
}

int ref_step_index(void* h) { return ((RefWorld*)h)->solver->stepIndex; }
int ref_num_bodies(void* h) { return (int)((RefWorld*)h)->bodies.size(); }

void ref_get_bodies(void* h, HarnessBody* out)
{
    RefWorld* w = (RefWorld*)h;
    for (size_t i = 0; i < w->bodies.size(); ++i) {
        const Rigid* b = w->bodies[i];
        HarnessBody& o = out[i];
        put3(o.pos, b->position); put4(o.quat, b->orientation);
        put3(o.linVel, b->linearVelocity); put3(o.angVel, b->angularVelocity);
        put3(o.prevLinVel, b->prevLinearVelocity); put3(o.prevAngVel, b->prevAngularVelocity);
        put3(o.initialPos, b->initialPosition); put4(o.initialQuat, b->initialOrientation);
        put3(o.inertialPos, b->inertialPosition); put4(o.inertialQuat, b->inertialOrientation);
        put3(o.size, b->size);
        o.mass = b->mass; o.invMass = b->invMass; o.friction = b->friction; o.radius = b->radius;
    }
}

int ref_num_forces(void* h)
{
    int n = 0;
    for (Force* f = ((RefWorld*)h)->solver->forces; f; f = f->next) ++n;
    return n;
}

/* Forces in solver-list order (newest first, force.cpp:16-18). */
void ref_get_forces(void* h, HarnessForce* out)
{
    RefWorld* w = (RefWorld*)h;
    int k = 0;
    for (Force* f = w->solver->forces; f; f = f->next, ++k) {
        HarnessForce& o = out[k];
        std::memset(&o, 0, sizeof(o));
        o.bodyA = indexOf(w, f->bodyA);
        o.bodyB = indexOf(w, f->bodyB);
        o.rows = f->getRowCount();
        for (int r = 0; r < MAX_CONSTRAINT_ROWS; ++r) {
            o.C[r] = f->C[r]; o.fmin[r] = f->fmin[r]; o.fmax[r] = f->fmax[r]; o.lambda[r] = f->lambda[r];
            o.penalty[r] = f->penalty[r]; o.motor[r] = f->motor[r]; o.stiffness[r] = f->stiffness[r];
        }
        if (Manifold* m = dynamic_cast<Manifold*>(f)) {
            o.kind = HK_MANIFOLD;
            o.numContacts = m->numContacts;
            o.combinedFriction = m->combinedFriction;
            for (int i = 0; i < m->numContacts; ++i) {
                const Manifold::Contact& c = m->contacts[i];
                HarnessContact& d = o.contacts[i];
                d.feature = c.feature.value; put3(d.rA, c.rA); put3(d.rB, c.rB); put3(d.normal, c.normal);
                d.penetration = c.penetration; d.C0_n = c.C0_n; put3(d.C0_t, c.C0_t); d.stick = c.stick ? 1 : 0;
            }
        } else if (Joint* j = dynamic_cast<Joint*>(f)) {
            o.kind = HK_JOINT; put3(o.rA, j->rA); put3(o.rB, j->rB); put4(o.q0, j->initialRelativeOrientation);
        } else if (Spring* sp = dynamic_cast<Spring*>(f)) {
            o.kind = HK_SPRING; put3(o.rA, sp->rA); put3(o.rB, sp->rB); o.restLength = sp->restLength;
        } else {
            o.kind = HK_IGNORE;
        }
    }
}

/* The row API of the reference called through its own virtuals (solver.h:102-108) on the list_index-th force of the
 * solver list (newest first).  computeConstraint mutates the force in place exactly as inside Solver::step(); the caller
 * reads the result back with ref_get_forces. */
int ref_force_compute_constraint(void* h, int list_index, float alpha)
{
    RefWorld* w = (RefWorld*)h;
    int k = 0;
    for (Force* f = w->solver->forces; f; f = f->next, ++k)
        if (k == list_index) { f->computeConstraint(alpha); return f->getRowCount(); }
    return -1;
}

int ref_force_compute_derivatives(void* h, int list_index, int body, int row, float* Jl, float* Ja)
{
    RefWorld* w = (RefWorld*)h;
    int k = 0;
    for (Force* f = w->solver->forces; f; f = f->next, ++k)
        if (k == list_index) {
            vec3 jl, ja;
            f->computeDerivatives(jl, ja, (body >= 0 && body < (int)w->bodies.size()) ? w->bodies[(size_t)body] : nullptr, row);
            put3(Jl, jl); put3(Ja, ja);
            return 0;
        }
    return -1;
}

void ref_get_diag(void* h, HarnessDiag* d)
{
    const Solver::Diagnostics& s = ((RefWorld*)h)->solver->lastDiagnostics;
    d->maxPenetration = s.maxPenetration; d->maxConstraintViolation = s.maxConstraintViolation;
    d->maxLinearSpeed = s.maxLinearSpeed; d->maxAngularSpeed = s.maxAngularSpeed;
    d->maxNormalImpulse = s.maxNormalImpulse; d->activeContacts = s.activeContacts;
    d->activeManifolds = s.activeManifolds; d->dynamicBodies = s.dynamicBodies;
}

void ref_set_diag_logging(void* h, int enable, int freq)
{
    Solver* s = ((RefWorld*)h)->solver;
    s->enableDiagnostics = enable != 0;
    s->logFrequency = freq;
}

/* ---- known-answer entry points ------------------------------------------------------- */

/* Manifold::collide (collision.cpp:420) on two free-standing boxes. */
int ref_collide(const float* posA, const float* qA, const float* sizeA, const float* posB, const float* qB,
                const float* sizeB, HarnessContact* out)
{
    Solver tmp;
    Rigid* a = new Rigid(&tmp, v3(sizeA), 1.0f, 0.5f, v3(posA), quat(qA[0], qA[1], qA[2], qA[3]));
    Rigid* b = new Rigid(&tmp, v3(sizeB), 1.0f, 0.5f, v3(posB), quat(qB[0], qB[1], qB[2], qB[3]));
    Manifold::Contact c[4];
    int n = Manifold::collide(a, b, c, false);
    for (int i = 0; i < n; ++i) {
        out[i].feature = c[i].feature.value; put3(out[i].rA, c[i].rA); put3(out[i].rB, c[i].rB);
        put3(out[i].normal, c[i].normal); out[i].penetration = c[i].penetration; out[i].C0_n = c[i].C0_n;
        put3(out[i].C0_t, c[i].C0_t); out[i].stick = c[i].stick ? 1 : 0;
    }
    return n; /* tmp's destructor deletes a and b (solver.cpp:230-238) */
}

/* maths.h:104 `solve`; A is column-major 3x3. */
void ref_solve3(const float* A, const float* b, float* x)
{
    mat3 M(v3(A), v3(A + 3), v3(A + 6));
    put3(x, solve(M, v3(b)));
}

/* solver.cpp:68-83 `solve6x6`; A = {ll, la, al, aa}, each column-major 3x3. */
void ref_solve6(const float* A, const float* b, float* x)
{
    mat66 M;
    M.ll = mat3(v3(A), v3(A + 3), v3(A + 6));
    M.la = mat3(v3(A + 9), v3(A + 12), v3(A + 15));
    M.al = mat3(v3(A + 18), v3(A + 21), v3(A + 24));
    M.aa = mat3(v3(A + 27), v3(A + 30), v3(A + 33));
    vec6 r = solve6x6(M, vec6{v3(b), v3(b + 3)});
    put3(x, r.l); put3(x + 3, r.a);
}

/* Solver::pick (solver.cpp:145-228) → creation index of the picked body or -1. */
int ref_pick(void* h, const float* origin, const float* dir, float* local)
{
    RefWorld* w = (RefWorld*)h;
    vec3 l(0.0f, 0.0f, 0.0f);
    Rigid* b = w->solver->pick(v3(origin), v3(dir), l);
    put3(local, l);
    return b ? indexOf(w, b) : -1;
}

/* Body-frame → world inertia (rigid.cpp:51-59) for a box body at a given orientation. */
void ref_world_inertia(const float* size, float density, const float* q, float* I9, float* invI9)
{
    Solver tmp;
    Rigid* b = new Rigid(&tmp, v3(size), density, 0.5f, vec3(), quat(q[0], q[1], q[2], q[3]));
    mat3 I = b->getInertiaTensorWorld(), J = b->getInvInertiaTensorWorld();
    for (int c = 0; c < 3; ++c) { put3(I9 + 3 * c, I.cols[c]); put3(invI9 + 3 * c, J.cols[c]); }
}

/* One trace block in the wire format of main.cpp:228-244. */
void ref_print_step(void* h, int step)
{
    Solver* s = ((RefWorld*)h)->solver;
    printf("Step %d:\n", step);
    for (Rigid* body = s->bodies; body != nullptr; body = body->next) {
        printf("  Body %d: Pos(%.4f, %.4f, %.4f)  ", body->id, body->position.x, body->position.y, body->position.z);
        printf("Rot(%.4f, %.4f, %.4f, %.4f)  ", body->orientation.x, body->orientation.y, body->orientation.z, body->orientation.w);
        printf("LinVel(%.4f, %.4f, %.4f)  ", body->linearVelocity.x, body->linearVelocity.y, body->linearVelocity.z);
        printf("AngVel(%.4f, %.4f, %.4f)\n", body->angularVelocity.x, body->angularVelocity.y, body->angularVelocity.z);
    }
    const Solver::Diagnostics& d = s->lastDiagnostics;
    printf("  Diagnostics: manifolds=%d contacts=%d dynBodies=%d maxPen=%.6f maxDrift=%.6f maxLin=%.3f maxAng=%.3f maxLambda=%.3f\n",
           d.activeManifolds, d.activeContacts, d.dynamicBodies, d.maxPenetration, d.maxConstraintViolation,
           d.maxLinearSpeed, d.maxAngularSpeed, d.maxNormalImpulse);
}

} // extern "C"

#ifdef REF_SHIM_MAIN
/* Headless driver: --nogfx/--headless, --scene/-s, --steps/-n as main.cpp:195-203, plus
 * build-only --quiet (no trace, prints ms/step) and --dump <file> (raw HarnessBody array after the
 * last step).  Unknown scene names fall back to index 0 "Empty" like main.cpp:211-219. */
int main(int argc, char** argv)
{
    const char* requested = nullptr;
    const char* dumpPath = nullptr;
    int steps = 300;
    bool quiet = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--nogfx") || !strcmp(argv[i], "--headless")) { /* always headless */ }
        else if ((!strcmp(argv[i], "--scene") || !strcmp(argv[i], "-s")) && i + 1 < argc) requested = argv[++i];
        else if ((!strcmp(argv[i], "--steps") || !strcmp(argv[i], "-n")) && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--quiet")) quiet = true;
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dumpPath = argv[++i];
    }
    void* w = ref_create();
    ref_set_diag_logging(w, quiet ? 0 : 1, 1);
    const char* shownName = sceneNames[0];
    int idx = requested ? ref_scene(w, requested) : -1;
    if (idx < 0) { ref_scene(w, sceneNames[0]); }
    else if (idx < sceneCount) shownName = sceneNames[idx];
    else shownName = requested;

    if (!quiet) printf("Running in headless mode: scene '%s', steps=%d\n", shownName, steps);
    auto t0 = std::chrono::steady_clock::now();
    for (int step = 0; step < steps; ++step) {
        ref_step(w, 1);
        if (!quiet) ref_print_step(w, step);
    }
    auto t1 = std::chrono::steady_clock::now();
    if (quiet) {
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        printf("{\"scene\": \"%s\", \"steps\": %d, \"ms_total\": %.3f, \"ms_per_step\": %.6f}\n", shownName, steps, ms, steps ? ms / steps : 0.0);
    }
    if (dumpPath) {
        int n = ref_num_bodies(w);
        std::vector<HarnessBody> hb(n);
        ref_get_bodies(w, hb.data());
        FILE* f = fopen(dumpPath, "wb");
        if (f) { fwrite(hb.data(), sizeof(HarnessBody), n, f); fclose(f); }
    }
    ref_destroy(w);
    return 0;
}
#endif
