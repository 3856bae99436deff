// User-defined Force subclasses written against the reference's public API (solver.h:85-109) — the SAME text compiles against
// /root/reference/source/*.h (oracle/user_force_ref.cpp, which produces tests/golden/user_force_states.bin through
// tests/golden/make_golden.py) and against the host mirror avbd-demo3d_amd/host/*.h (tests/host/user_force.cpp), so the device run
// is pinned by what the reference itself computes for these classes.  Test infrastructure, not product.
//
// The includer provides: solver.h / joint.h / spring.h / ignorecollision.h of its side, and AVBD_SYNC_FORCES(solver) (brings the
// mirrors' rows up to date; nothing on the reference side).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

struct Tether : Force {                       // what a user would write against the reference's solver.h
    vec3 rA, rB; float rest; int calls = 0;
    Tether(Solver* s, Rigid* a, Rigid* b, const vec3& la, const vec3& lb, float k, float rest_) : Force(s, a, b), rA(la), rB(lb), rest(rest_)
    {
        stiffness[0] = k;
        if (rest < 0) rest = length((a->position + rotate(a->orientation, rA)) - (b->position + rotate(b->orientation, rB)));
        lambda[0] = 0.0f; penalty[0] = PENALTY_MIN; fmin[0] = -FLT_MAX; fmax[0] = FLT_MAX;
    }
    int getRowCount() const override { return 1; }
    bool initialize() override { return true; }
    vec3 delta() const { return (bodyA->position + rotate(bodyA->orientation, rA)) - (bodyB->position + rotate(bodyB->orientation, rB)); }
    void computeConstraint(float) override { ++calls; C[0] = length(delta()) - rest; }
    void computeDerivatives(vec3& Jl, vec3& Ja, const Rigid* body, int) const override
    {
        const vec3 d = delta();
        const float len = length(d);
        if (len < 1.0e-6f) { Jl = vec3(); Ja = vec3(); return; }
        const vec3 n = d / len;
        const float sign = body == bodyA ? 1.0f : -1.0f;
        Jl = n * sign;
        Ja = cross(body == bodyA ? rotate(body->orientation, rA) : rotate(body->orientation, rB), n) * sign;
    }
};

struct Pin : Force {                          // five rows: anchor of bodyB held at a world point (3 hard rows), two soft angular rows with a motor
    vec3 target, rB;
    Pin(Solver* s, Rigid* b, const vec3& world, const vec3& local) : Force(s, nullptr, b), target(world), rB(local)
    {
        for (int i = 0; i < 5; ++i) { lambda[i] = 0.0f; penalty[i] = PENALTY_MIN; fmin[i] = -FLT_MAX; fmax[i] = FLT_MAX; stiffness[i] = i < 3 ? FLT_MAX : 200.0f; }
        motor[4] = 0.25f;
    }
    int getRowCount() const override { return 5; }
    bool initialize() override { return true; }
    void computeConstraint(float) override
    {
        const vec3 d = target - (bodyB->position + rotate(bodyB->orientation, rB));
        C[0] = d.x; C[1] = d.y; C[2] = d.z;
        C[3] = 2.0f * bodyB->orientation.x; C[4] = 2.0f * bodyB->orientation.z;
    }
    void computeDerivatives(vec3& Jl, vec3& Ja, const Rigid* body, int row) const override
    {
        Jl = vec3(); Ja = vec3();
        if (body != bodyB) return;
        if (row < 3) {
            const vec3 ax(row == 0 ? 1.0f : 0.0f, row == 1 ? 1.0f : 0.0f, row == 2 ? 1.0f : 0.0f);
            Jl = ax * -1.0f;
            Ja = cross(rotate(body->orientation, rB), ax) * -1.0f;
        } else Ja = row == 3 ? vec3(1, 0, 0) : vec3(0, 0, 1);
    }
};

struct Fuse : Tether {                        // a tether that burns out: initialize() fails at its `life`-th step and the solver deletes it (solver.cpp:274-279)
    int life, inits = 0; int* destroyed;
    Fuse(Solver* s, Rigid* a, Rigid* b, const vec3& la, const vec3& lb, float k, int life_, int* destroyed_) : Tether(s, a, b, la, lb, k, -1.0f), life(life_), destroyed(destroyed_) {}
    ~Fuse() { ++*destroyed; }
    bool initialize() override { return ++inits < life; }
};

static void buildScene(Solver* s, Rigid*& a, Rigid*& b)
{
    s->clear(); s->defaultParams();
    new Rigid(s, {100, 1, 100}, 0.0f, 0.5f, {0, -0.5f, 0});
    a = new Rigid(s, {1, 1, 1}, 1.0f, 0.5f, {0.0f, 0.5f, 0.0f}, quat(0.0f, 0.1f, 0.0f, 0.99498744f));
    b = new Rigid(s, {1, 0.6f, 0.8f}, 2.0f, 0.4f, {0.2f, 1.9f, 0.1f}, quat(), {0.3f, 0.0f, -0.2f});
    new Rigid(s, {0.7f, 0.7f, 0.7f}, 1.0f, 0.6f, {0.1f, 1.15f, 0.0f});            // a cube between them: contacts share the bodies with the tether
    s->iterations = 8;
}


// every body's pose and velocities, newest first (the order of the solver's list)
static void recordBodies(Solver* s, std::vector<float>& out)
{
    for (Rigid* b = s->bodies; b; b = b->next) {
        const float v[13] = {b->position.x, b->position.y, b->position.z, b->orientation.x, b->orientation.y, b->orientation.z, b->orientation.w,
                             b->linearVelocity.x, b->linearVelocity.y, b->linearVelocity.z, b->angularVelocity.x, b->angularVelocity.y, b->angularVelocity.z};
        out.insert(out.end(), v, v + 13);
    }
}

// The pinned runs: Tether with finite and with infinite stiffness (40 steps each), then Pin + Joint + IgnoreCollision (60 steps);
// body states every 10 steps and the user force's rows at the end.  Last: two Fuses (and a Tether created between them) — the first
// burns out at step 12, the second at step 25; the solver deletes each at that step, the pair it constrained starts colliding.  Then two Fuses
// that burn out in the SAME step, first and last of the user forces of a scene that also holds a world-anchored Joint.
static void runUserForceScenes(std::vector<float>& out)
{
    for (float k : {400.0f, FLT_MAX}) {
        Solver* s = new Solver();
        Rigid *a, *b;
        buildScene(s, a, b);
        Tether* t = new Tether(s, a, b, {0.0f, 0.5f, 0.0f}, {0.0f, -0.3f, 0.0f}, k, -1.0f);
        for (int step = 0; step < 40; ++step) { s->step(); if (step % 10 == 9) recordBodies(s, out); }
        AVBD_SYNC_FORCES(s);
        out.push_back(t->lambda[0]); out.push_back(t->penalty[0]); out.push_back(t->C[0]); out.push_back((float)t->calls);
        delete s;
    }
    {
        Solver* s = new Solver();
        Rigid *a, *b;
        buildScene(s, a, b);
        new Joint(s, a, {0.0f, 0.2f, 0.0f});
        Pin* pin = new Pin(s, b, {0.2f, 2.2f, 0.1f}, {0.0f, 0.3f, 0.0f});
        new IgnoreCollision(s, a, b);
        for (int step = 0; step < 60; ++step) { s->step(); if (step % 10 == 9) recordBodies(s, out); }
        AVBD_SYNC_FORCES(s);
        for (int r = 0; r < 5; ++r) { out.push_back(pin->lambda[r]); out.push_back(pin->penalty[r]); out.push_back(pin->C[r]); }
        delete s;
    }
    {
        Solver* s = new Solver();
        Rigid *a, *b;
        buildScene(s, a, b);
        Rigid* c = s->bodies;                                               // the small cube between them (created last)
        int destroyed = 0;
        new Fuse(s, a, b, {0.0f, 0.5f, 0.0f}, {0.0f, -0.3f, 0.0f}, FLT_MAX, 12, &destroyed);
        Tether* t = new Tether(s, a, c, {0.3f, 0.0f, 0.0f}, {-0.2f, 0.0f, 0.0f}, 300.0f, -1.0f);
        new Fuse(s, c, b, {0.0f, 0.1f, 0.0f}, {0.0f, 0.0f, 0.2f}, 500.0f, 25, &destroyed);
        for (int step = 0; step < 40; ++step) {
            s->step();
            if (step % 10 == 9) {
                recordBodies(s, out);
                out.push_back((float)destroyed);
                int generic = 0;
                AVBD_SYNC_FORCES(s);
                for (Force* f = s->forces; f; f = f->next) generic += f->getRowCount() == 1;
                out.push_back((float)generic);
            }
        }
        AVBD_SYNC_FORCES(s);
        out.push_back(t->lambda[0]); out.push_back(t->penalty[0]); out.push_back(t->C[0]); out.push_back((float)t->calls);
        delete s;
        out.push_back((float)destroyed);
    }
    {   // two Fuses that burn out in the SAME step (the first and the last of three user forces), a Joint between them in the list
        Solver* s = new Solver();
        Rigid *a, *b;
        buildScene(s, a, b);
        Rigid* c = s->bodies;
        int destroyed = 0;
        new Fuse(s, a, b, {0.0f, 0.5f, 0.0f}, {0.0f, -0.3f, 0.0f}, 800.0f, 7, &destroyed);
        new Joint(s, c, {0.1f, 1.15f, 0.0f}, 50.0f, 20.0f);
        Tether* t = new Tether(s, a, c, {0.3f, 0.0f, 0.0f}, {-0.2f, 0.0f, 0.0f}, FLT_MAX, -1.0f);
        new Fuse(s, c, b, {0.0f, 0.1f, 0.0f}, {0.0f, 0.0f, 0.2f}, FLT_MAX, 7, &destroyed);
        for (int step = 0; step < 20; ++step) {
            s->step();
            if (step % 5 == 4) { recordBodies(s, out); out.push_back((float)destroyed); }
        }
        AVBD_SYNC_FORCES(s);
        out.push_back(t->lambda[0]); out.push_back(t->penalty[0]); out.push_back(t->C[0]); out.push_back((float)t->calls);
        delete s;
        out.push_back((float)destroyed);
    }
}
