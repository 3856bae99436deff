// User-defined Force subclasses through the row API (solver.h:102-108), rows evaluated on the host between launches.
// Built by tests/test_gpu_host_mirror.py against libavbd_hip.so; prints "OK <checks>" on success.
//
//  1. `Tether` re-states the distance constraint of the built-in Spring (one row) as a user subclass: a scene that uses it
//     must end with the SAME BITS as the scene that uses Spring — with finite stiffness (no dual) and with FLT_MAX (dual path).
//  2. `Pin` is a five-row user force (two entries of three rows): positional rows + one motor row, hard rows through the dual.
#include "solver.h"
#include "joint.h"
#include "spring.h"
#include "ignorecollision.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

static int checks = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } ++checks; } while (0)

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

static void buildScene(Solver* s, Rigid*& a, Rigid*& b)
{
    s->clear(); s->defaultParams();
    new Rigid(s, {100, 1, 100}, 0.0f, 0.5f, {0, -0.5f, 0});
    a = new Rigid(s, {1, 1, 1}, 1.0f, 0.5f, {0.0f, 0.5f, 0.0f}, quat(0.0f, 0.1f, 0.0f, 0.99498744f));
    b = new Rigid(s, {1, 0.6f, 0.8f}, 2.0f, 0.4f, {0.2f, 1.9f, 0.1f}, quat(), {0.3f, 0.0f, -0.2f});
    new Rigid(s, {0.7f, 0.7f, 0.7f}, 1.0f, 0.6f, {0.1f, 1.15f, 0.0f});            // a cube between them: contacts share the bodies with the tether
    s->iterations = 8;
}

static bool sameState(Solver* x, Solver* y)
{
    Rigid* p = x->bodies; Rigid* q = y->bodies;
    for (; p && q; p = p->next, q = q->next) {
        if (std::memcmp(&p->position, &q->position, sizeof(vec3)) || std::memcmp(&p->orientation, &q->orientation, sizeof(quat)) ||
            std::memcmp(&p->linearVelocity, &q->linearVelocity, sizeof(vec3)) || std::memcmp(&p->angularVelocity, &q->angularVelocity, sizeof(vec3))) return false;
    }
    return !p && !q;
}

int main()
{
    for (float k : {400.0f, FLT_MAX}) {       // finite stiffness: primal rows only; FLT_MAX: lambda / penalty through the dual as well
        Solver* u = new Solver(); Solver* r = new Solver();
        Rigid *ua, *ub, *ra, *rb;
        buildScene(u, ua, ub); buildScene(r, ra, rb);
        Tether* t = new Tether(u, ua, ub, {0.0f, 0.5f, 0.0f}, {0.0f, -0.3f, 0.0f}, k, -1.0f);
        Spring* sp = new Spring(r, ra, rb, {0.0f, 0.5f, 0.0f}, {0.0f, -0.3f, 0.0f}, k, -1.0f);
        CHECK(t->rest == sp->restLength);
        for (int step = 0; step < 40; ++step) {
            u->step(); r->step();
            if (!sameState(u, r)) { std::printf("FAILED: user Tether and built-in Spring (stiffness %g) differ after step %d\n", (double)k, step + 1); return 1; }
        }
        ++checks;
        r->syncForces();
        CHECK(std::memcmp(&t->lambda[0], &sp->lambda[0], 4) == 0 && std::memcmp(&t->penalty[0], &sp->penalty[0], 4) == 0);
        CHECK(std::memcmp(&t->C[0], &sp->C[0], 4) == 0);
        CHECK(t->calls == 40 * (2 * 8 + 8));                             // two visits + one dual per iteration, like Solver::step
        if (k == FLT_MAX) CHECK(t->lambda[0] != 0.0f);
        CHECK(u->lastDiagnostics.activeContacts == r->lastDiagnostics.activeContacts && u->lastDiagnostics.activeContacts > 0);
        delete u; delete r;
    }
    {   // five rows, world-anchored, with a joint created in between (creation order of mixed kinds) — runs, holds, stays finite
        Solver* s = new Solver();
        Rigid *a, *b;
        buildScene(s, a, b);
        Joint* j = new Joint(s, a, {0.0f, 0.2f, 0.0f});
        Pin* pin = new Pin(s, b, {0.2f, 2.2f, 0.1f}, {0.0f, 0.3f, 0.0f});
        new IgnoreCollision(s, a, b);
        for (int step = 0; step < 60; ++step) s->step();
        const vec3 anchor = b->position + rotate(b->orientation, pin->rB);
        CHECK(std::isfinite(anchor.x) && std::isfinite(anchor.y) && std::isfinite(anchor.z));
        CHECK(std::fabs(anchor.x - 0.2f) < 0.05f && std::fabs(anchor.y - 2.2f) < 0.05f && std::fabs(anchor.z - 0.1f) < 0.05f);   // the hard rows hold the anchor
        CHECK(pin->lambda[1] != 0.0f && pin->penalty[0] > PENALTY_MIN);
        s->syncForces();
        int seen = 0;
        for (Force* f = s->forces; f; f = f->next) seen += (f == pin) + (f == j);
        CHECK(seen == 2);
        delete s;
    }
    std::printf("OK %d\n", checks);
    return 0;
}
