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

#define AVBD_SYNC_FORCES(s) (s)->syncForces()
#include "user_forces.h"

static bool sameState(Solver* x, Solver* y)
{
    Rigid* p = x->bodies; Rigid* q = y->bodies;
    for (; p && q; p = p->next, q = q->next) {
        if (std::memcmp(&p->position, &q->position, sizeof(vec3)) || std::memcmp(&p->orientation, &q->orientation, sizeof(quat)) ||
            std::memcmp(&p->linearVelocity, &q->linearVelocity, sizeof(vec3)) || std::memcmp(&p->angularVelocity, &q->angularVelocity, sizeof(vec3))) return false;
    }
    return !p && !q;
}

int main(int argc, char** argv)
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
    if (argc > 1) {   // the same classes compiled against the REFERENCE's headers produced this file (oracle/user_force_ref.cpp): bit for bit
        std::vector<float> mine;
        runUserForceScenes(mine);
        std::vector<float> ref(mine.size() + 1);
        FILE* f = std::fopen(argv[1], "rb");
        CHECK(f != nullptr);
        const size_t n = std::fread(ref.data(), sizeof(float), ref.size(), f);
        std::fclose(f);
        CHECK(n == mine.size());
        for (size_t i = 0; i < n; ++i)
            if (std::memcmp(&mine[i], &ref[i], 4)) { std::printf("FAILED: float %zu of the pinned user-force runs: device %a, reference %a\n", i, (double)mine[i], (double)ref[i]); return 1; }
        ++checks;
    }
    std::printf("OK %d\n", checks);
    return 0;
}
