// Host-mirror lifetime test (GPU): the row API called on mirrors never destroys them.
// Built by tests/test_gpu_host_mirror.py against libavbd_hip.so; prints "OK <checks>" on success.
#include "solver.h"
#include "joint.h"
#include "spring.h"
#include "ignorecollision.h"
#include "scenes.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

static int checks = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } ++checks; } while (0)

static std::vector<Manifold*> manifoldsOf(Solver* s)
{
    std::vector<Manifold*> v;
    for (Force* f = s->forces; f; f = f->next) if (f->isManifold()) v.push_back(static_cast<Manifold*>(f));
    return v;
}

int main()
{
    Solver* s = new Solver();
    // ---- contacts: computeConstraint / computeDerivatives on synced Manifold mirrors
    buildSceneByName(s, "Pyramid");
    for (int i = 0; i < 30; ++i) s->step();
    s->syncForces();
    std::vector<Manifold*> m0 = manifoldsOf(s);
    CHECK(m0.size() > 20);
    Manifold* m = m0[m0.size() / 2];
    Rigid* a = m->bodyA; Rigid* b = m->bodyB;
    const int rows = m->getRowCount();
    CHECK(rows >= 3 && rows <= 12);
    float before[12]; std::memcpy(before, m->lambda, sizeof(before));
    m->computeConstraint(s->alpha);                       // used to free `m` (syncForces deleted every mirror)
    CHECK(m->bodyA == a && m->bodyB == b && m->getRowCount() == rows);
    for (int r = 0; r < rows; ++r) CHECK(std::isfinite(m->C[r]) && m->fmin[r] <= m->fmax[r]);
    CHECK(m->fmax[0] == 0.0f && m->fmin[0] < 0.0f);        // normal row bounds (manifold.cpp:199-204)
    {   // still linked exactly once, same object
        int seen = 0;
        for (Force* f = s->forces; f; f = f->next) seen += f == m;
        CHECK(seen == 1);
    }
    vec3 JlA, JaA, JlB, JaB;
    m->computeDerivatives(JlA, JaA, a, 0);
    m->computeDerivatives(JlB, JaB, b, 0);
    CHECK(JlA.x == -JlB.x && JlA.y == -JlB.y && JlA.z == -JlB.z);
    // a second sync keeps the objects of manifolds that still exist
    s->syncForces();
    {
        std::vector<Manifold*> m1 = manifoldsOf(s);
        std::set<Manifold*> now(m1.begin(), m1.end());
        CHECK(m1.size() == m0.size());
        for (Manifold* x : m0) CHECK(now.count(x) == 1);
    }
    // stale mirror after a step: the row API re-syncs in place (listIndex < 0) instead of freeing the caller's object
    s->step();
    CHECK(m->listIndex < 0);
    m->computeConstraint(s->alpha);
    CHECK(m->listIndex >= 0 && m->bodyA == a && m->bodyB == b);
    // ---- `delete manifoldMirror` removes it from the device and from the mirror table (no double free on the next sync)
    const size_t nBefore = manifoldsOf(s).size();
    delete m;
    s->syncForces();
    CHECK(manifoldsOf(s).size() + 1 == nBefore);
    s->step();                                             // the broadphase re-creates the pair
    s->syncForces();
    CHECK(manifoldsOf(s).size() >= nBefore - 1);

    // ---- joints / springs: rows through the mirrors
    s->clear();
    s->defaultParams();
    new Rigid(s, {100, 1, 100}, 0.0f, 0.5f, {0, -0.5f, 0});
    Rigid* p = new Rigid(s, {1, 1, 1}, 1.0f, 0.5f, {0, 0.5f, 0});
    Rigid* q = new Rigid(s, {1, 1, 1}, 1.0f, 0.5f, {1.2f, 0.5f, 0});
    Joint* j = new Joint(s, p, q, {0.6f, 0, 0}, {-0.6f, 0, 0});
    Joint* jw = new Joint(s, q, {1.2f, 1.5f, 0});
    Spring* sp = new Spring(s, p, q, {0, 0.5f, 0}, {0, 0.5f, 0}, 500.0f);
    s->step();
    j->computeConstraint(s->alpha);
    jw->computeConstraint(s->alpha);
    sp->computeConstraint(s->alpha);
    for (int r = 0; r < 6; ++r) CHECK(std::isfinite(j->C[r]) && std::isfinite(jw->C[r]));
    CHECK(std::isfinite(sp->C[0]));
    vec3 Jl, Ja;
    j->computeDerivatives(Jl, Ja, p, 0);
    CHECK(Jl.x == 1.0f && Jl.y == 0.0f && Jl.z == 0.0f);   // joint.cpp:112-127: row 0 of body A is +x
    j->computeDerivatives(Jl, Ja, q, 4);
    CHECK(Ja.y == -1.0f);                                  // bug-compatible sign
    int users = 0;
    for (Force* f = s->forces; f; f = f->next) users += (f == j) + (f == jw) + (f == sp);
    CHECK(users == 3);
    delete s;
    std::printf("OK %d\n", checks);
    return 0;
}
