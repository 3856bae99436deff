/* ignorecollision.h — 0-row marker that keeps the broadphase from pairing two bodies
 * (reference: source/ignorecollision.h:11-23, solver.cpp:266). */
#pragma once
#include "solver.h"

struct IgnoreCollision : Force {
    IgnoreCollision(Solver* solver, Rigid* bodyA, Rigid* bodyB);
    int getRowCount() const override { return 0; }
    bool initialize() override { return true; }
    void computeConstraint(float) override {}
    void computeDerivatives(vec3&, vec3&, const Rigid*, int) const override {}
};
