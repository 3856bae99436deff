/* spring.h — 1-row distance constraint of the drop-in host API (reference: source/spring.h:13-37, spring.cpp). */
#pragma once
#include "solver.h"

struct Spring : Force {
    vec3 rA, rB;
    float restLength;
    float springStiffness;

    Spring(Solver* solver, Rigid* bodyA, Rigid* bodyB, const vec3& localAnchorA, const vec3& localAnchorB, float stiffness,
           float rest = -1.0f);

    int getRowCount() const override { return 1; }
    bool initialize() override { return true; }
    void computeConstraint(float alpha) override { deviceComputeConstraint(alpha); }
    void computeDerivatives(vec3& J_linear, vec3& J_angular, const Rigid* body, int row) const override
    {
        deviceComputeDerivatives(J_linear, J_angular, body, row);
    }
};
