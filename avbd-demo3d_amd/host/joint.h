/* joint.h — 6-row weld joint of the drop-in host API (reference: source/joint.h:14-47, joint.cpp). */
#pragma once
#include "solver.h"

struct Joint : Force {
    vec3 rA, rB;
    quat initialRelativeOrientation;
    float angularStiffness;
    float angularMotor;
    float angularFracture;

    /* body–body weld (joint.cpp:11-39) */
    Joint(Solver* solver, Rigid* bodyA, Rigid* bodyB, const vec3& localAnchorA, const vec3& localAnchorB,
          float linearStiffness = FLT_MAX, float angularStiffness = FLT_MAX, float motor = 0.0f, float fracture = FLT_MAX);
    /* body–world weld at a world-space anchor (joint.cpp:42-61) */
    Joint(Solver* solver, Rigid* bodyB, const vec3& worldAnchor, float linearStiffness = FLT_MAX,
          float angularStiffness = FLT_MAX, float motor = 0.0f, float fracture = FLT_MAX);

    int getRowCount() const override { return 6; }
    bool initialize() override { return true; }
    void computeConstraint(float alpha) override { deviceComputeConstraint(alpha); }
    void computeDerivatives(vec3& J_linear, vec3& J_angular, const Rigid* body, int row) const override
    {
        deviceComputeDerivatives(J_linear, J_angular, body, row);
    }
};
