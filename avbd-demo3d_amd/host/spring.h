/* spring.h — host mirror of the reference's distance spring (source/spring.h:13-37, spring.cpp), MI355X build.
 *
 * One row of finite stiffness between two anchors; like every built-in force it is a handle on a GForce record of the device
 * context and is evaluated there (csrc/avbd_rows.h: spring_compute_constraint / spring_compute_derivatives).  A negative rest
 * length means "the distance at creation" (spring.cpp:17-22). */
#pragma once
#include "solver.h"

struct Spring : Force {
    Spring(Solver* owner, Rigid* a, Rigid* b, const vec3& anchorOnA, const vec3& anchorOnB, float stiffness, float rest = -1.0f);

    int  getRowCount() const override { return 1; }
    bool initialize() override { return true; }
    void computeConstraint(float alpha) override { deviceComputeConstraint(alpha); }
    void computeDerivatives(vec3& Jlin, vec3& Jang, const Rigid* which, int row) const override { deviceComputeDerivatives(Jlin, Jang, which, row); }

    float springStiffness, restLength;         /* spring.h:17-18 */
    vec3  rA, rB;                              /* anchors in the bodies' frames */
};
