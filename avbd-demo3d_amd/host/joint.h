/* joint.h — host mirror of the reference's weld joint (source/joint.h:14-47, joint.cpp), MI355X build.
 *
 * The object is a handle: its six rows (three translational, three rotational) live in the device context as one GForce
 * record (csrc/avbd_types.h) and are evaluated there (csrc/avbd_rows.h: joint_compute_constraint / joint_compute_derivatives;
 * in the resident schedule one lane holds one half = three rows).  The members below exist so that scene code written
 * against the reference compiles unchanged; they are copied into the device record when the solver first steps.
 * Bug-compatible with the reference's row signs. */
#pragma once
#include "solver.h"

struct Joint : Force {
    /* ---- construction, same two forms as the reference ---- */
    Joint(Solver* owner, Rigid* a, Rigid* b, const vec3& anchorOnA, const vec3& anchorOnB,           /* two bodies (joint.cpp:11-39) */
          float linearStiffness = FLT_MAX, float angularStiffness = FLT_MAX, float motor = 0.0f, float fracture = FLT_MAX);
    Joint(Solver* owner, Rigid* b, const vec3& anchorInWorld,                                         /* body to world (joint.cpp:42-61) */
          float linearStiffness = FLT_MAX, float angularStiffness = FLT_MAX, float motor = 0.0f, float fracture = FLT_MAX);

    /* ---- Force interface: evaluated on the device ---- */
    int  getRowCount() const override { return 6; }
    bool initialize() override { return true; }
    void computeConstraint(float alpha) override { deviceComputeConstraint(alpha); }
    void computeDerivatives(vec3& Jlin, vec3& Jang, const Rigid* which, int row) const override { deviceComputeDerivatives(Jlin, Jang, which, row); }

    /* ---- public state of the reference's struct (joint.h:16-24) ---- */
    quat  initialRelativeOrientation;          /* conj(qA) * qB at creation */
    vec3  rA, rB;                              /* anchors in the bodies' frames (rA in world space for a world joint) */
    float angularStiffness, angularMotor, angularFracture;
};
