/* ignorecollision.h — host mirror of the reference's collision filter (source/ignorecollision.h:11-23), MI355X build.
 *
 * A force without rows: it only makes `isConstrainedTo` true for its two bodies, so the broadphase never pairs them
 * (solver.cpp:266; on the device: the adjacency test of broadphase_pair_passes, csrc/avbd_kernels.hip). */
#pragma once
#include "solver.h"

struct IgnoreCollision : Force {
    IgnoreCollision(Solver* owner, Rigid* a, Rigid* b);
    int  getRowCount() const override { return 0; }
    bool initialize() override { return true; }
    void computeConstraint(float) override {}
    void computeDerivatives(vec3&, vec3&, const Rigid*, int) const override {}
};
