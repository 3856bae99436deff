/*
 * harness_api.h — flat C records shared by the two CPU checkers under oracle/.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (avbd-demo3d_amd/, include/)
 * includes this file.  Two libraries export the same entry points with different
 * prefixes so tests can drive either through one Python harness:
 *
 *   ref_*  : oracle/_ref/libavbd_ref.so  — the UNMODIFIED reference sources under
 *            /root/reference/source compiled where they lie (ref_shim.cpp wraps them).
 *   orc_*  : oracle/liboracle.so         — our CPU restatement (avbd_oracle.cpp).
 *
 * Bodies are always addressed by creation index (0 = first body created), which is the
 * reverse of the reference's intrusive list order (rigid.cpp:20-21 prepends).
 */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

#define HB_FLOATS 40 /* floats per HarnessBody */

typedef struct HarnessBody {
    float pos[3];        /* Rigid::position            solver.h:56 */
    float quat[4];       /* Rigid::orientation (x,y,z,w) */
    float linVel[3];
    float angVel[3];
    float prevLinVel[3];
    float prevAngVel[3];
    float initialPos[3];
    float initialQuat[4];
    float inertialPos[3];
    float inertialQuat[4];
    float size[3];
    float mass, invMass, friction, radius;
} HarnessBody; /* 40 floats */

typedef struct HarnessContact {
    int feature;
    float rA[3], rB[3], normal[3];
    float penetration, C0_n, C0_t[3];
    int stick;
} HarnessContact;

enum { HK_MANIFOLD = 0, HK_JOINT = 1, HK_SPRING = 2, HK_IGNORE = 3 };

typedef struct HarnessForce {
    int kind;
    int bodyA, bodyB; /* creation index; -1 = world (nullptr) */
    int rows;
    float C[12], fmin[12], fmax[12], lambda[12], penalty[12], motor[12], stiffness[12];
    int numContacts;
    float combinedFriction;
    HarnessContact contacts[4];
    float rA[3], rB[3], q0[4], restLength; /* joint / spring */
} HarnessForce;

typedef struct HarnessDiag {
    float maxPenetration, maxConstraintViolation, maxLinearSpeed, maxAngularSpeed, maxNormalImpulse;
    int activeContacts, activeManifolds, dynamicBodies;
} HarnessDiag;

typedef struct HarnessParams {
    float dt, gravity[3];
    int iterations;
    float alpha, beta, gamma;
    int postStabilize;
} HarnessParams;

#ifdef __cplusplus
}
#endif
