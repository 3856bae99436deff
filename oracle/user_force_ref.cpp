// oracle/user_force_ref.cpp — TEST INFRASTRUCTURE (never linked by the product).
// The user-defined Force subclasses of tests/host/user_forces.h compiled against the UNMODIFIED reference headers and sources
// (/root/reference/source, oracle/Makefile target `ref`): what the reference itself computes for a user's Tether / Pin.
// tests/golden/make_golden.py runs it and commits the floats as tests/golden/user_force_states.bin.
#include "solver.h"
#include "joint.h"
#include "spring.h"
#include "ignorecollision.h"
#define AVBD_SYNC_FORCES(s) ((void)(s))
#include "../tests/host/user_forces.h"
#include <cstdio>

int main(int argc, char** argv)
{
    std::vector<float> out;
    runUserForceScenes(out);
    FILE* f = std::fopen(argc > 1 ? argv[1] : "user_force_states.bin", "wb");
    if (!f) return 1;
    std::fwrite(out.data(), sizeof(float), out.size(), f);
    std::fclose(f);
    std::printf("%zu floats\n", out.size());
    return 0;
}
