/*
 * scenes.h — scene builders of the drop-in host API.
 *
 * Same table as the reference (source/scenes.h:186-212: names, order, count) built through the same
 * public constructors, plus the build's StressN(n) generalisation of sceneStress1000
 * (scenes.h:87-132 with NX=NY=NZ=n; n=10 is the reference scene bit for bit) and its contact-rich
 * "Packed" variant used for measurement windows (not a reference scene).
 */
#pragma once

#include "solver.h"
#include "joint.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Joint;
struct Spring;
struct IgnoreCollision;

static void sceneEmpty(Solver* solver) { solver->clear(); }

static void sceneGround(Solver* solver)
{
    solver->clear();
    new Rigid(solver, {100, 1, 100}, 0.0f, 0.5f, {0, -0.5f, 0});
}

static void sceneStack(Solver* solver)
{
    sceneGround(solver);
    for (int i = 0; i < 10; ++i)
        new Rigid(solver, {1, 1, 1}, 1.0f, 0.5f, {0, i * 1.1f + 0.5f, 0});
}

static void scenePyramid(Solver* solver)
{
    sceneGround(solver);
    const int rows = 10;
    for (int y = 0; y < rows; ++y) {
        const int width = rows - y;
        for (int x = 0; x < width; ++x)
            new Rigid(solver, {1, 1, 1}, 1.0f, 0.5f, {(x - (width - 1) * 0.5f) * 1.1f, y * 1.05f + 0.5f, 0});
    }
}

static void sceneWall(Solver* solver)
{
    sceneGround(solver);
    const int columns = 8, courses = 8;
    const vec3 brick = {1.0f, 0.5f, 0.5f};
    const float pitchX = 1.03f, pitchY = 0.52f;
    const float y0 = brick.y * 0.5f;
    for (int course = 0; course < courses; ++course) {
        const float stagger = (course % 2 == 0) ? 0.0f : 0.5f * pitchX;
        for (int col = 0; col < columns; ++col) {
            float x = (col - (columns - 1) * 0.5f) * pitchX + stagger;
            float y = course * pitchY + y0;
            new Rigid(solver, brick, 1.0f, 0.4f, {x, y, -5});
        }
    }
}

static void sceneTwoBlockDrop(Solver* solver)
{
    sceneGround(solver);
    new Rigid(solver, {1.0f, 1.0f, 1.0f}, 1.0f, 0.5f, {0.0f, 0.5f, 0.0f});
    quat tilt(vec3(0.0f, 0.0f, 1.0f), 0.45f);
    new Rigid(solver, {1.0f, 1.0f, 1.0f}, 1.0f, 0.5f, {0.18f, 2.2f, 0.0f}, tilt, {0, 0, 0}, {0.0f, 0.0f, 1.0f});
}

/* integer hash → [0,1], scenes.h:108-115 */
static inline float avbdSceneHash01(unsigned int v)
{
    v ^= v >> 16; v *= 0x7feb352dU;
    v ^= v >> 15; v *= 0x846ca68bU;
    v ^= v >> 16;
    return (v & 0x00FFFFFFU) / 16777215.0f;
}

/* nx*ny*nz unit cubes dropped on the ground slab (scenes.h:87-132 generalised: the reference scene is 10,10,10);
 * `packed` starts them resting (contacts from step 1). */
static void sceneStressGrid(Solver* solver, int nx, int ny, int nz, bool packed = false)
{
    sceneGround(solver);
    solver->iterations = 20;
    solver->beta = 30000.0f;
    solver->gamma = 0.995f;
    const float pitchXZ = 1.15f;
    const float pitchY = packed ? 1.02f : 2.0f;
    const float baseY = packed ? 0.52f : 20.0f;
    const float wobbleXZ = 0.04f;
    const float wobbleY = packed ? 0.0f : 0.25f;
    for (int y = 0; y < ny; ++y)
        for (int z = 0; z < nz; ++z)
            for (int x = 0; x < nx; ++x) {
                unsigned int seed = (unsigned int)(x + nx * (z + nz * y) + 1);
                float jx = (avbdSceneHash01(seed * 9781U) * 2.0f - 1.0f) * wobbleXZ;
                float jz = (avbdSceneHash01(seed * 6271U) * 2.0f - 1.0f) * wobbleXZ;
                float jy = avbdSceneHash01(seed * 3343U) * wobbleY;
                float px = (x - (nx - 1) * 0.5f) * pitchXZ + jx;
                float py = baseY + y * pitchY + jy;
                float pz = (z - (nz - 1) * 0.5f) * pitchXZ + jz;
                new Rigid(solver, {1.0f, 1.0f, 1.0f}, 1.0f, 0.5f, {px, py, pz});
            }
}

static void sceneStressN(Solver* solver, int n, bool packed = false) { sceneStressGrid(solver, n, n, n, packed); }

static void sceneStress1000(Solver* solver) { sceneStressN(solver, 10); }

static void sceneRod(Solver* solver)
{
    solver->clear();
    for (int i = 0; i < 15; ++i)
        new Rigid(solver, {0.25f, 1, 0.25f}, (i == 0) ? 0.0f : 1.0f, 0.5f, {0, 10.0f - i * 1.0f, 0});
}

static void sceneSoftBody(Solver* solver)
{
    sceneGround(solver);
    const int W = 10, H = 10;
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < H; ++j)
            new Rigid(solver, {0.5f, 0.5f, 0.5f}, 1.0f, 0.3f, {i * 0.6f - W * 0.3f, j * 0.6f + 2.0f, 0});
}

static void (*scenes[])(Solver*) = {sceneEmpty, sceneGround, sceneStack, scenePyramid, sceneWall,
                                    sceneTwoBlockDrop, sceneStress1000, sceneRod, sceneSoftBody};

static const char* sceneNames[] = {"Empty", "Ground", "Stack", "Pyramid", "Wall",
                                   "TwoBlockDrop", "Stress1000", "Rod (WIP)", "Soft Body (WIP)"};

static const int sceneCount = sizeof(scenes) / sizeof(scenes[0]);

/* BASELINE configs[4]: weld Joints (joint.cpp rows) between x-adjacent cubes — Joint(cube 2k, cube 2k + 1, (0.5, 0, 0), (-0.5, 0, 0)) for the first
 * `count` pairs in creation order (the reference has no jointed scene: build-defined, like StressN itself) */
static void sceneAddWelds(Solver* solver, int count)
{
    int n = 0;
    for (Rigid* r = solver->bodies; r; r = r->next) ++n;
    if (n < 3 || count < 1) return;
    Rigid** byIndex = new Rigid*[n];
    int i = n;
    for (Rigid* r = solver->bodies; r; r = r->next) byIndex[--i] = r;      /* the list is newest first */
    for (int k = 0; k < count && 2 + 2 * k < n; ++k)
        new Joint(solver, byIndex[1 + 2 * k], byIndex[2 + 2 * k], vec3(0.5f, 0.0f, 0.0f), vec3(-0.5f, 0.0f, 0.0f));
    delete[] byIndex;
}

/* name → scene, also accepting "StressN:<n>", "Packed:<n>", "StressGrid:<x>,<y>,<z>", "PackedGrid:...", each optionally followed by "+welds:<count>";
 * returns the table index, 100..102 for the parametrised scenes, -1 when unknown. */
static int buildSceneByName(Solver* solver, const char* name);
static int buildSceneWithSuffix(Solver* solver, const char* name)
{
    const char* plus = strstr(name, "+welds:");
    if (!plus) return buildSceneByName(solver, name);
    char base[96];
    const size_t len = (size_t)(plus - name);
    if (len == 0 || len >= sizeof(base)) return -1;
    memcpy(base, name, len); base[len] = 0;
    const int idx = buildSceneByName(solver, base);
    if (idx < 0) return idx;
    sceneAddWelds(solver, atoi(plus + 7));
    return idx;
}
static int buildSceneByName(Solver* solver, const char* name)
{
    if (!strncmp(name, "StressGrid:", 11) || !strncmp(name, "PackedGrid:", 11)) {
        int nx = 0, ny = 0, nz = 0;
        if (sscanf(name + 11, "%d,%d,%d", &nx, &ny, &nz) != 3 || nx < 1 || ny < 1 || nz < 1) return -1;
        sceneStressGrid(solver, nx, ny, nz, name[0] == 'P');
        return 102;
    }
    if (!strncmp(name, "StressN:", 8)) { sceneStressN(solver, atoi(name + 8), false); return 100; }
    if (!strncmp(name, "Packed:", 7)) { sceneStressN(solver, atoi(name + 7), true); return 101; }
    for (int i = 0; i < sceneCount; ++i)
        if (!strcmp(sceneNames[i], name)) { scenes[i](solver); return i; }
    return -1;
}
