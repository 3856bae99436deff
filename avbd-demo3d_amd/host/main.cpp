/*
 * main.cpp — headless driver of the MI355X build: same flags and stdout wire format as the headless
 * branch of the reference's main.cpp:189-248 (the SDL/OpenGL/ImGui front-end is out of scope).
 *
 *   avbd_demo3d --nogfx --scene Stack --steps 300        # trace identical to the reference CPU run
 *
 * Build-only flags (do not exist in the reference): --quiet (no per-step trace; one timing JSON line),
 * --dump <file> (raw AvbdBody array after the last step), --device <n>, --batch (run all steps on the
 * device without per-step downloads; implies --quiet), --save-state <file> / --load-state <file>
 * (checkpoint after the last step / resume from a checkpoint instead of building --scene: SURVEY.md §8 f3),
 * --remove-body <creation index> --remove-at <step> (`delete body` before that step, through the host mirror),
 * --pick ox,oy,oz,dx,dy,dz (after the last step: Solver::pick along that ray; prints one `Pick:` line with the body id,
 * creation index and body-local hit point, floats as hex so the line can be compared bit for bit),
 * --gpus N [--cuts exact|gather] (the scene sharded over N ranks of this process: N contexts on the visible devices, round robin
 * — on one GPU they share it —, slabs along x, Solver::stepGroup; `gather` = one all-gather per iteration, Jacobi across the cuts
 * (default, BASELINE configs[3..4]), `exact` = bit-identical to --gpus 1; implies --batch; state and --dump come from rank 0),
 * --mode exact|fast (`exact`, the default, replays the reference's sweep bit for bit; `fast` is the throughput sweep of
 * avbd_fast.hip — multi-colour Gauss-Seidel, tree-reduced sums — within the tolerance of DESIGN.md section 2b = avbd_set_option(ctx, "sweep", 1)),
 * --option key=value (any avbd_set_option key; repeatable).
 */
#include "solver.h"
#include "joint.h"
#include "spring.h"
#include "ignorecollision.h"
#include "scenes.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

int main(int argc, char** argv)
{
    const char* requested = nullptr;
    const char* dumpPath = nullptr;
    const char* savePath = nullptr;
    const char* loadPath = nullptr;
    const char* pickRay = nullptr;
    int removeBody = -1, removeAt = -1;   /* --remove-body <creation index> --remove-at <step>: `delete body` mid-run */
    int steps = 300; /* main.cpp:192 */
    bool quiet = false, batch = false;
    int gpus = 1; bool exactCuts = false; int rebalanceEvery = 0;
    bool fastMode = false;
    std::vector<std::pair<std::string, int>> options;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--nogfx") || !strcmp(argv[i], "--headless")) { /* this build is always headless */ }
        else if ((!strcmp(argv[i], "--scene") || !strcmp(argv[i], "-s")) && i + 1 < argc) requested = argv[++i];
        else if ((!strcmp(argv[i], "--steps") || !strcmp(argv[i], "-n")) && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--quiet")) quiet = true;
        else if (!strcmp(argv[i], "--batch")) { batch = true; quiet = true; }
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dumpPath = argv[++i];
        else if (!strcmp(argv[i], "--save-state") && i + 1 < argc) savePath = argv[++i];
        else if (!strcmp(argv[i], "--load-state") && i + 1 < argc) loadPath = argv[++i];
        else if (!strcmp(argv[i], "--remove-body") && i + 1 < argc) removeBody = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--remove-at") && i + 1 < argc) removeAt = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--pick") && i + 1 < argc) pickRay = argv[++i];
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) setenv("AVBD_DEVICE", argv[++i], 1);
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) { gpus = atoi(argv[++i]); if (gpus < 1) gpus = 1; }
        else if (!strcmp(argv[i], "--cuts") && i + 1 < argc) exactCuts = !strcmp(argv[++i], "exact");
        else if (!strcmp(argv[i], "--rebalance") && i + 1 < argc) rebalanceEvery = atoi(argv[++i]);   /* --gpus N: new slabs from the current positions every K steps */
        else if (!strcmp(argv[i], "--mode") && i + 1 < argc) fastMode = !strcmp(argv[++i], "fast");
        else if (!strcmp(argv[i], "--option") && i + 1 < argc) {
            const char* kv = argv[++i];
            const char* eq = strchr(kv, '=');
            if (eq) options.emplace_back(std::string(kv, eq), atoi(eq + 1));
        }
    }

    if (gpus > 1) { batch = true; quiet = true; }
    Solver* solver = new Solver();
    if (!quiet) { solver->enableDiagnostics = true; solver->logFrequency = 1; } /* main.cpp:206-209 */
    std::vector<Solver*> ranks(1, solver);
    /* one context per rank.  All-gather form: devices round robin from --device on.  Exact cuts between ranks of ONE process need the
     * records of all ranks in one allocation, so every rank stays on rank 0's device (a step that cannot run exact cuts is an error,
     * never a silent Jacobi coupling: avbd_set_shard_mode 1). */
    const int dev0 = getenv("AVBD_DEVICE") ? atoi(getenv("AVBD_DEVICE")) : 0;
    for (int r = 1; r < gpus; ++r) {
        const int nd = avbd_device_count();
        char dev[16]; snprintf(dev, sizeof(dev), "%d", exactCuts ? dev0 : (nd > 0 ? (dev0 + r) % nd : 0));
        setenv("AVBD_DEVICE", dev, 1);
        ranks.push_back(new Solver());
    }

    /* unknown names silently select scene 0 "Empty" (main.cpp:211-219) */
    const char* shown = sceneNames[0];
    for (Solver* s : ranks) {
        if (loadPath) {
            s->loadSnapshot(loadPath);
            shown = loadPath;
        } else {
            int idx = requested ? buildSceneWithSuffix(s, requested) : -1;
            if (idx < 0) scenes[0](s);
            else if (idx < sceneCount) shown = sceneNames[idx];
            else shown = requested;
        }
    }
    for (Solver* s : ranks) {
        if (fastMode) avbd_set_option(s->context(), "sweep", 1);
        for (const auto& kv : options)
            if (avbd_set_option(s->context(), kv.first.c_str(), kv.second) != AVBD_OK) fprintf(stderr, "avbd_demo3d: --option %s: %s\n", kv.first.c_str(), avbd_last_error(s->context()));
    }
    if (gpus > 1) Solver::shardAcross(ranks.data(), gpus, 0, exactCuts);

    if (!quiet) printf("Running in headless mode: scene '%s', steps=%d\n", shown, steps);
    auto t0 = std::chrono::steady_clock::now();
    if (gpus > 1) {
        for (Solver* s : ranks) s->hostSync = false;
        if (rebalanceEvery > 0) {
            for (int done = 0; done < steps; done += rebalanceEvery) {
                if (done) Solver::rebalance(ranks.data(), gpus, 0);
                Solver::stepGroup(ranks.data(), gpus, steps - done < rebalanceEvery ? steps - done : rebalanceEvery);
            }
        } else Solver::stepGroup(ranks.data(), gpus, steps);
        solver->pull();
    } else if (batch) {
        solver->hostSync = false;
        solver->stepMany(steps);
        solver->pull();
    } else {
        for (int step = 0; step < steps; ++step) {
            if (step == removeAt && removeBody >= 0) delete solver->bodyByIndex(removeBody);
            solver->step();
            if (quiet) continue;
            printf("Step %d:\n", step);
            for (Rigid* body = solver->bodies; body != nullptr; body = body->next) {
                printf("  Body %d: Pos(%.4f, %.4f, %.4f)  ", body->id, body->position.x, body->position.y, body->position.z);
                printf("Rot(%.4f, %.4f, %.4f, %.4f)  ", body->orientation.x, body->orientation.y, body->orientation.z, body->orientation.w);
                printf("LinVel(%.4f, %.4f, %.4f)  ", body->linearVelocity.x, body->linearVelocity.y, body->linearVelocity.z);
                printf("AngVel(%.4f, %.4f, %.4f)\n", body->angularVelocity.x, body->angularVelocity.y, body->angularVelocity.z);
            }
            const Solver::Diagnostics& d = solver->lastDiagnostics;
            printf("  Diagnostics: manifolds=%d contacts=%d dynBodies=%d maxPen=%.6f maxDrift=%.6f maxLin=%.3f maxAng=%.3f maxLambda=%.3f\n",
                   d.activeManifolds, d.activeContacts, d.dynamicBodies, d.maxPenetration, d.maxConstraintViolation, d.maxLinearSpeed,
                   d.maxAngularSpeed, d.maxNormalImpulse);
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (quiet) {
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        /* what the last step RAN (AvbdStepStats.cutMode), not what was asked for */
        static const char* const cutNames[4] = {"none", "all-gather per iteration (one launch per iteration)", "all-gather per iteration (persistent launch)", "exact"};
        AvbdStepStats st{};
        if (steps > 0) avbd_get_step_stats(solver->context(), &st);
        printf("{\"scene\": \"%s\", \"steps\": %d, \"ms_total\": %.3f, \"ms_per_step\": %.6f, \"backend\": \"hip-gfx950\", \"ranks\": %d, \"cuts\": \"%s\", \"mode\": \"%s\", \"schedule\": %d}\n", shown, steps, ms,
               steps ? ms / steps : 0.0, gpus, cutNames[st.cutMode >= 0 && st.cutMode < 4 ? st.cutMode : 0], fastMode ? "fast" : "exact", st.schedule);
    }
    if (pickRay) {
        float r[6] = {0, 0, 0, 0, 0, 0};
        sscanf(pickRay, "%f,%f,%f,%f,%f,%f", &r[0], &r[1], &r[2], &r[3], &r[4], &r[5]);
        vec3 local(0.0f, 0.0f, 0.0f);
        Rigid* hit = solver->pick(vec3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5]), local);
        if (hit) printf("Pick: body id=%d index=%d local=(%a, %a, %a)\n", hit->id, hit->index, local.x, local.y, local.z);
        else printf("Pick: none\n");
    }
    if (savePath) solver->saveSnapshot(savePath);
    if (dumpPath) {
        int n = solver->bodyCount();
        std::vector<AvbdBody> hb((size_t)n);
        if (n) avbd_download_bodies(solver->context(), hb.data(), n);
        if (FILE* f = fopen(dumpPath, "wb")) { fwrite(hb.data(), sizeof(AvbdBody), (size_t)n, f); fclose(f); }
    }
    for (size_t r = ranks.size(); r-- > 0;) delete ranks[r];
    return 0;
}
