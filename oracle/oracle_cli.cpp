/*
 * oracle_cli.cpp — headless driver for the CPU restatement (TEST INFRASTRUCTURE).
 * Same flags and stdout wire format as the reference's main.cpp:189-248 so that
 *   oracle_cli --nogfx --scene Stack --steps 300 | md5sum
 * can be compared with oracle/_ref/avbd_ref and with the md5s in SURVEY.md §8c.
 * Extra build-only flags: --quiet (timing JSON instead of the trace), --dump <file>.
 */
#include "avbd_oracle.h"
#include "harness_api.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
void* orc_create(); void orc_destroy(void*); int orc_scene(void*, const char*); void orc_step(void*, int);
void orc_print_step(void*, int); void orc_set_diag_logging(void*, int, int); int orc_num_bodies(void*);
void orc_get_bodies(void*, HarnessBody*);
}

int main(int argc, char** argv)
{
    const char* requested = nullptr; const char* dumpPath = nullptr;
    int steps = 300; bool quiet = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--nogfx") || !strcmp(argv[i], "--headless")) {}
        else if ((!strcmp(argv[i], "--scene") || !strcmp(argv[i], "-s")) && i + 1 < argc) requested = argv[++i];
        else if ((!strcmp(argv[i], "--steps") || !strcmp(argv[i], "-n")) && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--quiet")) quiet = true;
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dumpPath = argv[++i];
    }
    void* w = orc_create();
    orc_set_diag_logging(w, quiet ? 0 : 1, 1);
    const char* shown = orc::kSceneNames[0];
    int idx = requested ? orc_scene(w, requested) : -1;
    if (idx < 0) orc_scene(w, orc::kSceneNames[0]);
    else if (idx < 9) shown = orc::kSceneNames[idx];
    else shown = requested;
    if (!quiet) printf("Running in headless mode: scene '%s', steps=%d\n", shown, steps);
    auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s) { orc_step(w, 1); if (!quiet) orc_print_step(w, s); }
    auto t1 = std::chrono::steady_clock::now();
    if (quiet) {
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        printf("{\"scene\": \"%s\", \"steps\": %d, \"ms_total\": %.3f, \"ms_per_step\": %.6f}\n", shown, steps, ms, steps ? ms / steps : 0.0);
    }
    if (dumpPath) {
        int n = orc_num_bodies(w);
        std::vector<HarnessBody> hb(n);
        orc_get_bodies(w, hb.data());
        if (FILE* f = fopen(dumpPath, "wb")) { fwrite(hb.data(), sizeof(HarnessBody), n, f); fclose(f); }
    }
    orc_destroy(w);
    return 0;
}
