// Two PROCESSES on one MI355X, one persistent kernel each, hand a stamped 16-byte record back and forth through memory that one of them
// allocated and the other opened with hipIpcOpenMemHandle: (1) does IPC work here, (2) do kernels of two processes run side by side,
// (3) what does a hop cost with device-scope (sc1) and system-scope (sc0 sc1) accesses, plain and fine-grained memory.
//   hipcc --offload-arch=gfx950 -O2 ipc_pingpong.hip -o ipc_pingpong.bin
//   ./ipc_pingpong.bin 0 /tmp/ipc_h.bin [fine] &  ./ipc_pingpong.bin 1 /tmp/ipc_h.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", g_rank, #x, hipGetErrorString(e_)); exit(2); } } while (0)
static int g_rank = 0;
template <int AUX>
__global__ void k_pingpong(float4* mem, unsigned bytes, int rank, int rounds, unsigned long long* out)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(mem, 0, (int)bytes, 0x00020000);
    const int mine = rank * 64, theirs = (1 - rank) * 64;          // two records, a cache line apart
    unsigned long long t0 = 0, spins = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (rank == 0) { v4u v = {(unsigned)i, 1u, 2u, (unsigned)i}; __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, AUX); }
        for (;;) {
            v4u g = __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, AUX);
            if (g.w == (unsigned)i && g.x == (unsigned)i) break;
            if (++spins > 400000000ull) { out[2] = 1; return; }      // the other kernel never ran beside this one
            __builtin_amdgcn_s_sleep(1);
        }
        if (i == 1) t0 = __builtin_amdgcn_s_memrealtime();
        if (rank == 1) { v4u v = {(unsigned)i, 1u, 2u, (unsigned)i}; __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, AUX); }
    }
    out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = spins; out[2] = 0;
}
int main(int argc, char** argv)
{
    g_rank = atoi(argv[1]); const char* path = argv[2]; const bool fine = argc > 3 && !strcmp(argv[3], "fine");
    CHECK(hipSetDevice(0));
    float4* mem = nullptr; const unsigned bytes = 4096;
    char donePath[512]; snprintf(donePath, sizeof(donePath), "%s.done", path);
    if (g_rank == 0) {
        if (fine) CHECK(hipExtMallocWithFlags((void**)&mem, bytes, hipDeviceMallocUncached)); else CHECK(hipMalloc((void**)&mem, bytes));
        CHECK(hipMemset(mem, 0, bytes)); CHECK(hipDeviceSynchronize());
        hipIpcMemHandle_t h; CHECK(hipIpcGetMemHandle(&h, mem));
        char tmp[512]; snprintf(tmp, sizeof(tmp), "%s.tmp", path);
        FILE* f = fopen(tmp, "wb"); fwrite(&h, sizeof(h), 1, f); fclose(f); rename(tmp, path);
    } else {
        hipIpcMemHandle_t h; FILE* f = nullptr;
        for (int k = 0; k < 3000 && !(f = fopen(path, "rb")); ++k) usleep(10000);
        if (!f) { fprintf(stderr, "rank 1: no handle file\n"); return 3; }
        if (fread(&h, sizeof(h), 1, f) != 1) return 3; fclose(f);
        CHECK(hipIpcOpenMemHandle((void**)&mem, h, hipIpcMemLazyEnablePeerAccess));
    }
    unsigned long long* out; CHECK(hipHostMalloc((void**)&out, 64)); memset(out, 0, 64);
    const int rounds = 2000;
    for (int variant = 0; variant < 2; ++variant) {
        // (the two processes start a variant together only roughly: the first hop absorbs the skew, timing starts behind it)
        if (variant == 0) hipLaunchKernelGGL(k_pingpong<16>, dim3(1), dim3(1), 0, 0, mem, bytes, g_rank, rounds, out);
        else hipLaunchKernelGGL(k_pingpong<17>, dim3(1), dim3(1), 0, 0, mem + 32, bytes - 512, g_rank, rounds, out);
        CHECK(hipDeviceSynchronize());
        printf("rank %d %s memory, %s: %s, %.3f us per round trip (2 hops), %llu polls\n", g_rank, fine ? "uncached" : "plain", variant ? "sc0 sc1" : "sc1",
               out[2] ? "TIMED OUT (kernels of the two processes did not run side by side)" : "ok", out[0] * 0.01 / (rounds - 1), out[1]);
    }
    if (g_rank == 1) { FILE* f = fopen(donePath, "wb"); fclose(f); CHECK(hipIpcCloseMemHandle(mem)); }
    else { for (int k = 0; k < 3000 && access(donePath, F_OK) != 0; ++k) usleep(10000); CHECK(hipFree(mem)); unlink(path); unlink(donePath); }
    return 0;
}
