// Two wavefronts of ONE launch hand a stamped 16-byte record back and forth (the hand-off of the sweep kernels, avbd_resident.h): what does a hop cost
// when both sit on the same XCD and the producer's store KEEPS the line in that XCD's L2 (plain store, sc1 poll), against the write-through form the
// kernels use for every consumer (sc1 store, sc1 poll)?  The XCD of a workgroup is read from HW_REG_XCC_ID, never assumed.
//   hipcc --offload-arch=gfx950 -O2 xcd_pingpong.hip -o xcd_pingpong.bin && ./xcd_pingpong.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }
template <int ST, int LD, int SLEEP>
__global__ void k_pingpong(float4* mem, unsigned bytes, int partner, int rounds, unsigned long long* out, unsigned* xcc)
{
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    const int rank = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == partner ? 1 : -1);
    if (rank < 0 || threadIdx.x != 0) return;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(mem, 0, (int)bytes, 0x00020000);
    const int mine = rank * 256, theirs = (1 - rank) * 256;          // two records on lines of their own
    unsigned long long t0 = 0, spins = 0;
    for (int i = 1; i <= rounds; ++i) {
        if (rank == 0) { v4u v = {(unsigned)i, 1u, 2u, (unsigned)i}; __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, ST); }
        for (;;) {
            v4u g = __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, LD);
            if (g.w == (unsigned)i && g.x == (unsigned)i) break;
            if (++spins > 50000000ull) { out[2] = 1; return; }
            if (SLEEP) __builtin_amdgcn_s_sleep(1);
        }
        if (i == 1) t0 = __builtin_amdgcn_s_memrealtime();
        if (rank == 1) { v4u v = {(unsigned)i, 1u, 2u, (unsigned)i}; __builtin_amdgcn_raw_buffer_store_b128(v, r, mine, 0, ST); }
    }
    if (rank == 0) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = spins; out[2] = 0; }
}
template <int ST, int LD, int SLEEP>
static void run(const char* what, float4* mem, int partner, unsigned long long* out, unsigned* xcc)
{
    const int rounds = 4000;
    CHECK(hipMemset(mem, 0, 4096)); out[2] = 7; CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k_pingpong<ST, LD, SLEEP>), dim3(32), dim3(64), 0, 0, mem, 4096u, partner, rounds, out, xcc);
    CHECK(hipDeviceSynchronize());
    printf("block 0 (XCC %u) <-> block %2d (XCC %u), %-34s %s %.3f us per hop, %llu polls\n", xcc[0], partner, xcc[partner], what,
           out[2] == 0 ? "ok" : "TIMED OUT (never visible)", out[0] * 0.01 / (rounds - 1) / 2.0, out[1]);
}
int main()
{
    CHECK(hipSetDevice(0));
    float4* mem; CHECK(hipMalloc((void**)&mem, 4096));
    unsigned long long* out; CHECK(hipHostMalloc((void**)&out, 64));
    unsigned* xcc; CHECK(hipHostMalloc((void**)&xcc, 32 * sizeof(unsigned)));
    for (int partner : {8, 16, 1, 3}) {
        run<16, 16, 1>("sc1 store, sc1 poll, s_sleep 1:", mem, partner, out, xcc);
        run<0, 16, 1>("plain store, sc1 poll, s_sleep 1:", mem, partner, out, xcc);
        run<16, 16, 0>("sc1 store, sc1 poll, no sleep:", mem, partner, out, xcc);
        run<0, 16, 0>("plain store, sc1 poll, no sleep:", mem, partner, out, xcc);
        run<1, 16, 0>("sc0 store, sc1 poll, no sleep:", mem, partner, out, xcc);
    }
    return 0;
}
