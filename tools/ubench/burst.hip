// Micro-benchmark (not product): latency of the memory phases of one pass of k_iterate_groups<true,*> in isolation, with every
// SIMD running two such waves: a burst of NL scattered 16-byte loads (one line per lane, planes far apart) until the last has
// returned, and a burst of NS scattered stores (plain / sc1 mix) until all are acknowledged.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/burst.hip -o tools/ubench/burst.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }

template <int NL, int NS, int NSC1>
__global__ void k(v4u* buf, unsigned planeBytes, unsigned long long* out, int iters, int activeLanes, int sleep)
{
    const __amdgpu_buffer_rsrc_t r = rsrc(buf, planeBytes * 32u);
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned long long tl = 0, ts = 0;
    unsigned acc = 0;
    unsigned seed = w * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        seed = seed * 1664525u + 1013904223u;
        // a "body": its entries are 4-contact manifolds: lanes 4m..4m+3 share a line
        const unsigned slot = ((seed >> 8) + (unsigned)(lane >> 2) * 977u) % (planeBytes / 64u);
        const unsigned off = slot * 64u + (lane & 3) * 16u;
        v4u q[NL > 0 ? NL : 1];
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (lane < activeLanes) {
#pragma unroll
            for (int k2 = 0; k2 < NL; ++k2) q[k2] = __builtin_amdgcn_raw_buffer_load_b128(r, off + (unsigned)k2 * planeBytes, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane < activeLanes) {
#pragma unroll
            for (int k2 = 0; k2 < NL; ++k2) acc += q[k2].x;
        }
        v4u v = {acc, (unsigned)i, 1u, 2u};
        if (lane < activeLanes) {
#pragma unroll
            for (int k2 = 0; k2 < NS; ++k2) {
                if (k2 < NSC1) __builtin_amdgcn_raw_buffer_store_b128(v, r, off + (unsigned)(k2 + 16) * planeBytes, 0, 16);
                else __builtin_amdgcn_raw_buffer_store_b128(v, r, off + (unsigned)(k2 + 16) * planeBytes, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        tl += t1 - t0; ts += t2 - t1;
        for (int s = 0; s < sleep; ++s) __builtin_amdgcn_s_sleep(32);
    }
    if (lane == 0) { out[2 * w] = tl + (acc & 1); out[2 * w + 1] = ts; }
}

template <int NL, int NS, int NSC1>
void run(v4u* buf, unsigned planeBytes, unsigned long long* dout, int waves, int activeLanes, int sleep)
{
    const int iters = 300, blocks = (waves + 3) / 4;
    hipLaunchKernelGGL((k<NL, NS, NSC1>), dim3(blocks), dim3(256), 0, 0, buf, planeBytes, dout, iters, activeLanes, sleep);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * blocks * 4);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double l = 0, s = 0; for (int i = 0; i < blocks * 4; ++i) { l += (double)h[2 * i]; s += (double)h[2 * i + 1]; }
    const double n = (double)blocks * 4 * iters * 2400.0;
    printf("waves %5d lanes %2d sleep %3d | %2d loads: %6.2f us | %2d stores (%d sc1): %6.2f us\n", waves, activeLanes, sleep, NL, l / n, NS, NSC1, s / n);
}

int main()
{
    const unsigned planeBytes = 24u << 20;      // 32 planes x 24 MB = 768 MB: nothing stays in L2, little in the Infinity Cache
    v4u* buf; unsigned long long* dout;
    hipMalloc(&buf, (size_t)planeBytes * 32); hipMalloc(&dout, 1 << 20);
    hipMemset(buf, 0, (size_t)planeBytes * 32);
    for (int waves : {4, 2048})
        for (int lanes : {13, 64})
            for (int sleep : {0, 8}) {
                run<1, 1, 0>(buf, planeBytes, dout, waves, lanes, sleep);
                run<1, 1, 1>(buf, planeBytes, dout, waves, lanes, sleep);
                run<8, 4, 0>(buf, planeBytes, dout, waves, lanes, sleep);
                run<16, 12, 0>(buf, planeBytes, dout, waves, lanes, sleep);
                run<16, 12, 5>(buf, planeBytes, dout, waves, lanes, sleep);
            }
    return 0;
}
