// Micro-benchmark (not product): how long until a wave's store is acknowledged (s_waitcnt vmcnt(0) returns), by cache policy,
// on an idle device and with every SIMD running two such waves; and the latency of an sc1 load issued behind such a store.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/store_ack.hip -o gpurun_out/store_ack && gpurun_out/store_ack
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }

template <int AUX, int MODE>   // MODE 0: store + drain; 1: store, then load (aux 16) + drain; 2: load only
__global__ void k(v4u* buf, unsigned bytes, unsigned long long* out, int iters, int stride)
{
    const __amdgpu_buffer_rsrc_t r = rsrc(buf, bytes);
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned long long sum = 0;
    unsigned acc = 0;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), m0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        const unsigned off = (unsigned)(((size_t)(w * 131 + i * 17) * stride + lane) * 16u) % (bytes - 1024u) & ~15u;
        v4u v = {(unsigned)i, (unsigned)w, acc, 7u};
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE != 2) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
        if (MODE != 0) { v4u q = __builtin_amdgcn_raw_buffer_load_b128(r, (off + 4096u) % (bytes - 1024u) & ~15u, 0, 16); acc += q.x; }
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        sum += t1 - t0;
        __builtin_amdgcn_s_sleep(20);
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), m1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[w] = sum + (acc & 1); if (w == 0) { out[8192] = r1 - r0; out[8193] = m1 - m0; } }
}

template <int AUX, int MODE>
void run(const char* name, v4u* buf, unsigned bytes, unsigned long long* dout, int waves, int stride)
{
    const int iters = 200, blocks = (waves + 3) / 4;
    hipLaunchKernelGGL((k<AUX, MODE>), dim3(blocks), dim3(waves < 4 ? 64 * waves : 256), 0, 0, buf, bytes, dout, iters, stride);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(8200);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < waves; ++i) s += (double)h[i];
    const double ticksPerUs = (double)h[8193] / ((double)h[8192] / 100.0);    /* s_memrealtime is 100 MHz */
    printf("%-44s waves %5d  mean %8.1f ticks of s_memtime = %.3f us  (s_memtime runs at %.1f MHz)\n", name, waves, s / waves / iters, s / waves / iters / ticksPerUs, ticksPerUs);
}

int main()
{
    const unsigned bytes = 512u << 20;
    v4u* buf; unsigned long long* dout;
    hipMalloc(&buf, bytes); hipMalloc(&dout, 1 << 20);
    hipMemset(buf, 0, bytes);
    for (int waves : {1, 256, 2048}) {
        run<0, 0>("plain store", buf, bytes, dout, waves, 64);
        run<16, 0>("sc1 store", buf, bytes, dout, waves, 64);
        run<17, 0>("sc0 sc1 store", buf, bytes, dout, waves, 64);
        run<2, 0>("nt store", buf, bytes, dout, waves, 64);
        run<18, 0>("sc1 nt store", buf, bytes, dout, waves, 64);
        run<0, 2>("sc1 load alone", buf, bytes, dout, waves, 64);
        run<0, 1>("plain store + sc1 load", buf, bytes, dout, waves, 64);
        run<16, 1>("sc1 store + sc1 load", buf, bytes, dout, waves, 64);
    }
    return 0;
}
