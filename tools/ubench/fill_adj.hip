// What bounds k_fill_adj (avbd_kernels.hip)?  The kernel on a synthetic 40 x 40 x 40 pile: every cube touches its +x / +y / +z neighbour and the bottom layer
// the ground (body 0, not swept) — 187 000 manifolds in random slot order —, with the parts of the kernel switched off one at a time.
//   hipcc --offload-arch=gfx950 -O2 fill_adj.hip -o fill_adj.bin && ./fill_adj.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
template <int MODE>   // 0 full, 1 no stores, 2 no atomics (position = slot-derived), 3 atomics without the wave-aggregated path, 4 one store instead of three
__device__ __forceinline__ int adj_position(int body, bool want, const int* __restrict__ solved, const int* __restrict__ adjStart, int* cursor, int item)
{
    const int lane = threadIdx.x & 63;
    const bool shared = MODE != 3 && want && !solved[body];
    int pos = -1;
    if (MODE == 2) { if (want) pos = adjStart[body] + (item & 3); return pos; }
    if (want && !shared) pos = adjStart[body] + atomicAdd(&cursor[body], 1);
    unsigned long long m = __ballot(shared);
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        const int v = __shfl(body, l, 64);
        const unsigned long long same = __ballot(shared && body == v);
        int base = 0;
        if (lane == l) base = atomicAdd(&cursor[v], (int)__popcll(same));
        base = __shfl(base, l, 64);
        if (shared && body == v) pos = adjStart[v] + base + (int)__popcll(same & ((1ull << lane) - 1ull));
        m &= ~same;
    }
    return pos;
}
template <int MODE>
__global__ void k_fill(const int2* __restrict__ pair, const unsigned long long* __restrict__ keys, int nm, const int* adjStart, int* cursor, int* adjRef, int* adjOther,
                       unsigned long long* adjKey, uint4* adjRec, const int* __restrict__ solved, int* sink)
{
    const int stride = gridDim.x * blockDim.x;
    int acc = 0;
    for (int base = blockIdx.x * blockDim.x; base < nm; base += stride) {
        const int item = base + (int)threadIdx.x;
        int a = -1, b = -1; unsigned long long key = 0;
        if (item < nm) { const int2 p = pair[item]; a = p.x; b = p.y; key = keys[item]; }
        const int pa = adj_position<MODE>(a, a >= 0, solved, adjStart, cursor, item);
        const int pb = adj_position<MODE>(b, b >= 0, solved, adjStart, cursor, item);
        if (MODE == 1) { acc += pa + pb; continue; }
        if (MODE == 4) {
            if (a >= 0) adjRec[pa] = make_uint4((unsigned)key, (unsigned)(key >> 32), (unsigned)item, (unsigned)b);
            if (b >= 0) adjRec[pb] = make_uint4((unsigned)key, (unsigned)(key >> 32), (unsigned)item, (unsigned)a);
        } else {
            if (a >= 0) { adjRef[pa] = item; adjOther[pa] = b; adjKey[pa] = key; }
            if (b >= 0) { adjRef[pb] = item; adjOther[pb] = a; adjKey[pb] = key; }
        }
    }
    if (MODE == 1 && acc == 0x7fffffff) *sink = acc;
}
int main()
{
    const int N = 40, nb = N * N * N + 1;
    std::vector<int2> pairs; std::vector<int> deg(nb, 0), solved(nb, 1); solved[0] = 0;
    auto id = [&](int x, int y, int z) { return 1 + (z * N + y) * N + x; };
    for (int z = 0; z < N; ++z) for (int y = 0; y < N; ++y) for (int x = 0; x < N; ++x) {
        const int i = id(x, y, z);
        if (x + 1 < N) pairs.push_back(make_int2(id(x + 1, y, z), i));
        if (y + 1 < N) pairs.push_back(make_int2(id(x, y + 1, z), i));
        if (z + 1 < N) pairs.push_back(make_int2(id(x, y, z + 1), i));
        if (y == 0) pairs.push_back(make_int2(i, 0));
    }
    std::mt19937 rng(7); std::shuffle(pairs.begin(), pairs.end(), rng);
    const int nm = (int)pairs.size();
    for (auto& p : pairs) { ++deg[p.x]; ++deg[p.y]; }
    std::vector<int> start(nb + 1, 0); for (int i = 0; i < nb; ++i) start[i + 1] = start[i] + deg[i];
    std::vector<unsigned long long> keys(nm); for (int i = 0; i < nm; ++i) keys[i] = ((unsigned long long)i << 20) | 5u;
    int2* dPair; unsigned long long *dKeys, *dAdjKey; int *dStart, *dCursor, *dRef, *dOther, *dSolved, *dSink; uint4* dRec;
    CHECK(hipMalloc(&dPair, nm * sizeof(int2))); CHECK(hipMalloc(&dKeys, nm * 8)); CHECK(hipMalloc(&dAdjKey, 2 * nm * 8)); CHECK(hipMalloc(&dStart, (nb + 1) * 4));
    CHECK(hipMalloc(&dCursor, nb * 4)); CHECK(hipMalloc(&dRef, 2 * nm * 4)); CHECK(hipMalloc(&dOther, 2 * nm * 4)); CHECK(hipMalloc(&dSolved, nb * 4)); CHECK(hipMalloc(&dSink, 4));
    CHECK(hipMalloc(&dRec, 2 * nm * 16));
    CHECK(hipMemcpy(dPair, pairs.data(), nm * sizeof(int2), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dKeys, keys.data(), nm * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dStart, start.data(), (nb + 1) * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dSolved, solved.data(), nb * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = (nm + 255) / 256;
    auto run = [&](const char* what, auto kern, int grid) {
        float best = 1e9f;
        for (int rep = 0; rep < 7; ++rep) {
            CHECK(hipMemset(dCursor, 0, nb * 4)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dPair, dKeys, nm, dStart, dCursor, dRef, dOther, dAdjKey, dRec, dSolved, dSink);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("%-64s %7.1f us  (%d manifolds, %d workgroups)\n", what, best * 1e3f, nm, grid);
    };
    run("as in the library:", k_fill<0>, blocks);
    run("no stores:", k_fill<1>, blocks);
    run("no atomics (positions from the slot):", k_fill<2>, blocks);
    run("ground entries through the per-lane atomic like the others:", k_fill<3>, blocks);
    run("one 16-byte record per entry instead of three stores:", k_fill<4>, blocks);
    run("as in the library, 256 workgroups (grid-stride):", k_fill<0>, 256);
    run("as in the library, 2048 workgroups:", k_fill<0>, 2048);
    run("as in the library, 128 workgroups:", k_fill<0>, 128);
    run("as in the library, 512 workgroups:", k_fill<0>, 512);
    run("one record per entry, 256 workgroups:", k_fill<4>, 256);
    run("no atomics, 256 workgroups:", k_fill<2>, 256);
    return 0;
}
