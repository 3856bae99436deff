/* avbd_util.h — small device helpers shared by the translation units (avbd_capi.hip via avbd_kernels.hip, avbd_single.hip). */
#pragma once
#include "avbd_types.h"
#include "avbd_rows.h"

namespace avbd {

__device__ __forceinline__ f3 ld3(const float4* p, int i) { float4 v = p[i]; return mk3(v.x, v.y, v.z); }
__device__ __forceinline__ q4 ldq(const float4* p, int i) { float4 v = p[i]; return mkq(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 pack3(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ __forceinline__ float4 packq(q4 q) { return make_float4(q.x, q.y, q.z, q.w); }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* max over non-negative floats through their uint bit patterns; NaNs are skipped; -0.0f (what max(0.0f, -separation) yields
 * for a separation of exactly +0) counts as +0.0f — as a uint it would beat every positive float */
__device__ __forceinline__ void atomic_max_nonneg(unsigned* dst, float v)
{
    if (v == v) atomicMax(dst, __float_as_uint(v + 0.0f));
}

/* what the reference prints when it resets a non-finite quantity (solver.cpp:51-66) goes into the log behind the Counters word */
__device__ __noinline__ inline void warn_log(Counters* cnt, unsigned key, int body, int kind, float x, float y, float z)
{
    const int i = atomicAdd(&cnt->warnings, 1);
    if (i < kWarnCap) {
        WarnRec r; r.key = key; r.body = body; r.x = x; r.y = y; r.z = z; r.kind = kind; r.pad0 = 0; r.pad1 = 0;
        reinterpret_cast<WarnRec*>(cnt + 1)[i] = r;
    }
}
/* solver.cpp:51-66: reset to zero / identity; the warning text is printed by the host from the log */
__device__ __forceinline__ void sanitize3(f3& v, Counters* cnt, unsigned key, int body, int kind) { if (!finite3(v)) { warn_log(cnt, key, body, kind, v.x, v.y, v.z); v = zero3(); } }
__device__ __forceinline__ void sanitizeq(q4& q, Counters* cnt, unsigned key, int body, int kind) { if (!finiteq(q)) { warn_log(cnt, key, body, kind, q.x, q.y, q.z); q = identq(); } }

} // namespace avbd
