/*
 * avbd_single.hip — k_iterate_single (avbd_iterate.h), the kernel the headline workload spends 92 % of its step in, as a translation
 * unit of its own so that it can be compiled with -O3 (Makefile): 0.585 -> 0.576 ms per launch at Stress1000, same bits (the floating-point
 * flags are the same; -O3 only changes inlining / unrolling / scheduling decisions).  The rest of the device code stays at -O2: the group
 * schedule at 8 000 cubes is 0.8 % slower with -O3, the several-bodies form the same.
 */
#define AVBD_TU_SINGLE 1
#include "avbd_iterate.h"

namespace avbd {
template __global__ void k_iterate_single<false>(BodyTable, ManifoldTable, GForce*, const int* __restrict__, const int* __restrict__, const Counters* __restrict__,
                                                 const int* __restrict__, const int* __restrict__, SolverParams, int*, Counters*, int, int, ResidentExchange);
template __global__ void k_iterate_single<true>(BodyTable, ManifoldTable, GForce*, const int* __restrict__, const int* __restrict__, const Counters* __restrict__,
                                                const int* __restrict__, const int* __restrict__, SolverParams, int*, Counters*, int, int, ResidentExchange);
} // namespace avbd
