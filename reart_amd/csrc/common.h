// reart_amd/csrc/common.h -- shared device helpers for libreart_hip.so (gfx950 only).
//
// Floating-point contract: the whole library is compiled with -ffp-contract=off, so
// every rounding in the source is a rounding in the ISA.  Where a fused multiply-add
// is wanted it is written as fmaf() explicitly.  The CPU oracle (oracle/*.c, test
// infrastructure) is compiled under the same rule, which is what makes bit-exact
// comparisons of indices and most fp32 values possible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/reart_hip.h"

#define REART_WAVE 64

typedef float f2 __attribute__((ext_vector_type(2)));

#include <stdio.h>
#include <stdlib.h>
// Launch-error check; with REART_DEBUG set in the environment the HIP error string and the
// source location go to stderr (the library itself never prints otherwise).
#define REART_CHECK_LAUNCH()                                                          \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            if (getenv("REART_DEBUG"))                                                \
                fprintf(stderr, "[reart] %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return REART_ERR_LAUNCH;                                                  \
        }                                                                             \
    } while (0)

// Dynamic LDS a kernel may request without hipFuncSetAttribute(MaxDynamicSharedMemorySize); launches that need
// more raise the cap at the call (a host-side attribute write, no state kept in the library).
#define REART_LDS_DEFAULT_CAP ((size_t)48 * 1024)

static inline size_t reart_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int reart_div_up(int a, int b) { return (a + b - 1) / b; }

// Squared distance with the library-wide rounding contract:
//   ((dx*dx) + (dy*dy)) + (dz*dz), fp32, no contraction.
__device__ __forceinline__ float reart_sqdist3(float ax, float ay, float az,
                                               float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

// Wave-level sum in a FIXED order (butterfly over lane xor 32,16,...,1) so that results
// do not depend on scheduling.  All 64 lanes get the total.
__device__ __forceinline__ float reart_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double reart_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Minimum / maximum over each group of 16 consecutive lanes (a DPP row), every lane receives the result.
// Four DPP steps (quad xor 1, quad xor 2, half-row mirror, row mirror): one VALU instruction each, where
// __shfl_xor lowers to ds_bpermute_b32 (an LDS crossbar round trip per step).
template <int CTRL>
__device__ __forceinline__ float reart_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float reart_row16_min(float v) {
    v = fminf(v, reart_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fminf(v, reart_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fminf(v, reart_dpp<0x141>(v));   // row_half_mirror
    v = fminf(v, reart_dpp<0x140>(v));   // row_mirror
    return v;
}
__device__ __forceinline__ float reart_row16_max(float v) {
    v = fmaxf(v, reart_dpp<0xB1>(v));
    v = fmaxf(v, reart_dpp<0x4E>(v));
    v = fmaxf(v, reart_dpp<0x141>(v));
    v = fmaxf(v, reart_dpp<0x140>(v));
    return v;
}

// Full-wave butterfly without LDS: the value a lane's partner holds at each of the six steps of an xor-butterfly
// reduction.  Steps 0,1: quad permutes (xor 1, xor 2); steps 2,3: half-row / row mirror, which pair the same GROUPS as
// xor 4 / xor 8 -- equivalent once the groups of 4 / 8 lanes are uniform, i.e. in a reduction with a symmetric combine
// that runs the steps in this order; steps 4,5: gfx950's v_permlane16_swap / v_permlane32_swap (exact xor 16 / xor 32).
// One or two VALU instructions per 32-bit word and step; all 64 lanes must be active.
typedef unsigned reart_v2u __attribute__((ext_vector_type(2)));
template <int STEP>
__device__ __forceinline__ int reart_bfly(int v) {
    if (STEP == 0) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, true);
    if (STEP == 1) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, true);
    if (STEP == 2) return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, true);
    if (STEP == 3) return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, true);
    if (STEP == 4) {
        const reart_v2u r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((threadIdx.x & 16) ? r.x : r.y);
    }
    const reart_v2u r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)((threadIdx.x & 32) ? r.x : r.y);
}
template <int STEP>
__device__ __forceinline__ double reart_bfly_d(double v) {
    return __hiloint2double(reart_bfly<STEP>(__double2hiint(v)), reart_bfly<STEP>(__double2loint(v)));
}
// (smallest value, its lowest index) over the wave, every lane gets it
template <int STEP>
__device__ __forceinline__ void reart_argmin_step(double &v, int &j) {
    const double ov = reart_bfly_d<STEP>(v);
    const int oj = reart_bfly<STEP>(j);
    if (ov < v || (ov == v && oj < j)) { v = ov; j = oj; }
}
__device__ __forceinline__ void reart_wave_argmin_d(double &v, int &j) {
    reart_argmin_step<0>(v, j); reart_argmin_step<1>(v, j); reart_argmin_step<2>(v, j);
    reart_argmin_step<3>(v, j); reart_argmin_step<4>(v, j); reart_argmin_step<5>(v, j);
}

// XCD-aware remap of a linear workgroup id (cdna guide T1): the dispatcher places
// workgroup L on XCD L % 8; give each XCD a contiguous chunk of work items so that
// neighbours (same batch / same target slice) share one L2.  Returns -1 for the
// padding ids of a grid rounded up to a multiple of 8.
__device__ __forceinline__ int reart_xcd_remap(int L, int n_items) {
    const int per = (n_items + 7) >> 3;
    const int w = (L & 7) * per + (L >> 3);
    return (w < n_items && (L >> 3) < per) ? w : -1;
}
static inline int reart_xcd_grid(int n_items) { return ((n_items + 7) / 8) * 8; }
