// reart_amd/csrc/blocksort.h -- workgroup-level stable LSD radix sort of ids by an integer key.
//
// Used by the K-NN backward kernels to turn the scatter-add "grad_p2[idx[i]] -= ..." of the
// reference (utils/chamfer.py:206-208 -> knn_points_backward) into a per-target gather whose
// summation order is the ascending source order -- deterministic, no floating-point atomics,
// and O(M) per pass even when every source hits the same target (an insertion sort of the
// buckets is O(M^2) on such degenerate inputs).
//
// Stability comes from ownership, not atomics: thread t owns the contiguous slice
// [t*chunk, (t+1)*chunk) of the current order and private digit counters s_cnt[d][t]; an
// exclusive scan over the digit-major flattening gives every (digit, thread) its output
// offset, so equal keys keep their relative order.
#pragma once
#include "common.h"

#define RS_BS 1024     // threads of the sorting workgroup
#define RS_BITS 3      // radix bits per pass: 8 digits x 1024 threads x 4 B = 32 KiB of LDS
#define RS_DIG (1 << RS_BITS)

// exclusive scan of one int per thread across the workgroup; s_wave: [RS_BS/64] ints
__device__ __forceinline__ int block_excl_scan(int v, int *s_wave, int *total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        int w = (lane < RS_BS / 64) ? s_wave[lane] : 0;
        int winc = w;
#pragma unroll
        for (int o = 1; o < RS_BS / 64; o <<= 1) {
            const int u = __shfl_up(winc, o, 64);
            if (lane >= o) winc += u;
        }
        if (lane < RS_BS / 64) s_wave[lane] = winc - w;  // exclusive prefix of the wave totals
        if (lane == RS_BS / 64 - 1 && total) *total = winc;
    }
    __syncthreads();
    const int res = s_wave[wv] + inc - v;
    __syncthreads();  // s_wave may be reused by the caller's next scan
    return res;
}

// Sort the ids 0..M-1 by key(id) in [0, 2^nbits).  bufA / bufB: global int scratch [M].
// s_cnt: LDS [RS_DIG * RS_BS] ints, s_wave: LDS [RS_BS/64] ints.  Returns the buffer that
// holds the sorted ids (all threads get the same pointer).  Must be called by all RS_BS
// threads of the workgroup.
template <typename KeyFn>
__device__ int *block_stable_sort_ids(int M, int nbits, int *bufA, int *bufB, int *s_cnt,
                                      int *s_wave, KeyFn key) {
    const int t = threadIdx.x;
    const int chunk = (M + RS_BS - 1) / RS_BS;
    const int q0 = t * chunk < M ? t * chunk : M;
    const int q1 = q0 + chunk < M ? q0 + chunk : M;
    int *src = bufA, *dst = bufB;
    bool first = true;
    for (int shift = 0; shift < nbits || first; shift += RS_BITS) {
#pragma unroll
        for (int d = 0; d < RS_DIG; ++d) s_cnt[d * RS_BS + t] = 0;
        for (int q = q0; q < q1; ++q) {
            const int id = first ? q : src[q];
            s_cnt[((key(id) >> shift) & (RS_DIG - 1)) * RS_BS + t] += 1;
        }
        __syncthreads();
        // exclusive scan over the digit-major flattening f = d * RS_BS + thread
        int loc[RS_DIG];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < RS_DIG; ++k) { loc[k] = s_cnt[t * RS_DIG + k]; sum += loc[k]; }
        int base = block_excl_scan(sum, s_wave, nullptr);
#pragma unroll
        for (int k = 0; k < RS_DIG; ++k) { s_cnt[t * RS_DIG + k] = base; base += loc[k]; }
        __syncthreads();
        for (int q = q0; q < q1; ++q) {
            const int id = first ? q : src[q];
            int *slot = &s_cnt[((key(id) >> shift) & (RS_DIG - 1)) * RS_BS + t];
            dst[*slot] = id;
            *slot += 1;
        }
        __syncthreads();
        if (first) { first = false; src = bufB; dst = bufA; }
        else { int *tmp = src; src = dst; dst = tmp; }
    }
    return src;
}

static inline int reart_bits_for(int n) {  // number of key bits for keys in [0, n)
    int b = 1;
    while ((1 << b) < n && b < 30) ++b;
    return b;
}
