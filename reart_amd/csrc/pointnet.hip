// reart_amd/csrc/pointnet.hip -- farthest point sampling and ball query for gfx950.
//
// Replaces the two LIVE kernels of the reference's vendored CUDA extension
// (networks/pointnet_lib/src/sampling_gpu.cu:93-209 furthest_point_sampling_kernel,
//  ball_query_gpu.cu:9-45 ball_query_kernel_fast; pybind names in pointnet2_api.cpp:11-26)
// and the CPU fallbacks the reference takes without CUDA
// (networks/pointnet2_utils.py:74-99 farthest_point_sample, :102-140 query_ball_point).
// Both semantics are selectable (cuda_mode); the default is the CPU-fallback rule set
// evaluated with the direct-difference distance (oracle/pointnet.c).
#include "common.h"
#include "internal.h"
#include <math.h>

// ---------------------------------------------------------------------------------------
// FPS: one workgroup per cloud (256 threads up to 1024 points, 512 above); points and running min-distances live in
// registers (pairs of points per thread: the distance update runs on packed fp32), the cloud's coordinates in LDS for
// the broadcast read of the current farthest point.  The op is a chain of npoint dependent rounds, so a round is built
// to be short rather than wide -- per round ONE barrier and no data-dependent branch:
//   every thread   running min-distances of its points, their maximum (values only)
//   every wave     maximum over the wave (six DPP steps), then the smallest tie KEY among the points that attain it
//                  (six more); one (value, key) slot per wave, double buffered
//   after barrier  every wave merges the slots again redundantly (log2(waves) DPP steps for the value, the same for the
//                  key among the slots that attain it)
// The key encodes the reference's choice between equal distances, so exact ties need no slow path:
//   CUDA rule (sampling_gpu.cu's tree arg-max): lowest thread id of the CUDA block, then lowest index -> (k & mask) << 16 | k
//   CPU rule (torch.max(...)[1]): first maximum -> k
// ---------------------------------------------------------------------------------------
typedef float reart_f2 __attribute__((ext_vector_type(2)));

// Distances are sums of squares (>= +0) and the padding is -inf: on such values the order of the floats is the order of
// their bit patterns as signed integers, so the maxima run on v_max_i32 with the DPP operand folded in (fmaxf would add a
// canonicalising instruction per operand).
template <int STEPS>
__device__ __forceinline__ int fps_lanes_max(int v) {
    v = max(v, reart_bfly<0>(v));
    if (STEPS > 1) v = max(v, reart_bfly<1>(v));
    if (STEPS > 2) v = max(v, reart_bfly<2>(v));
    if (STEPS > 3) v = max(v, reart_bfly<3>(v));
    if (STEPS > 4) v = max(v, reart_bfly<4>(v));
    if (STEPS > 5) v = max(v, reart_bfly<5>(v));
    return v;
}
template <int STEPS>
__device__ __forceinline__ int fps_lanes_min(int v) {
    v = min(v, reart_bfly<0>(v));
    if (STEPS > 1) v = min(v, reart_bfly<1>(v));
    if (STEPS > 2) v = min(v, reart_bfly<2>(v));
    if (STEPS > 3) v = min(v, reart_bfly<3>(v));
    if (STEPS > 4) v = min(v, reart_bfly<4>(v));
    if (STEPS > 5) v = min(v, reart_bfly<5>(v));
    return v;
}

// BS threads, NP pairs of points per thread: thread t holds the points t + u * BS, u < 2 NP
template <int BS, int NP, bool CUDA_MODE>
__global__ __launch_bounds__(BS) void fps_kernel(const float *__restrict__ xyz, int N, int M,
                                                 const int *__restrict__ start, int bsmask,
                                                 int *__restrict__ idx32, int64_t *__restrict__ idx64) {
    extern __shared__ __attribute__((aligned(16))) float s_xyz[];  // [N][3]
    constexpr int NW = BS / 64, LG = NW <= 4 ? 2 : (NW <= 8 ? 3 : 4);
    static_assert(NW == 4 || NW == 8 || NW == 16, "the slots' merge walks 2, 3 or 4 butterfly steps");
    __shared__ int s_v[2][NW], s_k[2][NW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *p = xyz + (size_t)b * N * 3;
    for (int e = tid; e < 3 * N; e += BS) s_xyz[e] = p[e];
    reart_f2 px[NP], py[NP], pz[NP], dm[NP];
    int key[2 * NP];
#pragma unroll
    for (int u = 0; u < 2 * NP; ++u) {
        const int k = tid + u * BS;
        const bool ok = k < N;
        px[u >> 1][u & 1] = ok ? p[3 * k] : 0.f; py[u >> 1][u & 1] = ok ? p[3 * k + 1] : 0.f; pz[u >> 1][u & 1] = ok ? p[3 * k + 2] : 0.f;
        dm[u >> 1][u & 1] = ok ? 1e10f : -INFINITY;  // padding can never win the arg-max
        key[u] = CUDA_MODE ? (((k & bsmask) << 16) | k) : k;
    }
    int far = start ? start[b] : 0;
    __syncthreads();
    for (int it = 0; it < M; ++it) {
        if (tid == 0) {
            if (idx32) idx32[(size_t)b * M + it] = far;
            if (idx64) idx64[(size_t)b * M + it] = far;
        }
        if (it == M - 1) break;
        const float fx = s_xyz[3 * far], fy = s_xyz[3 * far + 1], fz = s_xyz[3 * far + 2];
        const reart_f2 fx2 = {fx, fx}, fy2 = {fy, fy}, fz2 = {fz, fz};
        int bv = (int)0xff800000;                                          // -inf
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const reart_f2 dx = px[u] - fx2, dy = py[u] - fy2, dz = pz[u] - fz2;
            const reart_f2 d = (dx * dx + dy * dy) + dz * dz;              // reart_sqdist3, two points at a time
            dm[u].x = d.x < dm[u].x ? d.x : dm[u].x;
            dm[u].y = d.y < dm[u].y ? d.y : dm[u].y;
            bv = max(max(bv, __float_as_int(dm[u].x)), __float_as_int(dm[u].y));
        }
        const int wmax = fps_lanes_max<6>(bv);
        int bk = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < 2 * NP; ++u) bk = min(bk, __float_as_int(dm[u >> 1][u & 1]) == wmax ? key[u] : 0x7fffffff);
        bk = fps_lanes_min<6>(bk);
        const int buf = it & 1;
        if (lane == 0) { s_v[buf][wv] = wmax; s_k[buf][wv] = bk; }
        __syncthreads();
        const int v = s_v[buf][lane & (NW - 1)];
        const int k = s_k[buf][lane & (NW - 1)];
        const int gmax = fps_lanes_max<LG>(v);
        far = __builtin_amdgcn_readfirstlane(fps_lanes_min<LG>(v == gmax ? k : 0x7fffffff)) & 0xffff;
    }
}

static int fps_block_mask(int N) {  // opt_n_threads(N) - 1 (cuda_utils.h:10-14)
    int p = 1;
    while (p * 2 <= N) p *= 2;
    if (p > 1024) p = 1024;
    return p - 1;
}

template <int BS, int NP>
static int fps_launch(const float *xyz, int B, int N, int M, const int *start, int cuda_mode,
                      int *idx32, int64_t *idx64, hipStream_t st) {
    const size_t lds = sizeof(float) * 3 * (size_t)N;
    if (lds > REART_LDS_DEFAULT_CAP) {   // stateless: no function-static "done once" flags in the library
        const void *fn = cuda_mode ? (const void *)fps_kernel<BS, NP, true> : (const void *)fps_kernel<BS, NP, false>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
            return REART_ERR_LAUNCH;
    }
    if (cuda_mode)
        hipLaunchKernelGGL((fps_kernel<BS, NP, true>), dim3(B), dim3(BS), lds, st, xyz, N, M, start,
                           fps_block_mask(N), idx32, idx64);
    else
        hipLaunchKernelGGL((fps_kernel<BS, NP, false>), dim3(B), dim3(BS), lds, st, xyz, N, M, start,
                           fps_block_mask(N), idx32, idx64);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

#ifndef FPS_BS_BIG
#define FPS_BS_BIG 512     // threads per cloud above 1024 points
#endif
extern "C" int reart_fps(const float *xyz, int B, int N, int npoint, const int32_t *start,
                         int cuda_mode, int32_t *idx32, int64_t *idx64, void *stream) {
    if (B < 0 || N < 1 || npoint < 0) return REART_ERR_INVALID_ARG;
    if (B == 0 || npoint == 0) return REART_OK;
    if (!xyz || (!idx32 && !idx64)) return REART_ERR_INVALID_ARG;
    if (N > 12288) return REART_ERR_UNSUPPORTED;  // cloud must fit in LDS (12 B/point)
    hipStream_t st = (hipStream_t)stream;
    if (N <= 512) return fps_launch<256, 1>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
    if (N <= 1024) return fps_launch<256, 2>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
    const int pairs = reart_div_up(N, 2 * FPS_BS_BIG);
    if (pairs <= 2) return fps_launch<FPS_BS_BIG, 2>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
    if (pairs <= 4) return fps_launch<FPS_BS_BIG, 4>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
    if (pairs <= 8) return fps_launch<FPS_BS_BIG, 8>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
    return fps_launch<FPS_BS_BIG, 12 * 512 / FPS_BS_BIG>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
}

// ---------------------------------------------------------------------------------------
// Ball query: one WAVE per query centre.  64 lanes test 64 consecutive points per step
// (coalesced), a ballot gives the in-ball mask, popcount-below-lane gives each hit its slot
// in ascending index order, and the loop stops as soon as nsample slots are filled -- the
// reference kernel walks all N points with ONE thread per centre.
// ---------------------------------------------------------------------------------------
#define BQ_BS 256

template <bool CUDA_MODE>
__global__ __launch_bounds__(BQ_BS) void ball_query_kernel(const float *__restrict__ xyz,
                                                           const float *__restrict__ new_xyz, int N,
                                                           int S, float r2, int nsample,
                                                           int *__restrict__ idx32,
                                                           int64_t *__restrict__ idx64) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * (BQ_BS / 64) + (threadIdx.x >> 6), b = blockIdx.y;
    if (s >= S) return;
    const float *q = new_xyz + 3 * ((size_t)b * S + s);
    const float qx = q[0], qy = q[1], qz = q[2];
    const float sq = (qx * qx + qy * qy) + qz * qz;
    const float *p = xyz + (size_t)b * N * 3;
    const size_t obase = ((size_t)b * S + s) * nsample;
    int cnt = 0, first = -1;
    float nd = INFINITY;  // nearest point so far (lane-local), for the CPU-fallback padding
    int ni = 0x7fffffff;
    for (int k0 = 0; k0 < N && cnt < nsample; k0 += 64) {
        const int k = k0 + lane;
        float d = INFINITY;
        if (k < N) {
            const float px = p[3 * k], py = p[3 * k + 1], pz = p[3 * k + 2];
            if (CUDA_MODE) {
                d = reart_sqdist3(qx, qy, qz, px, py, pz);               // ball_query_gpu.cu:30-33
            } else {
                // square_distance(new_xyz, xyz) of the CPU fallback (networks/pointnet2_utils.py:33-55,126):
                // the matmul expansion with torch's CPU rounding, bit for bit (oracle/pointnet.c)
                const float mm = fmaf(qz, pz, fmaf(qy, py, qx * px));
                d = ((-2.0f * mm) + sq) + ((px * px + py * py) + pz * pz);
            }
        }
        if (!CUDA_MODE && d < nd) { nd = d; ni = k; }
        const bool hit = CUDA_MODE ? (d < r2) : (d <= r2);
        const unsigned long long m = __ballot(hit);
        if (m) {
            if (first < 0) first = k0 + __ffsll((long long)m) - 1;
            const int slot = cnt + __popcll(m & ((1ull << lane) - 1ull));
            if (hit && slot < nsample) {
                if (idx32) idx32[obase + slot] = k;
                if (idx64) idx64[obase + slot] = k;
            }
            cnt += __popcll(m);
        }
    }
    if (cnt >= nsample) return;
    int pad;
    if (CUDA_MODE) {
        pad = first < 0 ? 0 : first;  // first hit; the reference's idx buffer is pre-zeroed
    } else {
        // the early exit did not trigger, so every point has been visited: nearest = arg-min
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float d2 = __shfl_xor(nd, o, 64);
            const int i2 = __shfl_xor(ni, o, 64);
            if (d2 < nd || (d2 == nd && i2 < ni)) { nd = d2; ni = i2; }
        }
        pad = ni;
    }
    for (int l = cnt + lane; l < nsample; l += 64) {
        if (idx32) idx32[obase + l] = pad;
        if (idx64) idx64[obase + l] = pad;
    }
}

extern "C" int reart_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S,
                                double radius, int nsample, int cuda_mode, int32_t *idx32,
                                int64_t *idx64, void *stream) {
    if (B < 0 || N < 1 || S < 0 || nsample < 1 || !(radius >= 0.0)) return REART_ERR_INVALID_ARG;
    if (B == 0 || S == 0) return REART_OK;
    if (!xyz || !new_xyz || (!idx32 && !idx64)) return REART_ERR_INVALID_ARG;
    const dim3 grid(reart_div_up(S, BQ_BS / 64), B);
    hipStream_t st = (hipStream_t)stream;
    if (cuda_mode) {
        const float r2 = (float)radius * (float)radius;  // ball_query_gpu.cu:22 (float radius)
        hipLaunchKernelGGL(ball_query_kernel<true>, grid, dim3(BQ_BS), 0, st, xyz, new_xyz, N, S, r2, nsample,
                           idx32, idx64);
    } else {
        const float r2 = (float)(radius * radius);  // python `radius ** 2` -> float32, pointnet2_utils.py:129
        hipLaunchKernelGGL(ball_query_kernel<false>, grid, dim3(BQ_BS), 0, st, xyz, new_xyz, N, S, r2, nsample,
                           idx32, idx64);
    }
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The channel-major gather / interpolate operators of the reference's pybind module that nothing in the reference calls
// (networks/pointnet_lib/src/pointnet2_api.cpp:11-26; reachable only through pointnet2_modules.py, which is never imported).
// Pure index work, bound by memory: a thread per output element, the index row read once per (batch, point) and reused
// across channels by the cache; the backward forms accumulate with float atomics like the reference's (their sums depend on
// the order of arrival there as here).
//   gather:  out[b][c][m] = points[b][c][idx[b][m]]                      (sampling_gpu.cu:8-24; group_points_gpu.cu:39-54 is
//            the same map with m = point * nsample + sample)
//   scatter: grad_points[b][c][idx[b][m]] += grad_out[b][c][m]           (sampling_gpu.cu:46-63; group_points_gpu.cu:8-21)
__global__ __launch_bounds__(256) void pn2_gather_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx, int C, int N,
                                                         int M, float *__restrict__ out) {
    const int m = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (m >= M) return;
    out[((size_t)b * C + c) * M + m] = points[((size_t)b * C + c) * N + idx[(size_t)b * M + m]];
}
__global__ __launch_bounds__(256) void pn2_scatter_add_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ idx, int C, int N,
                                                              int M, float *__restrict__ grad_points) {
    const int m = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (m >= M) return;
    atomicAdd(grad_points + ((size_t)b * C + c) * N + idx[(size_t)b * M + m], grad_out[((size_t)b * C + c) * M + m]);
}
//   interpolate: out[b][c][n] = (w0 p[i0] + w1 p[i1]) + w2 p[i2]          (interpolate_gpu.cu:149-169), p = points[b][c][:]
//   its backward: grad_points[b][c][i_j] += grad_out[b][c][n] * w_j       (interpolate_gpu.cu:192-214)
__global__ __launch_bounds__(256) void pn2_interp_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                                         const float *__restrict__ weight, int C, int M, int N, float *__restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (n >= N) return;
    const int32_t *i = idx + ((size_t)b * N + n) * 3;
    const float *w = weight + ((size_t)b * N + n) * 3, *p = points + ((size_t)b * C + c) * M;
    out[((size_t)b * C + c) * N + n] = (w[0] * p[i[0]] + w[1] * p[i[1]]) + w[2] * p[i[2]];
}
__global__ __launch_bounds__(256) void pn2_interp_grad_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
                                                              const float *__restrict__ weight, int C, int N, int M,
                                                              float *__restrict__ grad_points) {
    const int n = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (n >= N) return;
    const int32_t *i = idx + ((size_t)b * N + n) * 3;
    const float *w = weight + ((size_t)b * N + n) * 3;
    const float g = grad_out[((size_t)b * C + c) * N + n];
    float *gp = grad_points + ((size_t)b * C + c) * M;
    atomicAdd(gp + i[0], g * w[0]); atomicAdd(gp + i[1], g * w[1]); atomicAdd(gp + i[2], g * w[2]);
}

static bool pn2_grid_ok(int B, int C) { return B >= 1 && C >= 1 && B <= 65535 && C <= 65535; }

extern "C" int reart_pn2_gather_points(const float *points, const int32_t *idx, int B, int C, int N, int M, float *out, void *stream) {
    if (B < 0 || C < 0 || N < 1 || M < 0) return REART_ERR_INVALID_ARG;
    if (B == 0 || C == 0 || M == 0) return REART_OK;
    if (!points || !idx || !out || !pn2_grid_ok(B, C)) return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pn2_gather_kernel, dim3(reart_div_up(M, 256), C, B), dim3(256), 0, (hipStream_t)stream, points, idx, C, N, M, out);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
extern "C" int reart_pn2_gather_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int M, float *grad_points,
                                            void *stream) {
    if (B < 0 || C < 0 || N < 1 || M < 0) return REART_ERR_INVALID_ARG;
    if (B == 0 || C == 0 || M == 0) return REART_OK;
    if (!grad_out || !idx || !grad_points || !pn2_grid_ok(B, C)) return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pn2_scatter_add_kernel, dim3(reart_div_up(M, 256), C, B), dim3(256), 0, (hipStream_t)stream, grad_out, idx, C, N, M,
                       grad_points);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
extern "C" int reart_pn2_three_interpolate(const float *points, const int32_t *idx, const float *weight, int B, int C, int M, int N,
                                           float *out, void *stream) {
    if (B < 0 || C < 0 || M < 1 || N < 0) return REART_ERR_INVALID_ARG;
    if (B == 0 || C == 0 || N == 0) return REART_OK;
    if (!points || !idx || !weight || !out || !pn2_grid_ok(B, C)) return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pn2_interp_kernel, dim3(reart_div_up(N, 256), C, B), dim3(256), 0, (hipStream_t)stream, points, idx, weight, C, M, N,
                       out);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
extern "C" int reart_pn2_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight, int B, int C, int N, int M,
                                                float *grad_points, void *stream) {
    if (B < 0 || C < 0 || M < 1 || N < 0) return REART_ERR_INVALID_ARG;
    if (B == 0 || C == 0 || N == 0) return REART_OK;
    if (!grad_out || !idx || !weight || !grad_points || !pn2_grid_ok(B, C)) return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pn2_interp_grad_kernel, dim3(reart_div_up(N, 256), C, B), dim3(256), 0, (hipStream_t)stream, grad_out, idx, weight, C,
                       N, M, grad_points);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
