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
// FPS: one 1024-thread workgroup per cloud; points and running min-distances live in
// registers (IPT per thread), the cloud's coordinates in LDS for the broadcast read of the
// current farthest point; per round ONE barrier: wave arg-max by shuffles, 16 wave winners
// through a double-buffered LDS slot, every wave reduces the 16 again redundantly.
// The op is inherently sequential in npoint (latency bound): ~1 us per round.
// ---------------------------------------------------------------------------------------
#define FPS_BS 1024

struct FpsKey { float v; int i; };

template <bool CUDA_MODE>
__device__ __forceinline__ bool fps_better(float v2, int i2, float v, int i, int bsmask) {
    if (v2 > v) return true;
    if (v2 < v) return false;
    if (CUDA_MODE) {  // tree arg-max of the CUDA kernel: lowest thread id, then lowest index
        const int t2 = i2 & bsmask, t = i & bsmask;
        return (t2 < t) || (t2 == t && i2 < i);
    }
    return i2 < i;    // torch.max(...)[1]: first maximum
}

template <int IPT, bool CUDA_MODE>
__global__ __launch_bounds__(FPS_BS) void fps_kernel(const float *__restrict__ xyz, int N, int M,
                                                     const int *__restrict__ start, int bsmask,
                                                     int *__restrict__ idx32, int64_t *__restrict__ idx64) {
    extern __shared__ __attribute__((aligned(16))) float s_xyz[];  // [N][3]
    __shared__ float s_v[2][FPS_BS / 64];
    __shared__ int s_i[2][FPS_BS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float *p = xyz + (size_t)b * N * 3;
    for (int e = tid; e < 3 * N; e += FPS_BS) s_xyz[e] = p[e];
    float px[IPT], py[IPT], pz[IPT], dm[IPT];
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int k = tid + u * FPS_BS;
        const bool ok = k < N;
        px[u] = ok ? p[3 * k] : 0.f; py[u] = ok ? p[3 * k + 1] : 0.f; pz[u] = ok ? p[3 * k + 2] : 0.f;
        dm[u] = ok ? 1e10f : -INFINITY;  // padding can never win the arg-max
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
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            const float d = reart_sqdist3(px[u], py[u], pz[u], fx, fy, fz);
            dm[u] = d < dm[u] ? d : dm[u];
            const int k = tid + u * FPS_BS;
            if (fps_better<CUDA_MODE>(dm[u], k, bv, bi, bsmask)) { bv = dm[u]; bi = k; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (fps_better<CUDA_MODE>(v2, i2, bv, bi, bsmask)) { bv = v2; bi = i2; }
        }
        const int buf = it & 1;
        if (lane == 0) { s_v[buf][wv] = bv; s_i[buf][wv] = bi; }
        __syncthreads();
        bv = s_v[buf][lane & (FPS_BS / 64 - 1)];
        bi = s_i[buf][lane & (FPS_BS / 64 - 1)];
#pragma unroll
        for (int o = FPS_BS / 128; o >= 1; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (fps_better<CUDA_MODE>(v2, i2, bv, bi, bsmask)) { bv = v2; bi = i2; }
        }
        far = bi;
    }
}

static int fps_block_mask(int N) {  // opt_n_threads(N) - 1 (cuda_utils.h:10-14)
    int p = 1;
    while (p * 2 <= N) p *= 2;
    if (p > 1024) p = 1024;
    return p - 1;
}

template <int IPT>
static int fps_launch(const float *xyz, int B, int N, int M, const int *start, int cuda_mode,
                      int *idx32, int64_t *idx64, hipStream_t st) {
    const size_t lds = sizeof(float) * 3 * (size_t)N;
    if (lds > REART_LDS_DEFAULT_CAP) {   // stateless: no function-static "done once" flags in the library
        const void *fn = cuda_mode ? (const void *)fps_kernel<IPT, true> : (const void *)fps_kernel<IPT, false>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
            return REART_ERR_LAUNCH;
    }
    if (cuda_mode)
        hipLaunchKernelGGL((fps_kernel<IPT, true>), dim3(B), dim3(FPS_BS), lds, st, xyz, N, M, start,
                           fps_block_mask(N), idx32, idx64);
    else
        hipLaunchKernelGGL((fps_kernel<IPT, false>), dim3(B), dim3(FPS_BS), lds, st, xyz, N, M, start,
                           fps_block_mask(N), idx32, idx64);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

extern "C" int reart_fps(const float *xyz, int B, int N, int npoint, const int32_t *start,
                         int cuda_mode, int32_t *idx32, int64_t *idx64, void *stream) {
    if (B < 0 || N < 1 || npoint < 0) return REART_ERR_INVALID_ARG;
    if (B == 0 || npoint == 0) return REART_OK;
    if (!xyz || (!idx32 && !idx64)) return REART_ERR_INVALID_ARG;
    if (N > 12288) return REART_ERR_UNSUPPORTED;  // cloud must fit in LDS (12 B/point)
    hipStream_t st = (hipStream_t)stream;
    const int ipt = reart_div_up(N, FPS_BS);
    if (ipt <= 1) return fps_launch<1>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
    if (ipt <= 2) return fps_launch<2>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
    if (ipt <= 4) return fps_launch<4>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
    if (ipt <= 8) return fps_launch<8>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
    return fps_launch<12>(xyz, B, N, npoint, start, cuda_mode, idx32, idx64, st);
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
