// reart_amd/csrc/assign.hip -- the device-side glue of an assignment refresh (reference run_robot.py:165-178): the sampled
// source points of the moved canonical cloud in, the matched target of every sampled point out.  The solve between the two
// is reart_lap_resolve_points_mc (lap.hip); with these two launches a refresh touches the host only for the B certificate
// flags.
#include "common.h"

// out[b][r] = pc[b][index[r]]   (run_robot.py:169 index_points(pc_trans_list, fps_idx): one FPS sample shared by all frames)
__global__ __launch_bounds__(256) void gather_points_kernel(const float *__restrict__ pc, const int32_t *__restrict__ index,
                                                            int N, int n, float *__restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (r >= n) return;
    const float *p = pc + ((size_t)b * N + index[r]) * 3;
    float *o = out + ((size_t)b * n + r) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

extern "C" int reart_gather_points(const float *pc, const int32_t *index, int B, int N, int n, float *out, void *stream) {
    if (!pc || !index || !out || B < 1 || N < 1 || n < 1) return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gather_points_kernel, dim3(reart_div_up(n, 256), B), dim3(256), 0, (hipStream_t)stream, pc, index, N, n, out);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// assign_map[b][p] = the target point matched to canonical point p in frame b, -1 for a point outside the sample
// (run_robot.py:177-178: pc_tgt gathered by the solver's columns, paired with the sampled source points in order)
__global__ __launch_bounds__(256) void assign_pairs_kernel(const int32_t *__restrict__ col4row, const int32_t *__restrict__ slot_of_point,
                                                           const int32_t *__restrict__ tgt_index, int N, int n,
                                                           int32_t *__restrict__ assign_map) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= N) return;
    const int r = slot_of_point[p];
    int v = -1;
    if (r >= 0) {
        const int c = col4row[(size_t)b * n + r];
        v = (c >= 0 && c < n) ? tgt_index[(size_t)b * n + c] : -1;
    }
    assign_map[(size_t)b * N + p] = v;
}

extern "C" int reart_assign_pairs(const int32_t *col4row, const int32_t *slot_of_point, const int32_t *tgt_index, int B, int N,
                                  int n, int32_t *assign_map, void *stream) {
    if (!col4row || !slot_of_point || !tgt_index || !assign_map || B < 1 || N < 1 || n < 1) return REART_ERR_INVALID_ARG;
    hipLaunchKernelGGL(assign_pairs_kernel, dim3(reart_div_up(N, 256), B), dim3(256), 0, (hipStream_t)stream, col4row,
                       slot_of_point, tgt_index, N, n, assign_map);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// The words a refresh's host waits for -- certificate flags, tie flags, the solver's statistics -- written by ONE launch straight
// into pinned host memory (the pointer hipHostMalloc returned is valid on the device), in the order a | b | c: three device-to-
// host copies were three launches of ~5 us each at the end of every refresh, behind which the host's wake-up waits.
__global__ __launch_bounds__(256) void publish_words_kernel(const int32_t *__restrict__ a, int na, const int32_t *__restrict__ b, int nb,
                                                            const int32_t *__restrict__ c, int nc, int32_t *__restrict__ out) {
    for (int i = threadIdx.x; i < na + nb + nc; i += 256)
        out[i] = i < na ? a[i] : (i < na + nb ? b[i - na] : c[i - na - nb]);
}

extern "C" int reart_publish_words(const int32_t *a, int na, const int32_t *b, int nb, const int32_t *c, int nc, int32_t *host_out,
                                   void *stream) {
    if (na < 0 || nb < 0 || nc < 0 || (na && !a) || (nb && !b) || (nc && !c) || !host_out) return REART_ERR_INVALID_ARG;
    if (na + nb + nc == 0) return REART_OK;
    hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, na, b, nb, c, nc, host_out);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
