// reart_amd/csrc/kinpost.hip -- what the kinematic projection's iteration does between the assignment re-solve and the FK backward
// (reference run_robot.py:165-209 for `--model kinematic --use_assign_loss [--use_flow_loss]`): the matched pairs' loss and its
// gradient (:177-184), the flow targets of every frame pair (blend_anchor_motion x (T-1), :194-201), the flow loss (:203-209,
// networks/loss.py:10-21) and dL/d pc_trans of both -- as ONE entry point of nine launches.  reart_amd/kinematic_engine.py used to
// issue this part as ~45 launches (36 of them the per-frame searches and blends, the rest ATen tensor expressions: cat / sub /
// gather / index_put / mul / add): with the re-solve's median at 1.1 ms they were as long as the solve (round 6).
//
// The values are the ones the tensor expressions produced, operation for operation: dL/d pc_trans is bit-identical to the former
// path; the two loss sums are accumulated in double precision (the former path: torch's fp32 tree sums).
#include "common.h"
#include "internal.h"
#include <math.h>

struct KinPostArgs {
    const float *pc_trans;   // [B][N][3] the articulated frames (canonical frame not among them)
    const float *cano;       // [N][3]
    int B, N, c;             // c = cano_idx: comp = pc_trans[:c] | cano | pc_trans[c:]   ([T = B + 1][N][3])
    float *comp, *pred;      // [T][N][3] | [B][N][3] = comp[f + 1] - comp[f]
    // assignment branch
    const float *pc_src;     // [B][n][3] the sampled source points as the solve saw them
    const float *tgt;        // [B][n][3]
    const int *cols;         // [B][n] the optimum
    const int *slot;         // [N] sample slot of canonical point p, or -1
    int n;
    float two_lambda, lambda_assign;
    // flow branch (gp null: none)
    const float *gp;         // [B][N][3] d flow loss / d pred
    const float *loss_flow;  // device scalar (unscaled flow loss)
    float lambda_flow;
    float *G;                // [B][N][3] out: dL / d pc_trans
    float *matched;          // [B][n][3] out (nullable): the matched target of every sampled point
    double *partial;         // [blocks] the assignment loss's partial sums
    float *losses;           // [3] out: lambda_assign x assignment loss | lambda_flow x flow loss | total
};

__global__ __launch_bounds__(256) void kin_comp_kernel(KinPostArgs a) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x, per = (size_t)a.N * 3, total = (size_t)(a.B + 1) * per;
    if (e >= total) return;
    const int ci = (int)(e / per);
    const size_t r = e - (size_t)ci * per;
    auto at = [&](int f) { return f < a.c ? a.pc_trans[(size_t)f * per + r] : (f == a.c ? a.cano[r] : a.pc_trans[(size_t)(f - 1) * per + r]); };
    const float v = at(ci);
    a.comp[e] = v;
    if (ci < a.B) a.pred[e] = at(ci + 1) - v;
}

#define KP_BS 256
__global__ __launch_bounds__(KP_BS) void kin_grad_kernel(KinPostArgs a) {
    __shared__ double s_red[KP_BS / 64];
    const int fi = blockIdx.y, p = blockIdx.x * KP_BS + threadIdx.x;
    double acc = 0.0;
    if (p < a.N) {
        const size_t per = (size_t)a.N * 3, o = (size_t)fi * per + 3 * (size_t)p;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        const int sl = a.slot ? a.slot[p] : -1;
        if (sl >= 0) {
            const size_t so = ((size_t)fi * a.n + sl) * 3;
            const int cj = a.cols[(size_t)fi * a.n + sl];
            const float *t = a.tgt + ((size_t)fi * a.n + cj) * 3;
            const float d0 = a.pc_src[so] - t[0], d1 = a.pc_src[so + 1] - t[1], d2 = a.pc_src[so + 2] - t[2];
            g0 = a.two_lambda * d0; g1 = a.two_lambda * d1; g2 = a.two_lambda * d2;
            acc = (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2);
            if (a.matched) { a.matched[so] = t[0]; a.matched[so + 1] = t[1]; a.matched[so + 2] = t[2]; }
        }
        if (a.gp) {
            // pred = comp[1:] - comp[:-1]: +gp to frame ci of the later pair, -gp of the earlier one; the canonical frame takes none
            const int ci = fi < a.c ? fi : fi + 1;
            const size_t r = 3 * (size_t)p;
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                float t = ci >= 1 ? a.gp[(size_t)(ci - 1) * per + r + x] * a.lambda_flow : 0.f;
                if (ci <= a.B - 1) t = t - a.gp[(size_t)ci * per + r + x] * a.lambda_flow;
                if (x == 0) g0 = g0 + t; else if (x == 1) g1 = g1 + t; else g2 = g2 + t;
            }
        }
        a.G[o] = g0; a.G[o + 1] = g1; a.G[o + 2] = g2;
    }
    acc = reart_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < KP_BS / 64; ++w) t += s_red[w];
        a.partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(64) void kin_loss_kernel(KinPostArgs a, int blocks) {
    double t = 0.0;
    for (int e = threadIdx.x; e < blocks; e += 64) t += a.partial[e];
    t = reart_wave_sum_d(t);
    if (threadIdx.x == 0) {
        const float ass = a.lambda_assign * (float)t;
        const float fl = a.gp ? *a.loss_flow * a.lambda_flow : 0.f;
        a.losses[0] = ass; a.losses[1] = fl; a.losses[2] = ass + fl;
    }
}

static size_t kp_layout(int B, int N, int nr_max, int k, bool flow, size_t *o_comp, size_t *o_pred, size_t *o_gt, size_t *o_mask, size_t *o_gp,
                        size_t *o_lf, size_t *o_part, size_t *o_fl, size_t *o_blend) {
    size_t off = 0;
    const size_t per = sizeof(float) * (size_t)N * 3;
    *o_comp = off; off += reart_align_up(per * (B + 1), 256);
    *o_pred = off; off += reart_align_up(per * B, 256);
    *o_gt = off; off += reart_align_up(per * B, 256);
    *o_mask = off; off += reart_align_up((size_t)B * N, 256);
    *o_gp = off; off += reart_align_up(per * B, 256);
    *o_lf = off; off += 256;
    *o_part = off; off += reart_align_up(sizeof(double) * (size_t)B * reart_div_up(N, KP_BS), 256);
    *o_fl = off; off += flow ? reart_align_up(reart_flow_loss_workspace_bytes(), 256) : 0;
    *o_blend = off; off += flow ? reart_blend_anchor_motion_batch_workspace_bytes(B, N, nr_max, k) : 0;
    return off;
}

extern "C" size_t reart_kin_post_workspace_bytes(int B, int N, int nr_max, int k) {
    if (B <= 0 || N <= 0) return 0;
    size_t o[9];
    return kp_layout(B, N, nr_max > 0 ? nr_max : 1, k > 0 ? k : 1, nr_max > 0, o, o + 1, o + 2, o + 3, o + 4, o + 5, o + 6, o + 7, o + 8);
}

extern "C" int reart_kin_post(const float *pc_trans, const float *cano, int B, int N, int cano_idx, const float *pc_src, const float *tgt,
                              const int32_t *cols, const int32_t *slot_of_point, int n, float lambda_assign, const float *ref,
                              const float *ref_flow, const int64_t *ref_len, int nr_max, int k, int euclidean, float lambda_flow, int robust,
                              float smooth_weight, float *G, float *matched, float *losses, void *workspace, size_t workspace_bytes,
                              void *stream) {
    if (B < 1 || N < 1 || n < 0 || cano_idx < 0 || cano_idx > B) return REART_ERR_INVALID_ARG;
    if (!pc_trans || !cano || !G || !losses || !workspace) return REART_ERR_INVALID_ARG;
    if (n > 0 && (!pc_src || !tgt || !cols || !slot_of_point)) return REART_ERR_INVALID_ARG;
    const bool flow = ref != nullptr;
    if (flow && (!ref_flow || nr_max < k || k < 1)) return REART_ERR_INVALID_ARG;
    size_t o_comp, o_pred, o_gt, o_mask, o_gp, o_lf, o_part, o_fl, o_blend;
    if (workspace_bytes < kp_layout(B, N, flow ? nr_max : 1, flow ? k : 1, flow, &o_comp, &o_pred, &o_gt, &o_mask, &o_gp, &o_lf, &o_part, &o_fl, &o_blend))
        return REART_ERR_INVALID_ARG;
    char *ws = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    KinPostArgs a{};
    a.pc_trans = pc_trans; a.cano = cano; a.B = B; a.N = N; a.c = cano_idx;
    a.comp = (float *)(ws + o_comp); a.pred = (float *)(ws + o_pred);
    a.pc_src = pc_src; a.tgt = tgt; a.cols = cols; a.slot = n > 0 ? slot_of_point : nullptr; a.n = n;
    a.two_lambda = (float)(2.0 * (double)lambda_assign); a.lambda_assign = lambda_assign;
    a.gp = flow ? (const float *)(ws + o_gp) : nullptr; a.loss_flow = (const float *)(ws + o_lf); a.lambda_flow = lambda_flow;
    a.G = G; a.matched = matched; a.partial = (double *)(ws + o_part); a.losses = losses;
    if (flow) {
        const size_t total = (size_t)(B + 1) * N * 3;
        hipLaunchKernelGGL(kin_comp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
        REART_CHECK_LAUNCH();
        float *gt = (float *)(ws + o_gt);
        uint8_t *mask = (uint8_t *)(ws + o_mask);
        int rc = reart_blend_anchor_motion_batch(a.comp, ref, ref_flow, ref_len, B, N, nr_max, k, euclidean, gt, mask, ws + o_blend,
                                                 workspace_bytes - o_blend, stream);
        if (rc != REART_OK) return rc;
        rc = reart_flow_loss(gt, a.pred, mask, B, N, robust, smooth_weight, (float *)(ws + o_lf), (float *)(ws + o_gp), ws + o_fl,
                             reart_flow_loss_workspace_bytes(), stream);
        if (rc != REART_OK) return rc;
    }
    const int gx = reart_div_up(N, KP_BS);
    hipLaunchKernelGGL(kin_grad_kernel, dim3(gx, B), dim3(KP_BS), 0, st, a);
    REART_CHECK_LAUNCH();
    hipLaunchKernelGGL(kin_loss_kernel, dim3(1), dim3(64), 0, st, a, gx * B);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
