// reart_amd/csrc/step.hip -- the fused relaxation iteration (reference run_robot.py:154-221,
// Chamfer + flow branch): forward, both Chamfer directions, k=3 flow blending, losses, the
// whole backward and the Adam update as a fixed sequence of five kernel launches (forward, both
// searches, both of their consumers, backward, finalize + Adam + bookkeeping) with NO host
// interaction -- iteration counter, Gumbel RNG offset, temperature and loss log all live in
// device memory, so the sequence can be captured once in a hipGraph and replayed.
//
// The reference does, per iteration: ~40 PyTorch ops for the model, 2 knn_points launches,
// 19 KNN(k=3) launches + ~200 small ops for the flow blend, autograd, Adam, and 2-4
// .cpu() syncs plus a print (run_robot.py:186,190,212,216-217).
#include "common.h"
#include "internal.h"
#include "blocksort.h"
#include <math.h>
#include <string.h>

// ------------------------------------------------------------------------------ layout
struct StepPlan {
    int S1, L1, Npad;        // Chamfer K=1: slices, slice length, padded SoA row
    int S3, Mpad;            // flow K=3: slices, padded SoA row of the reference sets
    int nchunk, nparams;
    size_t o_ysoa, o_xsoa, o_rsoa, o_rlen, o_qmap;
    size_t o_pd0, o_pi0, o_pd1, o_pi1, o_pd3, o_pi3;
    size_t o_yT, o_hT, o_hard, o_rt, o_G, o_gpf, o_fx, o_floss, o_fpart, o_grads, o_bwd;
    size_t o_bc;              // Adam bias corrections of the coming step (double[2])
    size_t o_ticket;          // last-workgroup ticket of the finalize kernel (zero between launches)
    size_t o_boxY, o_boxX, o_boxR;  // AABBs of every NN_BOX targets (block-skip test)
    int pruned;               // box-pruned, warm-started search (prune.hip) instead of the brute-force slice kernels
    int W1, W3;               // pruned: waves (box slices) per search workgroup, K = 1 / K = 3
    int sparse;               // pruned: sparse-scan limit
    size_t o_seed3;           // last iteration's flow neighbour indices [B,N,3] (the Chamfer seeds are the K = 1 records themselves)
    size_t o_border;          // [3][B * nqg] launch order of the (frame, query group) pairs of the three search jobs
    size_t o_cost;            // [3][B * nqg] work counts of the last search launch, by launch position
    size_t o_prof, o_prof_pairs, o_prof_acc;   // cfg.profile: per-workgroup stamps of the search launch and their accumulators
    size_t o_gridY, o_gridR;  // exact-search grids over pc_list and the flow reference sets
    int gstrideY, gstrideR;
    size_t bwd_bytes, total;
};

static size_t take(size_t &off, size_t bytes) {
    const size_t o = off;
    off += reart_align_up(bytes, 256);
    return o;
}

#define FLOW_BS 256    // stand-alone flow_blend_kernel; inside post_kernel the blend uses CG_BS threads
#define CG_BS 1024
// points (= live threads) of a flow-blend workgroup inside post_kernel.  Same-box A/B of the headline (tools/ab_headline.sh),
// 1024 / 512 / 256: 12 560 / 12 950 / 12 520 it/s -- at 256 the blend workgroup lives 6.1 us instead of 10, but the 304 of them
// delay the Chamfer-gradient workgroups dispatched behind them (7.1 -> 8.3 us); dispatching those first changes nothing at 512.
#define POST_FBS 512
#define CG_RANGE 1024   // targets x_j owned by one workgroup of chamfer_grad_kernel
static_assert(CG_BS == CG_RANGE, "chamfer_grad_kernel: thread tid owns target r0 + tid");

static int step_plan(const reart_relax_config *c, StepPlan *p) {
    if (c->N <= 0 || c->P <= 0 || c->P > 32 || c->B <= 0 || c->H <= 0) return REART_ERR_INVALID_ARG;
    if (c->use_flow && (c->flow_k != 3 || c->M_max < 3)) return REART_ERR_UNSUPPORTED;
    const long waves1 = 2L * c->B * reart_div_up(c->N, NN_BS);
    // 1: exact box-pruned, warm-started search (prune.hip, default); 0: cold brute-force slices (search_mode 1, the
    // grid variant, or clouds that are not spatially sorted) -- same results, A/B and in-situ parity checks
    p->pruned = (c->use_boxes && !c->use_grid && c->search_mode != 1) ? 1 : 0;
    // pruned: ONE record per query (S = 1 in the consumers); the box slices are the waves of a search workgroup
    p->W1 = (c->tune_slices >= 1 && c->tune_slices <= 4) ? c->tune_slices : 3;
    p->W3 = (c->tune_slices_flow >= 1 && c->tune_slices_flow <= 4) ? c->tune_slices_flow : p->W1;
    p->sparse = c->tune_sparse < 0 ? 0 : (c->tune_sparse == 0 ? 40 : (c->tune_sparse > 64 ? 64 : c->tune_sparse));
    p->S1 = p->pruned ? 1 : reart_knn_pick_split(waves1, c->N, 1);
    p->L1 = (int)reart_align_up((size_t)reart_div_up(c->N, p->S1), NN_BOX);
    p->Npad = p->L1 * p->S1;
    p->S3 = 1; p->Mpad = 0;
    if (c->use_flow) {
        const long waves3 = (long)c->B * reart_div_up(c->N, NN_BS);
        p->S3 = p->pruned ? 1 : reart_knn_pick_split(waves3, c->M_max, 3);
        p->Mpad = (int)reart_align_up((size_t)reart_div_up(c->M_max, p->S3), NN_BOX) * p->S3;
    }
    p->nchunk = reart_div_up(c->N, 64);
    p->nparams = 3 * c->H + c->H + c->P * c->H + 6 * c->B * c->P + 3 * c->B * c->P;
    size_t off = 0;
    const size_t BN = (size_t)c->B * c->N;
    p->o_ysoa = take(off, sizeof(float) * 3 * c->B * (size_t)p->Npad);
    p->o_xsoa = take(off, sizeof(float) * 3 * c->B * (size_t)p->Npad);
    p->o_rsoa = take(off, sizeof(float) * 3 * c->B * (size_t)p->Mpad);
    p->o_rlen = take(off, sizeof(int) * c->B);
    p->o_qmap = take(off, sizeof(int) * (c->B + 1));
    p->o_pd0 = take(off, sizeof(float) * p->S1 * BN);
    p->o_pi0 = take(off, sizeof(int) * p->S1 * BN);
    p->o_pd1 = take(off, sizeof(float) * p->S1 * BN);
    p->o_pi1 = take(off, sizeof(int) * p->S1 * BN);
    p->o_pd3 = take(off, c->use_flow ? sizeof(float) * p->S3 * BN * 3 : 0);
    p->o_pi3 = take(off, c->use_flow ? sizeof(int) * p->S3 * BN * 3 : 0);
    p->o_yT = take(off, sizeof(float) * (size_t)c->P * c->N);
    p->o_hT = take(off, sizeof(float) * (size_t)c->H * c->N);
    p->o_hard = take(off, sizeof(int) * (size_t)c->N);
    p->o_rt = take(off, sizeof(float) * 12 * (size_t)c->B * c->P);
    p->o_G = take(off, sizeof(float) * 3 * BN);
    p->o_gpf = take(off, sizeof(float) * 3 * BN);
    p->o_fx = take(off, sizeof(double));
    p->o_floss = take(off, sizeof(double) * (size_t)c->B * reart_div_up(c->N, CG_RANGE));
    p->o_fpart = take(off, sizeof(double) * (size_t)c->B * reart_div_up(c->N, FLOW_BS));
    p->o_grads = take(off, sizeof(float) * p->nparams);
    p->bwd_bytes = reart_base_backward_workspace_bytes(c->N, c->P, c->B, c->H);
    p->o_bwd = take(off, p->bwd_bytes);
    p->o_bc = take(off, 2 * sizeof(double));
    p->o_ticket = take(off, 4 * sizeof(unsigned int));   // [0] finalize ticket, [2..3] persistent-search counters
    p->o_boxY = take(off, sizeof(float) * 8 * (size_t)c->B * (p->Npad / NN_BOX));
    p->o_boxX = take(off, sizeof(float) * 8 * (size_t)c->B * (p->Npad / NN_BOX));
    p->o_boxR = take(off, sizeof(float) * 8 * (size_t)c->B * (p->Mpad / NN_BOX + 1));
    p->o_seed3 = take(off, (p->pruned && c->use_flow) ? sizeof(int) * BN * 3 : 0);
    const size_t G = (size_t)c->B * reart_div_up(c->N, NN_BS);
    p->o_border = take(off, sizeof(int) * 3 * G);
    p->o_cost = take(off, sizeof(unsigned int) * 3 * G);
    const size_t nprof = (c->profile && p->pruned) ? (size_t)reart_search_grid(2, 1, (int)G) : 0;
    p->o_prof = take(off, sizeof(unsigned long long) * 2 * nprof);
    p->o_prof_pairs = take(off, sizeof(unsigned int) * nprof);
    p->o_prof_acc = take(off, sizeof(unsigned long long) * 4);
    p->gstrideY = (int)reart_align_up((size_t)c->N, 64);
    p->gstrideR = (int)reart_align_up((size_t)(c->M_max > 0 ? c->M_max : 1), 64);
    p->o_gridY = take(off, c->use_grid ? reart_grid_bytes(c->B, p->gstrideY) : 0);
    p->o_gridR = take(off, (c->use_grid && c->use_flow) ? reart_grid_bytes(c->B, p->gstrideR) : 0);
    p->total = off;
    return REART_OK;
}

extern "C" size_t reart_relax_workspace_bytes(const reart_relax_config *cfg) {
    StepPlan p;
    if (!cfg || step_plan(cfg, &p) != REART_OK) return 0;
    return p.total;
}

// ------------------------------------------------------------------------------ prepare
__global__ void relax_init_kernel(reart_relax_config c, const int *__restrict__ ref_off,
                                  int *__restrict__ rlen, int *__restrict__ qmap,
                                  int64_t *__restrict__ iter, float *__restrict__ tau,
                                  const float *__restrict__ pc_list, int *__restrict__ fx_bits,
                                  double *__restrict__ bias_corr, int *__restrict__ seed3, int *__restrict__ border) {
    __shared__ float s_max[1024];
    const int t = threadIdx.x;
    // warm start of the first k=3 search: any 3 distinct valid indices
    if (seed3)
        for (size_t e = t; e < (size_t)c.B * c.N * 3; e += 1024) seed3[e] = (int)(e % 3);
    // fixed-point scale for the exact (order-independent) sums of observed points in
    // chamfer_grad_kernel: N * max|y| * scale < 2^61
    float mx = 0.f;
    for (size_t e = t; e < (size_t)c.B * c.N * 3; e += 1024) mx = fmaxf(mx, fabsf(pc_list[e]));
    s_max[t] = mx;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if (t < o) s_max[t] = fmaxf(s_max[t], s_max[t + o]);
        __syncthreads();
    }
    if (t == 0) {
        // sums of up to N differences |x - y| (allow 8 x the data range): N * 8 max|y| * 2^bits < 2^61, and
        // bits <= 39 so that a 24-bit mantissa never overflows the shift
        const double bound = 8.0 * (double)c.N * fmax((double)s_max[0], 1e-30);
        int bits = (int)floor(61.0 - log2(bound));
        fx_bits[0] = bits > 39 ? 39 : (bits < 0 ? 0 : bits);
    }
    if (t < c.B) {
        // Launch order of the search items: position k = pair k (frame-major), so that every XCD's eighth of the
        // positions is a run of consecutive frames; the consumer launch re-sorts every eighth by measured work.
        {
            const int nqg = (c.N + NN_BS - 1) / NN_BS, ng = c.B * nqg;
            for (int g = 0; g < nqg; ++g)
                for (int j = 0; j < 3; ++j) border[j * ng + t * nqg + g] = t * nqg + g;
        }
        // flow pair f (complete frames f -> f+1) queries complete frame f (run_robot.py:196):
        // complete frame f is pc_trans[f] before the canonical index, the canonical cloud at it,
        // pc_trans[f-1] after it.
        qmap[t] = (t < c.cano_idx) ? t : (t == c.cano_idx ? -1 : t - 1);
        if (c.use_flow) rlen[t] = ref_off[t + 1] - ref_off[t];
    }
    if (t == 0) {
        // `iter` is caller state: it is NOT reset here, so a resumed run continues its schedule
        const long it = (long)iter[0];
        tau[0] = c.fixed_tau > 0.f ? c.fixed_tau : reart_tau_schedule(it + 1, c.n_iter, c.end_tau, c.start_tau);
        bias_corr[0] = 1.0 - pow((double)c.beta1, (double)(it + 1));
        bias_corr[1] = sqrt(1.0 - pow((double)c.beta2, (double)(it + 1)));
    }
}

// ragged reference sets -> +INF padded SoA [B][3][Mpad]
__global__ __launch_bounds__(256) void ref_soa_kernel(const float *__restrict__ ref_loc,
                                                      const int *__restrict__ ref_off, int Mpad,
                                                      float *__restrict__ soa) {
    const int f = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Mpad) return;
    const int o = ref_off[f], m = ref_off[f + 1] - o;
    float x = INFINITY, y = INFINITY, z = INFINITY;
    if (j < m) {
        const float *p = ref_loc + 3 * ((size_t)o + j);
        x = p[0]; y = p[1]; z = p[2];
    }
    float *d = soa + (size_t)f * 3 * Mpad;
    d[j] = x; d[Mpad + j] = y; d[2 * (size_t)Mpad + j] = z;
}

extern "C" int reart_relax_prepare(const reart_relax_config *cfg, const reart_relax_buffers *bufs,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    StepPlan p;
    if (!cfg || !bufs) return REART_ERR_INVALID_ARG;
    int rc = step_plan(cfg, &p);
    if (rc != REART_OK) return rc;
    if (!workspace || workspace_bytes < p.total) return REART_ERR_INVALID_ARG;
    if (!bufs->cano || !bufs->pc_list || !bufs->iter || !bufs->tau) return REART_ERR_INVALID_ARG;
    if (cfg->use_flow && (!bufs->ref_loc || !bufs->ref_flow || !bufs->ref_off)) return REART_ERR_INVALID_ARG;
    if (cfg->B > 1024) return REART_ERR_UNSUPPORTED;
    char *ws = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    SoaArgs sa = {};
    for (int j = 0; j < 2; ++j) {
        sa.job[j].src = bufs->pc_list; sa.job[j].len = nullptr;
        sa.job[j].dst = (float *)(ws + p.o_ysoa); sa.job[j].P = cfg->N; sa.job[j].Ppad = p.Npad;
    }
    rc = reart_soa_launch(sa, p.Npad, cfg->B, 1, st);
    if (rc != REART_OK) return rc;
    hipLaunchKernelGGL(relax_init_kernel, dim3(1), dim3(1024), 0, st, *cfg, bufs->ref_off,
                       (int *)(ws + p.o_rlen), (int *)(ws + p.o_qmap), bufs->iter, bufs->tau, bufs->pc_list,
                       (int *)(ws + p.o_fx), (double *)(ws + p.o_bc),
                       (p.pruned && cfg->use_flow) ? (int *)(ws + p.o_seed3) : nullptr, (int *)(ws + p.o_border));
    if (hipMemsetAsync(ws + p.o_ticket, 0, 4 * sizeof(unsigned int), st) != hipSuccess) return REART_ERR_LAUNCH;
    if (hipMemsetAsync(ws + p.o_prof_acc, 0, 4 * sizeof(unsigned long long), st) != hipSuccess) return REART_ERR_LAUNCH;
    if (p.pruned) {  // warm start of the first Chamfer search: index 0 (any valid index); the K = 1 records are the seeds
        if (hipMemsetAsync(ws + p.o_pi0, 0, sizeof(int) * (size_t)cfg->B * cfg->N, st) != hipSuccess) return REART_ERR_LAUNCH;
        if (hipMemsetAsync(ws + p.o_pi1, 0, sizeof(int) * (size_t)cfg->B * cfg->N, st) != hipSuccess) return REART_ERR_LAUNCH;
    }
    rc = reart_boxes_launch((const float *)(ws + p.o_ysoa), cfg->B, p.Npad, (float *)(ws + p.o_boxY), st);
    if (rc != REART_OK) return rc;
    if (cfg->use_grid) {
        GridBuildArgs gb = {};
        reart_grid_layout(ws + p.o_gridY, cfg->B, p.gstrideY, &gb);
        gb.pts = bufs->pc_list; gb.offsets = nullptr; gb.N = cfg->N;
        rc = reart_grid_build_launch(gb, cfg->B, st);
        if (rc != REART_OK) return rc;
        if (cfg->use_flow) {
            GridBuildArgs gr = {};
            reart_grid_layout(ws + p.o_gridR, cfg->B, p.gstrideR, &gr);
            gr.pts = bufs->ref_loc; gr.offsets = bufs->ref_off; gr.N = cfg->M_max;
            rc = reart_grid_build_launch(gr, cfg->B, st);
            if (rc != REART_OK) return rc;
        }
    }
    if (cfg->use_flow)
    {
        hipLaunchKernelGGL(ref_soa_kernel, dim3(reart_div_up(p.Mpad, 256), cfg->B), dim3(256), 0, st,
                           bufs->ref_loc, bufs->ref_off, p.Mpad, (float *)(ws + p.o_rsoa));
        rc = reart_boxes_launch((const float *)(ws + p.o_rsoa), cfg->B, p.Mpad, (float *)(ws + p.o_boxR), st);
        if (rc != REART_OK) return rc;
    }
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// ------------------------------------------------------------------------------ flow blend
// merge the S3 partial top-3 lists, blend (utils/flow_utils.py:160-167), flow-loss term and
// d loss / d pred_flow (networks/loss.py:10-21, x lambda_flow) for pair f = blockIdx.y
struct FlowArgs {
    const float *pd; const int *pi;      // [S3][B][N][3]
    const float *ref_flow; const int *ref_off; const int *qmap;
    const float *X; const float *cano;   // pc_trans [B,N,3], canonical cloud
    int N, B, S, euclidean, robust, cano_idx;
    int blocks;                          // 1: the partial lists hold 8-target BLOCKS (minimum, first index)
    const float *rsoa; int Mpad;         //    of the SoA reference image, to be rescanned here
    float smooth, lambda;
    float *gpf;                          // [B,N,3]
    double *part;                        // [B][gridDim.x]
    int *seed_out;                       // nullable [B,N,3]: the 3 neighbour indices, warm start of the next search
};

__device__ __forceinline__ const float *complete_frame(const FlowArgs &a, int f) {
    // complete_pred = cat(pc_trans[:cano], cano, pc_trans[cano:])   (run_robot.py:206)
    return f < a.cano_idx ? a.X + (size_t)f * a.N * 3
                          : (f == a.cano_idx ? a.cano : a.X + (size_t)(f - 1) * a.N * 3);
}

__device__ __forceinline__ float huber1s(float x) {
    const float ax = fabsf(x);
    return ax <= 1.0f ? 0.5f * x * x : (ax - 0.5f);
}
__device__ __forceinline__ float huber1s_grad(float x) {
    return fabsf(x) <= 1.0f ? x : (x > 0.f ? 1.0f : -1.0f);
}

template <bool ONE, int FBS>   // ONE: S <= 4 -- every partial of a query is loaded before the first compare
__device__ __forceinline__ void flow_blend_body(const FlowArgs &a, const int bx, const int f, const int nbx) {
    __shared__ double s_red[FBS / REART_WAVE];
    const int n = bx * FBS + threadIdx.x;
    double term = 0.0;
    if (n < a.N) {
        // the kernel is a chain of dependent gathers (partials -> blocks -> reference flows): every
        // stage issues all of its loads first and then reduces them without branches
        float kd[3] = {INFINITY, INFINITY, INFINITY};
        int ki[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        const size_t stride = (size_t)a.B * a.N * 3, o0 = ((size_t)f * a.N + n) * 3;
        // the point in both frames of the pair: independent of the search results, issued first
        const float *c0 = complete_frame(a, f) + 3 * (size_t)n;
        const float *c1 = complete_frame(a, f + 1) + 3 * (size_t)n;
        const float c0v[3] = {c0[0], c0[1], c0[2]}, c1v[3] = {c1[0], c1[1], c1[2]};
        for (int s0 = 0; s0 < (ONE ? 1 : a.S); s0 += 4) {
            float e[4][3];
            int q[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = s0 + u < a.S;
                const size_t o = o0 + (size_t)(ok ? s0 + u : 0) * stride;
#pragma unroll
                for (int k = 0; k < 3; ++k) { e[u][k] = a.pd[o + k]; q[u][k] = a.pi[o + k]; }
                if (!ok) { e[u][0] = INFINITY; e[u][1] = INFINITY; e[u][2] = INFINITY; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 3; ++k)   // (INF, x) never enters: INF < INF is false and ids of real entries are smaller
                    reart_top3_insert(kd, ki, e[u][k], e[u][k] < INFINITY ? q[u][k] : 0x7fffffff);
        }
        if (a.blocks) {
            // kd/ki are the 3 best blocks over all slices (by minimum, then index); the 3 nearest
            // targets lie inside them: one exact rescan of 24 targets with the full (d, index) key
            const float qx = c0v[0], qy = c0v[1], qz = c0v[2];
            const float *tx = a.rsoa + (size_t)f * 3 * a.Mpad, *ty = tx + a.Mpad, *tz = ty + a.Mpad;
            int bb[3];
            float4 X[3][2], Y[3][2], Z[3][2];
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                bb[cb] = (kd[cb] < INFINITY) ? ki[cb] : -1;
                const int blk = bb[cb] < 0 ? 0 : bb[cb];          // blocks are 8-aligned: two 16-byte loads per axis
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    X[cb][h] = *(const float4 *)(tx + blk + 4 * h);
                    Y[cb][h] = *(const float4 *)(ty + blk + 4 * h);
                    Z[cb][h] = *(const float4 *)(tz + blk + 4 * h);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { kd[k] = INFINITY; ki[k] = 0x7fffffff; }
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 vx = X[cb][u >> 2], vy = Y[cb][u >> 2], vz = Z[cb][u >> 2];
                    const int c = u & 3;
                    const float px = c == 0 ? vx.x : (c == 1 ? vx.y : (c == 2 ? vx.z : vx.w));
                    const float py = c == 0 ? vy.x : (c == 1 ? vy.y : (c == 2 ? vy.z : vy.w));
                    const float pz = c == 0 ? vz.x : (c == 1 ? vz.y : (c == 2 ? vz.z : vz.w));
                    float d = reart_sqdist3(qx, qy, qz, px, py, pz);
                    if (bb[cb] < 0) d = INFINITY;
                    reart_top3_insert(kd, ki, d, d < INFINITY ? bb[cb] + u : 0x7fffffff);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) ki[k] = ki[k] == 0x7fffffff ? 0 : ki[k];   // fewer than 3 finite candidates
        if (a.seed_out) {
            int *so = a.seed_out + 3 * ((size_t)f * a.N + n);
            so[0] = ki[0]; so[1] = ki[1]; so[2] = ki[2];
        }
        const float *rf = a.ref_flow + 3 * (size_t)a.ref_off[f];
        float rfl[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) rfl[k][c] = rf[3 * (size_t)ki[k] + c];
        float w[3], wsum = 0.f, dmin = INFINITY, fmx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float d = a.euclidean ? sqrtf(kd[k]) : kd[k];
            if (d < 1e-10f) d = 1e-10f;
            w[k] = 1.0f / d;
            wsum += w[k];
            dmin = fminf(dmin, d);
            fmx = fmaxf(fmx, (rfl[k][0] * rfl[k][0] + rfl[k][1] * rfl[k][1]) + rfl[k][2] * rfl[k][2]);
        }
        float gtf[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float wn = w[k] / wsum;
            gtf[0] += rfl[k][0] * wn; gtf[1] += rfl[k][1] * wn; gtf[2] += rfl[k][2] * wn;
        }
        const bool m = (dmin <= fmx) || (dmin <= 0.05f);
        float fl = 0.f, sm = 0.f;
        float *go = a.gpf + 3 * ((size_t)f * a.N + n);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p = c1v[c] - c0v[c];  // pred_flow (run_robot.py:207)
            const float d = p - gtf[c];
            fl += a.robust ? huber1s(d) : d * d;
            sm += p * p;
            const float gf = a.robust ? huber1s_grad(d) : 2.0f * d;
            go[c] = a.lambda * (m ? gf : a.smooth * (2.0f * p));
        }
        term = m ? (double)fl : (double)(a.smooth * sm);
    }
    term = reart_wave_sum_d(term);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < FBS / REART_WAVE; ++w) t += s_red[w];
        a.part[(size_t)f * nbx + bx] = t;
    }
}

// ------------------------------------------------------------------------------ Chamfer grad
// One workgroup per frame b: merge the slice partials of both directions, per-frame recon
// loss (networks/loss.py:27-28), and G[b,i] = d(recon + lambda*flow)/d pc_trans[b,i]:
//   2 (x_i - y_nn(i))  +  sum_{j: nn_yx(j) = i} 2 (x_i - y_j)  +  flow terms.
// The scatter-add of the reference's knn_points_backward (utils/chamfer.py:206-208) becomes, per
// target x_i, 2 (c_i x_i - sum_j y_j) with the sum of the observed points y_j accumulated in
// 64-bit FIXED POINT by integer atomics: exact to 2^-scale, independent of the order in which
// the sources arrive -> deterministic without sorting, O(N) even when every source picks the same
// target.  (The stand-alone reart_knn_points_backward keeps the sorted, bit-reproducing gather.)
// float -> 64-bit fixed point with `sbits` fractional bits, by integer shifts of the mantissa
// (gfx950 has no f64 -> i64 convert; llrint() is a long emulation).  Truncates toward zero below
// 2^-sbits: deterministic, error < 2^-sbits per term.
__device__ __forceinline__ long long fixed_from_float(float v, int sbits) {
    const unsigned bits = __float_as_uint(v);
    const int ex = (int)((bits >> 23) & 0xffu);
    const long long mant = (long long)((bits & 0x7fffffu) | (ex ? 0x800000u : 0u));
    const int sh = (ex ? ex : 1) - 150 + sbits;
    const long long mag = sh >= 0 ? (sh < 40 ? (mant << sh) : 0x7fffffffffffffffll) : (sh > -64 ? (mant >> (-sh)) : 0ll);
    return (bits >> 31) ? -mag : mag;
}

struct CGradArgs {
    const float *X, *Y;                  // pc_trans, pc_list [B,N,3]
    const float *pd0; const int *pi0;    // x -> y partials [S][B][N]
    const float *pd1; const int *pi1;    // y -> x partials
    const int *fx_bits;                  // device scalar: fractional bits of the fixed-point sums
    int N, B, S0, S1;                    // slices of the x->y / y->x partial lists
    float *G;                            // [B,N,3]: d recon / d pc_trans (both directions)
    double *loss_part;                   // [B][gridDim.x]
    int *seed0, *seed1;                  // nullable [B,N]: the neighbour indices, warm start of the next search
};
// Workgroup (r, b) owns the targets x_j, j in [r*CG_RANGE, (r+1)*CG_RANGE), of frame b:
//   1. it walks ALL observed points y_i of the frame (merging their slice partials), and adds the
//      difference (x_j - y_i) of those that chose one of its targets to that target's 64-bit
//      fixed-point sum in LDS (integer atomics: exact, order-independent -> deterministic);
//   2. for its own range it merges the x -> y partials, and writes the complete gradient
//      G[b,j] = 2 (x_j - y_nn(j)) + 2 sum_{i: nn(i) = j} (x_j - y_i), the loss terms and the seeds.
// No global atomics, no accumulator buffer to clear.
// Smallest (distance, index) key over the S slice partials of one query.  All loads of a batch of
// four slices are issued before the first compare, and the compare is branch-free: the partial lists
// come straight from the search kernel, so this is a chain of L2 round trips if written naively.
template <bool ONE>   // ONE: S <= 4, a single batch (no loop: independent merges interleave)
__device__ __forceinline__ void merge_slices(const float *__restrict__ pd, const int *__restrict__ pi, int S,
                                             size_t stride, size_t o0, int nmax, float &d, int &j) {
    d = INFINITY; j = 0x7fffffff;
    for (int s0 = 0; s0 < (ONE ? 1 : S); s0 += 4) {
        float e[4];
        int q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = o0 + (size_t)(s0 + u < S ? s0 + u : S - 1) * stride;   // duplicates change nothing
            e[u] = pd[o]; q[u] = pi[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool l = (e[u] < d) | ((e[u] == d) & (q[u] < j));
            d = l ? e[u] : d; j = l ? q[u] : j;
        }
    }
    j = j < 0 ? 0 : (j >= nmax ? nmax - 1 : j);   // in range whatever the inputs (NaN clouds)
}

template <bool ONE>
__device__ __forceinline__ void chamfer_grad_body(const CGradArgs &a, const int bx, const int b, const int nbx) {
    __shared__ unsigned long long s_acc[CG_RANGE * 3];
    __shared__ double s_red[CG_BS / REART_WAVE];
    const int tid = threadIdx.x, N = a.N;
    const int r0 = bx * CG_RANGE;
    const float *x = a.X + (size_t)b * N * 3, *y = a.Y + (size_t)b * N * 3;
    const size_t stride = (size_t)a.B * N, ob = (size_t)b * N;
    for (int e = tid; e < CG_RANGE * 3; e += CG_BS) s_acc[e] = 0ull;
    const int sbits = a.fx_bits[0];
    // own target: x -> y partials (loads issued before the barrier)
    const int io = r0 + tid;
    float d0 = INFINITY;
    int j0 = 0;
    if (io < N) merge_slices<ONE>(a.pd0, a.pi0, a.S0, stride, ob + io, N, d0, j0);
    // the gathers of the own target are issued now: they complete under the walk over the observed points
    float yn[3] = {0.f, 0.f, 0.f}, xo[3] = {0.f, 0.f, 0.f};
    if (io < N) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { yn[k] = y[3 * j0 + k]; xo[k] = x[3 * io + k]; }
    }
    __syncthreads();
    float d1own = 0.f;
    constexpr int IL = 4;   // observed points in flight per thread
    for (int ib = tid; ib < N; ib += IL * CG_BS) {
        float d1[IL];
        int j1[IL];
#pragma unroll
        for (int u = 0; u < IL; ++u) {
            const int i = ib + u * CG_BS;
            merge_slices<ONE>(a.pd1, a.pi1, a.S1, stride, ob + (i < N ? i : N - 1), N, d1[u], j1[u]);   // no branch: 4 merges in flight
        }
        float df[IL][3];
#pragma unroll
        for (int u = 0; u < IL; ++u) {
            const int i = ib + u * CG_BS;
            const int ic = i < N ? i : N - 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) df[u][k] = x[3 * j1[u] + k] - y[3 * ic + k];
        }
#pragma unroll
        for (int u = 0; u < IL; ++u) {
            const int i = ib + u * CG_BS;
            if (i >= N) continue;
            if (i == io) {   // this thread's own point
                d1own = d1[u];
                if (a.seed1) a.seed1[ob + i] = j1[u];
            }
            const int jl = j1[u] - r0;
            if (jl >= 0 && jl < CG_RANGE) {
                // y_i chose x_{j1}: differences are small, so they are also the better-conditioned
                // quantity to accumulate
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    atomicAdd(&s_acc[3 * jl + k], (unsigned long long)fixed_from_float(df[u][k], sbits));
            }
        }
    }
    if (io < N && a.seed0) a.seed0[ob + io] = j0;
    __syncthreads();
    double term = 0.0;
    if (io < N) {
        term = (double)(d0 + d1own);  // chamfer_forward + chamfer_backward (utils/chamfer.py:119-123)
        const double inv2 = 2.0 * exp2((double)-sbits);
        float *G = a.G + 3 * (ob + io);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            G[k] = 2.0f * (xo[k] - yn[k]) + (float)((double)(long long)s_acc[3 * tid + k] * inv2);
    }
    term = reart_wave_sum_d(term);
    if ((tid & 63) == 0) s_red[tid >> 6] = term;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < CG_BS / REART_WAVE; ++w) t += s_red[w];
        a.loss_part[(size_t)b * nbx + bx] = t;
    }
}

template <bool ONE>
__global__ __launch_bounds__(CG_BS) void chamfer_grad_kernel(CGradArgs a) {
    chamfer_grad_body<ONE>(a, blockIdx.x, blockIdx.y, gridDim.x);
}
template <bool ONE>
__global__ __launch_bounds__(FLOW_BS) void flow_blend_kernel(FlowArgs a) {
    flow_blend_body<ONE, FLOW_BS>(a, blockIdx.x, blockIdx.y, gridDim.x);
}
// the same consumers for K instances in one launch (reart_relax_step_batch): instance k on the grid's plane z = k
template <bool ONE>
__global__ __launch_bounds__(CG_BS) void chamfer_grad_batch_kernel(Batched<CGradArgs> ab) {
    chamfer_grad_body<ONE>(ab.a[blockIdx.z], blockIdx.x, blockIdx.y, gridDim.x);
}
template <bool ONE>
__global__ __launch_bounds__(FLOW_BS) void flow_blend_batch_kernel(Batched<FlowArgs> ab) {
    flow_blend_body<ONE, FLOW_BS>(ab.a[blockIdx.z], blockIdx.x, blockIdx.y, gridDim.x);
}
// Both consumers of the searches in ONE launch (same reason as knn_pruned_pair_kernel): workgroups
// [0, nflow) blend the flow of (frame pair, 1024 points), the rest reduce the Chamfer gradient.
// Launch order of the next search: the items of a launch are dealt in order, so a launch that ends on its
// heaviest items has a long tail.  Every item leaves a count of the boxes it tested and scanned; one extra
// workgroup of the consumer launch sums them per frame (integer sums: deterministic) and ranks the frames,
// heaviest first.
struct OrderArgs {
    const unsigned int *cost[3];   // per job: x -> y, y -> x, flow; [groups] work counts by launch position (CURRENT order)
    int *border[3];                // per job: launch position -> (frame, query group) pair (read, then rewritten)
    int groups;                    // B * nqg pairs per job
    int per;                       // positions per XCD chunk = ceil(groups / 8): chunk x = positions [x*per, (x+1)*per);
                                   // per >= groups: one chunk (interleaved launch order: a global sort)
    // cfg.profile: reduction of the search launch's per-workgroup stamps
    const unsigned long long *prof; const unsigned int *prof_pairs; unsigned long long *prof_acc; int nprof;
};
#define ORD_MAX 4                  // pairs per thread: groups <= ORD_MAX * CG_BS, otherwise the order is left alone
// One workgroup per job: inside every XCD chunk, counting sort of the pairs by their work of the iteration that just
// ran, heaviest first.  A pair never leaves its chunk (= its XCD: the frames an L2 holds stay the same).
__device__ __forceinline__ void order_body(const OrderArgs &o, const int j) {
    __shared__ unsigned int s_hist[8 * 256];
    const int tid = threadIdx.x;
    if (!o.cost[j] || o.groups > ORD_MAX * CG_BS) return;
    for (int e = tid; e < 8 * 256; e += CG_BS) s_hist[e] = 0u;
    __syncthreads();
    int old[ORD_MAX], bucket[ORD_MAX];
#pragma unroll
    for (int u = 0; u < ORD_MAX; ++u) {
        const int k = tid + u * CG_BS;
        old[u] = 0; bucket[u] = -1;
        if (k < o.groups) {
            const unsigned int c = o.cost[j][k] >> 5;
            old[u] = o.border[j][k];
            bucket[u] = (k / o.per) * 256 + 255 - (int)(c < 255u ? c : 255u);   // heaviest pairs in the chunk's first buckets
            atomicAdd(&s_hist[bucket[u]], 1u);
        }
    }
    __syncthreads();
    // exclusive prefix of every chunk's 256 counters, starting at the chunk's first position: each chunk by 64 lanes (four
    // counters per lane + a wave scan) instead of a serial walk by one thread per chunk
    for (int c0 = (tid >> 6); c0 < 8; c0 += CG_BS / 64) {
        const int l = tid & 63;
        unsigned int v[4], sum = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = s_hist[c0 * 256 + 4 * l + k]; sum += v[k]; }
        unsigned int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int t = __shfl_up(incl, off, 64);
            if (l >= off) incl += t;
        }
        unsigned int run = (unsigned int)(c0 * o.per) + (incl - sum);
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_hist[c0 * 256 + 4 * l + k] = run; run += v[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ORD_MAX; ++u)
        if (bucket[u] >= 0) o.border[j][atomicAdd(&s_hist[bucket[u]], 1u)] = old[u];
}
// cfg.profile: duration of the search launch that just ran = last workgroup end - first workgroup start (constant-rate
// wall clock), and the distance evaluations it executed; accumulated on the device, read by reart_relax_profile
__device__ __forceinline__ void prof_body(const OrderArgs &o) {
    __shared__ unsigned long long s_lo[CG_BS / 64], s_hi[CG_BS / 64], s_sum[CG_BS / 64], s_busy[CG_BS / 64];
    const int tid = threadIdx.x;
    unsigned long long lo = ~0ull, hi = 0ull, sum = 0ull, busy = 0ull;
    for (int k = tid; k < o.nprof; k += CG_BS) {
        const unsigned long long a = o.prof[2 * (size_t)k], b = o.prof[2 * (size_t)k + 1];
        lo = a < lo ? a : lo; hi = b > hi ? b : hi; sum += o.prof_pairs[k]; busy += b - a;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64), s2 = __shfl_xor(sum, off, 64);
        const unsigned long long b2 = __shfl_xor(busy, off, 64);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; sum += s2; busy += b2;
    }
    if ((tid & 63) == 0) { s_lo[tid >> 6] = lo; s_hi[tid >> 6] = hi; s_sum[tid >> 6] = sum; s_busy[tid >> 6] = busy; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < CG_BS / 64; ++w) {
            lo = s_lo[w] < lo ? s_lo[w] : lo; hi = s_hi[w] > hi ? s_hi[w] : hi; sum += s_sum[w]; busy += s_busy[w];
        }
        o.prof_acc[0] += 1ull; o.prof_acc[1] += hi - lo; o.prof_acc[2] += sum; o.prof_acc[3] += busy;
    }
}
struct PostArgs { FlowArgs fl; CGradArgs cg; OrderArgs od; int nfx, nflow, ncx, nwork, norder; };
#ifdef REART_PHASE_CLOCK   // diagnostic build only (make stats; tools/phase_clock.py): lifetimes of one workgroup of every kind
__device__ unsigned long long g_post_ts[8];      // flow blend | Chamfer gradient | launch order | profile: (start, end) each
extern "C" int reart_debug_post_clock(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_post_ts), sizeof(g_post_ts)) == hipSuccess ? REART_OK : REART_ERR_LAUNCH;
}
#define POST_TS(k) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.y == 0) g_post_ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define POST_TS(k) do { } while (0)
#endif
template <bool ONE, bool BATCH>
__global__ __launch_bounds__(CG_BS) void post_kernel(Batched<PostArgs> ab) {
    const PostArgs &a = ab.a[BATCH ? blockIdx.y : 0];
    const int w = blockIdx.x;
    if (w < a.nflow) {
        if (threadIdx.x >= POST_FBS) return;
        if (w == 1) POST_TS(0);
        flow_blend_body<ONE, POST_FBS>(a.fl, w % a.nfx, w / a.nfx, a.nfx);
        if (w == 1) POST_TS(1);
    } else if (w < a.nwork) {
        if (w == a.nflow + 1) POST_TS(2);
        chamfer_grad_body<ONE>(a.cg, (w - a.nflow) % a.ncx, (w - a.nflow) / a.ncx, a.ncx);
        if (w == a.nflow + 1) POST_TS(3);
    } else if (w < a.nwork + a.norder) {
        if (w == a.nwork) POST_TS(4);
        order_body(a.od, w - a.nwork);
        if (w == a.nwork) POST_TS(5);
    } else {
        POST_TS(6);
        prof_body(a.od);
        POST_TS(7);
    }
}

// ------------------------------------------------------------------------------ assignment loss
// run_robot.py:181-184 for fixed assignments: loss = lambda * sum_b sum_r |x[b,src_r] - y[b,tgt_(b,r)]|^2 and its
// gradient w.r.t. pc_trans (zero for the points that are not among the sampled sources)
struct AssignArgs {
    const float *X, *Y;      // pc_trans, pc_list [B,N,3]
    const int *map;          // [B,N] index into Y[b] or -1
    int N, B;
    float lambda;
    float *G;                // [B,N,3]
    double *loss_part;       // [B][gridDim.x]
};
__device__ __forceinline__ void assign_grad_body(const AssignArgs &a, int bx, int b, int gx) {
    __shared__ double s_red[CG_BS / REART_WAVE];
    const int i = bx * CG_BS + threadIdx.x, tid = threadIdx.x;
    double term = 0.0;
    if (i < a.N) {
        const size_t o = (size_t)b * a.N + i;
        const int m = a.map[o];
        float g[3] = {0.f, 0.f, 0.f};
        if (m >= 0 && m < a.N) {
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = a.X[3 * o + k] - a.Y[3 * ((size_t)b * a.N + m) + k];
                sq += d * d;
                g[k] = a.lambda * (2.0f * d);
            }
            term = (double)sq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) a.G[3 * o + k] = g[k];
    }
    term = reart_wave_sum_d(term);
    if ((tid & 63) == 0) s_red[tid >> 6] = term;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < CG_BS / REART_WAVE; ++w) t += s_red[w];
        a.loss_part[(size_t)b * gx + bx] = t * (double)a.lambda;
    }
}
__global__ __launch_bounds__(CG_BS) void assign_grad_kernel(AssignArgs a) { assign_grad_body(a, blockIdx.x, blockIdx.y, gridDim.x); }
__global__ __launch_bounds__(CG_BS) void assign_grad_batch_kernel(Batched<AssignArgs> ab) {
    assign_grad_body(ab.a[blockIdx.z], blockIdx.x, blockIdx.y, gridDim.x);
}

// ------------------------------------------------------------------------------ the step
#define MARK(k) do { if (ev) (void)hipEventRecord(ev[k], st); } while (0)

// the four launches of the default iteration (box-pruned search, Chamfer + flow) as argument blocks: what
// reart_relax_step_batch collects from every instance before it launches each kernel once for all of them
struct StepLaunch {
    BaseFwdArgs fa; SearchArgs sa; PostArgs pa; int post_blocks;
    BaseBwdArgs ba; FinalizeAdam ad; StepBook bk; void *ws_bwd; size_t bwd_bytes;
    // the unmerged iterations (Chamfer only; assignment loss [+ flow]): separate consumers after the search
    int merged, has_search, has_fl, has_aa, has_cg, fgx, fgy, ncg, nB;
    FlowArgs fl; AssignArgs aa; CGradArgs cg;
};

static int relax_step_impl(const reart_relax_config *cfg, const reart_relax_buffers *bufs,
                           void *workspace, size_t workspace_bytes, void *stream, hipEvent_t *ev,
                           bool forward_only = false, StepLaunch *collect = nullptr) {
    StepPlan p;
    if (!cfg || !bufs) return REART_ERR_INVALID_ARG;
    int rc = step_plan(cfg, &p);
    if (rc != REART_OK) return rc;
    if (!workspace || workspace_bytes < p.total) return REART_ERR_INVALID_ARG;
    if (!bufs->W1 || !bufs->b1 || !bufs->W2 || !bufs->p6d || !bufs->pt || !bufs->adam_m || !bufs->adam_v ||
        !bufs->pc_trans || !bufs->iter || !bufs->tau)
        return REART_ERR_INVALID_ARG;
    const reart_relax_config &c = *cfg;
    char *ws = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int N = c.N, B = c.B, P = c.P, H = c.H;
    float *G = (float *)(ws + p.o_G);

    // 1. forward: seg head + Gumbel-softmax + 6D + rigid apply (networks/model.py:39-70)
    BaseFwdArgs fa = {};
    fa.cano = bufs->cano; fa.W1 = bufs->W1; fa.b1 = bufs->b1; fa.W2 = bufs->W2; fa.p6d = bufs->p6d;
    fa.pt = bufs->pt; fa.gumbel = bufs->gumbel; fa.tau_ptr = bufs->tau; fa.iter_ptr = bufs->iter;
    fa.seed = c.seed; fa.tau = 1.0f; fa.N = N; fa.P = P; fa.B = B; fa.H = H; fa.Npad = p.Npad;
    fa.out = bufs->pc_trans; fa.out_soa = (float *)(ws + p.o_xsoa); fa.seg_part = bufs->seg_part;
    fa.trans_list = bufs->trans_list; fa.yT = (float *)(ws + p.o_yT); fa.hT = (float *)(ws + p.o_hT);
    fa.hard_idx = (int *)(ws + p.o_hard);
    fa.rt_table = (float *)(ws + p.o_rt);
    fa.boxes = c.use_boxes ? (float *)(ws + p.o_boxX) : nullptr;
    fa.pts = c.tune_fwd_pts;
    MARK(0);
    if (collect) collect->fa = fa;
    else {
        rc = reart_base_forward_ex(fa, st);
        if (rc != REART_OK) return rc;
    }
    MARK(1);
    if (forward_only) return REART_OK;
    if (c.use_assign && !bufs->assign_map) return REART_ERR_INVALID_ARG;

    const int nqg = reart_div_up(N, NN_BS);
    const int *qmap = (const int *)(ws + p.o_qmap);
    // job descriptions: K = 1 x -> y (0), y -> x (1); K = 3 flow
    KnnArgs ka = {};
    ka.N = B; ka.S = p.S1; ka.K = 1; ka.euclidean = 0;
    for (int j = 0; j < 2; ++j) {
        KnnJob &kj = ka.job[j];
        kj.q = j == 0 ? bufs->pc_trans : bufs->pc_list;
        kj.tsoa = (const float *)(ws + (j == 0 ? p.o_ysoa : p.o_xsoa));
        kj.P1 = N; kj.P2 = N; kj.Ppad = p.Npad; kj.L = p.L1; kj.nqg = nqg;
        kj.pd = (float *)(ws + (j == 0 ? p.o_pd0 : p.o_pd1));
        kj.pi = (int *)(ws + (j == 0 ? p.o_pi0 : p.o_pi1));
        kj.boxes = c.use_boxes ? (const float *)(ws + (j == 0 ? p.o_boxY : p.o_boxX)) : nullptr;
        kj.seed = p.pruned ? kj.pi : nullptr;            // last iteration's record is this iteration's seed (in place)
        kj.border = (const int *)(ws + p.o_border) + (size_t)j * B * nqg;
        kj.cost = p.pruned ? (unsigned int *)(ws + p.o_cost) + (size_t)j * B * nqg : nullptr;
    }
    ka.items0 = B * nqg * p.S1; ka.items = 2 * ka.items0;
    KnnArgs k3 = {};
    if (c.use_flow) {
        k3.N = B; k3.S = p.S3; k3.K = 3; k3.euclidean = 0;
        KnnJob &kj = k3.job[0];
        kj.q = bufs->pc_trans; kj.q_alt = bufs->cano; kj.qmap = qmap;
        kj.tsoa = (const float *)(ws + p.o_rsoa); kj.tlen = (const int *)(ws + p.o_rlen);
        kj.boxes = c.use_boxes ? (const float *)(ws + p.o_boxR) : nullptr;
        kj.P1 = N; kj.P2 = c.M_max; kj.Ppad = p.Mpad; kj.L = p.Mpad / p.S3; kj.nqg = nqg;
        kj.pd = (float *)(ws + p.o_pd3); kj.pi = (int *)(ws + p.o_pi3);
        kj.seed = p.pruned ? (const int *)(ws + p.o_seed3) : nullptr;
        kj.border = (const int *)(ws + p.o_border) + 2 * (size_t)B * nqg;
        kj.cost = p.pruned ? (unsigned int *)(ws + p.o_cost) + 2 * (size_t)B * nqg : nullptr;
        k3.job[1] = kj;
        k3.items0 = B * nqg * p.S3; k3.items = k3.items0;
    }
    // consumers' argument blocks
    int S3 = p.S3, S0 = p.S1, nfp = 0;
    FlowArgs fl = {};
    const bool chamfer = !c.use_assign;
    // merged: Chamfer + flow on the pruned path -- ONE search launch, ONE consumer launch
    const bool merged = p.pruned && c.use_flow && chamfer;
    if (collect && !p.pruned) return REART_ERR_UNSUPPORTED;         // the brute-force / grid variants launch as they go
    if (collect) {
        collect->merged = merged ? 1 : 0; collect->has_search = collect->has_fl = collect->has_aa = collect->has_cg = 0;
        collect->nB = B;
    }
    // Brute-force / grid variants keep their separate launches; with an auxiliary stream from the caller their
    // flow branch runs beside the Chamfer search (fork / join).  The timed variant is always serial.
    const bool forked = !p.pruned && !ev && bufs->aux_stream && bufs->ev_fork && bufs->ev_join && c.use_flow;
    hipStream_t fst = forked ? (hipStream_t)bufs->aux_stream : st;
    if (forked) {
        if (hipEventRecord((hipEvent_t)bufs->ev_fork, st) != hipSuccess) return REART_ERR_LAUNCH;
        if (hipStreamWaitEvent(fst, (hipEvent_t)bufs->ev_fork, 0) != hipSuccess) return REART_ERR_LAUNCH;
    }

    // 2. the searches
    int search_wgs = 0;
    bool search_static_order = false;
    if (p.pruned) {
        SearchArgs sa = {};
        sa.G = B * nqg; sa.S1 = p.W1; sa.S3 = p.W3; sa.sparse = p.sparse; sa.share = c.tune_share < 0 ? 0 : (c.tune_share == 1 ? 1 : 2); sa.interleave = c.tune_xcd > 0 ? 0 : 1;
        sa.cloud_resident = c.tune_cloud > 0 ? 1 : 0; sa.cloud_slices = c.tune_cloud > 0 ? c.tune_cloud : 0;
        if (chamfer) { sa.k1[0] = ka.job[0]; sa.k1[1] = ka.job[1]; sa.n1 = 2; }
        if (c.use_flow) { sa.k3 = k3.job[0]; sa.n3 = 1; }
        if (c.profile && merged) {
            sa.prof = (unsigned long long *)(ws + p.o_prof); sa.prof_pairs = (unsigned int *)(ws + p.o_prof_pairs);
        }
        if (sa.n1 + sa.n3 > 0) {
            search_wgs = reart_search_workgroups(sa);
            search_static_order = search_wgs != reart_search_grid(sa.n1, sa.n3, sa.G);   // cloud-resident form: fixed order
            if (collect) {
                if (search_static_order) return REART_ERR_UNSUPPORTED;
                collect->sa = sa; collect->has_search = 1;
            } else {
                rc = reart_search_launch(sa, st);
                if (rc != REART_OK) return rc;
            }
        }
    } else {
        if (c.use_flow) {
            if (c.use_grid) {
                GridBuildArgs gr = {};
                reart_grid_layout(ws + p.o_gridR, B, p.gstrideR, &gr);
                GridQueryArgs gq = {};
                gq.q = bufs->pc_trans; gq.q_alt = bufs->cano; gq.qmap = qmap; gq.nq = N; gq.E = B; gq.stride = p.gstrideR;
                gq.gx = gr.gx; gq.gy = gr.gy; gq.gz = gr.gz; gq.gorig = gr.gorig; gq.cell_start = gr.cell_start;
                gq.meta = gr.meta; gq.od = (float *)(ws + p.o_pd3); gq.oi = (int *)(ws + p.o_pi3);
                rc = reart_grid_query_launch(gq, 3, fst);
                S3 = 1;
            } else {
                rc = reart_knn_launch_slices(k3, 3, fst);
            }
            if (rc != REART_OK) return rc;
        }
    }
    MARK(2);
    // 3. flow consumer (separate launch unless merged): top-3 merge / rescan, blend, flow-loss term + gradient
    if (c.use_flow) {
        fl.pd = (const float *)(ws + p.o_pd3); fl.pi = (const int *)(ws + p.o_pi3); fl.ref_flow = bufs->ref_flow;
        fl.ref_off = bufs->ref_off; fl.qmap = qmap; fl.X = bufs->pc_trans; fl.cano = bufs->cano; fl.N = N; fl.B = B;
        fl.blocks = c.use_grid ? 0 : 1; fl.rsoa = (const float *)(ws + p.o_rsoa); fl.Mpad = p.Mpad;
        fl.S = S3; fl.euclidean = c.euclidean; fl.robust = c.robust; fl.cano_idx = c.cano_idx;
        fl.smooth = c.smooth_weight; fl.lambda = c.lambda_flow;
        fl.gpf = (float *)(ws + p.o_gpf); fl.part = (double *)(ws + p.o_fpart);
        fl.seed_out = p.pruned ? (int *)(ws + p.o_seed3) : nullptr;
        // merged consumers (post_kernel, workgroups of CG_BS threads): a blend workgroup covers POST_FBS points with its first
        // POST_FBS threads (the other waves leave at once: the hardware barrier counts live waves only) -- the blend is a chain
        // of gathers, 16 waves of them on ONE compute unit queue up behind its one address unit; four times the workgroups of
        // four waves spread them over four times the compute units
        const dim3 fg(reart_div_up(N, merged ? POST_FBS : FLOW_BS), B);
        nfp = fg.x * fg.y;
        if (!merged && collect) {
            if (fl.S > 4) return REART_ERR_UNSUPPORTED;
            collect->fl = fl; collect->fgx = fg.x; collect->fgy = fg.y; collect->has_fl = 1;
        } else if (!merged) {
            if (fl.S <= 4) hipLaunchKernelGGL(flow_blend_kernel<true>, fg, dim3(FLOW_BS), 0, fst, fl);
            else hipLaunchKernelGGL(flow_blend_kernel<false>, fg, dim3(FLOW_BS), 0, fst, fl);
            REART_CHECK_LAUNCH();
        }
    }
    if (forked && hipEventRecord((hipEvent_t)bufs->ev_join, fst) != hipSuccess) return REART_ERR_LAUNCH;
    MARK(3);

    if (c.use_assign) {
        // assignment loss instead of the Chamfer loss: no search, the pairs come from the caller's assign_map
        MARK(4);
        AssignArgs aa = {};
        aa.X = bufs->pc_trans; aa.Y = bufs->pc_list; aa.map = bufs->assign_map; aa.N = N; aa.B = B;
        aa.lambda = c.lambda_assign; aa.G = G; aa.loss_part = (double *)(ws + p.o_floss);
        if (collect) { collect->aa = aa; collect->has_aa = 1; }
        else {
            hipLaunchKernelGGL(assign_grad_kernel, dim3(reart_div_up(N, CG_BS), B), dim3(CG_BS), 0, st, aa);
            REART_CHECK_LAUNCH();
        }
    } else {
        // Chamfer (utils/chamfer.py:78-94) on the brute-force paths.  pc_list never changes: with use_grid the
        // direction pc_trans -> pc_list goes through its pre-built exact grid, and only pc_list -> pc_trans (moving
        // targets) is searched by brute force; without it both directions share one brute-force launch.
        if (!p.pruned) {
            if (c.use_grid) {
                GridBuildArgs gy = {};
                reart_grid_layout(ws + p.o_gridY, B, p.gstrideY, &gy);
                GridQueryArgs gq = {};
                gq.q = bufs->pc_trans; gq.nq = N; gq.E = B; gq.stride = p.gstrideY; gq.gx = gy.gx; gq.gy = gy.gy; gq.gz = gy.gz;
                gq.gorig = gy.gorig; gq.cell_start = gy.cell_start; gq.meta = gy.meta;
                gq.od = (float *)(ws + p.o_pd0); gq.oi = (int *)(ws + p.o_pi0);
                rc = reart_grid_query_launch(gq, 1, st);
                if (rc != REART_OK) return rc;
                S0 = 1;
                ka.job[0] = ka.job[1];
                ka.items = ka.items0;
            }
            rc = reart_knn_launch_slices(ka, 1, st);
            if (rc != REART_OK) return rc;
        }
        MARK(4);
        // merge + recon loss + direct gradient term + fixed-point scatter (fully parallel)
        CGradArgs cg = {};
        cg.X = bufs->pc_trans; cg.Y = bufs->pc_list;
        cg.pd0 = (const float *)(ws + p.o_pd0); cg.pi0 = (const int *)(ws + p.o_pi0);
        cg.pd1 = (const float *)(ws + p.o_pd1); cg.pi1 = (const int *)(ws + p.o_pi1);
        cg.fx_bits = (const int *)(ws + p.o_fx);
        cg.N = N; cg.B = B; cg.S0 = S0; cg.S1 = p.S1; cg.G = G;
        cg.loss_part = (double *)(ws + p.o_floss);
        const int ncg = reart_div_up(N, CG_RANGE);
        if (merged) {
            PostArgs pa = {};
            pa.fl = fl; pa.cg = cg; pa.nfx = reart_div_up(N, POST_FBS); pa.nflow = pa.nfx * B; pa.ncx = ncg;
            pa.nwork = pa.nflow + ncg * B;
            const bool reorder = c.tune_reorder >= 0 && !search_static_order;
            for (int j = 0; j < 3; ++j) {
                const KnnJob &kj = j < 2 ? ka.job[j] : k3.job[0];
                pa.od.cost[j] = reorder ? kj.cost : nullptr;
                pa.od.border[j] = (int *)(ws + p.o_border) + (size_t)j * B * nqg;
            }
            pa.od.groups = B * nqg; pa.od.per = c.tune_xcd > 0 ? reart_div_up(B * nqg, 8) : B * nqg;
            pa.norder = reorder ? 3 : 0;
            int nblk = pa.nwork + pa.norder;
            if (c.profile) {
                pa.od.prof = (const unsigned long long *)(ws + p.o_prof);
                pa.od.prof_pairs = (const unsigned int *)(ws + p.o_prof_pairs);
                pa.od.prof_acc = (unsigned long long *)(ws + p.o_prof_acc);
                pa.od.nprof = search_wgs;
                nblk += 1;
            }
            if (collect) { collect->pa = pa; collect->post_blocks = nblk; }
            else hipLaunchKernelGGL((post_kernel<true, false>), dim3(nblk), dim3(CG_BS), 0, st, reart_batched(&pa, 1));
        } else if (collect) {
            if (cg.S0 > 4 || cg.S1 > 4) return REART_ERR_UNSUPPORTED;
            collect->cg = cg; collect->ncg = ncg; collect->has_cg = 1;
        } else if (cg.S0 <= 4 && cg.S1 <= 4) hipLaunchKernelGGL(chamfer_grad_kernel<true>, dim3(ncg, B), dim3(CG_BS), 0, st, cg);
        else hipLaunchKernelGGL(chamfer_grad_kernel<false>, dim3(ncg, B), dim3(CG_BS), 0, st, cg);
        REART_CHECK_LAUNCH();
    }
    MARK(5);

    if (forked && hipStreamWaitEvent(st, (hipEvent_t)bufs->ev_join, 0) != hipSuccess) return REART_ERR_LAUNCH;
    // 6-7. model backward; the finalize kernel also applies Adam with the reference's two
    // parameter groups (run_robot.py:146-148) to the parameter each thread just reduced
    float *grads = (float *)(ws + p.o_grads);
    float *gW1 = grads, *gb1 = gW1 + 3 * H, *gW2 = gb1 + H, *g6d = gW2 + P * H, *gt = g6d + 6 * B * P;
    BaseBwdArgs ba = {};
    ba.cano = bufs->cano; ba.W2 = bufs->W2; ba.p6d = bufs->p6d; ba.pt = bufs->pt; ba.yT = fa.yT; ba.hT = fa.hT;
    ba.hard_idx = fa.hard_idx; ba.tau_ptr = bufs->tau; ba.tau = 1.0f; ba.G = G; ba.rt_table = fa.rt_table;
    ba.cano_idx = c.cano_idx;
    ba.gpf = c.use_flow ? (const float *)(ws + p.o_gpf) : nullptr;
    ba.cpts = c.tune_bwd_pts;
    ba.N = N; ba.P = P; ba.B = B; ba.H = H; ba.gW1 = gW1; ba.gb1 = gb1; ba.gW2 = gW2; ba.g6d = g6d; ba.gt = gt;
    FinalizeAdam ad = {};
    ad.enabled = 1; ad.W1 = bufs->W1; ad.b1 = bufs->b1; ad.W2 = bufs->W2; ad.p6d = bufs->p6d; ad.pt = bufs->pt;
    ad.m = bufs->adam_m; ad.v = bufs->adam_v; ad.seg_lr = c.seg_lr; ad.trans_lr = c.trans_lr;
    ad.beta1 = c.beta1; ad.beta2 = c.beta2; ad.eps = c.eps; ad.step_ptr = bufs->iter;
    ad.weight_decay = c.weight_decay;
    ad.bias_corr = (const double *)(ws + p.o_bc);
    // loss log, iter++, next tau and bias corrections: last workgroup of the finalize kernel
    const int ncgp = reart_div_up(N, CG_RANGE);
    StepBook bk = {};
    bk.enabled = 1; bk.frame_loss = (const double *)(ws + p.o_floss); bk.n_frame_part = B * ncgp;
    bk.flow_part = (const double *)(ws + p.o_fpart); bk.n_flow_part = nfp;
    bk.iter = bufs->iter; bk.tau = bufs->tau; bk.losses = bufs->losses; bk.bias_corr = (double *)(ws + p.o_bc);
    bk.ticket = (unsigned int *)(ws + p.o_ticket);
    bk.ring = c.ring; bk.n_iter = c.n_iter; bk.lambda_flow = c.lambda_flow; bk.fixed_tau = c.fixed_tau;
    bk.end_tau = c.end_tau; bk.start_tau = c.start_tau; bk.beta1 = c.beta1; bk.beta2 = c.beta2;
    if (collect) {
        collect->ba = ba; collect->ad = ad; collect->bk = bk; collect->ws_bwd = ws + p.o_bwd; collect->bwd_bytes = p.bwd_bytes;
        return REART_OK;
    }
    rc = reart_base_backward_ex(ba, &ad, &bk, ws + p.o_bwd, p.bwd_bytes, st);
    if (rc != REART_OK) return rc;
    MARK(6);
    MARK(7);
    MARK(8);
    return REART_OK;
}

extern "C" int reart_relax_step(const reart_relax_config *cfg, const reart_relax_buffers *bufs,
                                void *workspace, size_t workspace_bytes, void *stream) {
    return relax_step_impl(cfg, bufs, workspace, workspace_bytes, stream, nullptr);
}

// K independent instances of ONE shape (same N, B, P, H, M_max and switches; poses, clouds, canonical index, seeds and
// learning rates are each instance's own) advance one iteration in the launches of a single instance: every kernel of the
// iteration runs once with K argument blocks, instance k on the grid's row k.  One instance leaves most of the chip idle
// (64-608 workgroups per launch on 256 compute units, and the iteration is a chain of five dependent launches): a sweep
// over canonical frames (README.md:60) fills it this way instead of with K streams that the hardware queues interleave
// as they please.  Each instance computes exactly what reart_relax_step computes for it.  Box-pruned search paths: the
// merged Chamfer + flow iteration, Chamfer only, and the assignment loss with or without flow (the second phase of the
// README recipe, run_robot.py:164-192); K <= 6.
extern "C" int reart_relax_step_batch(const reart_relax_config *cfgs, const reart_relax_buffers *bufs, void *const *workspaces,
                                      size_t workspace_bytes, int K, void *stream) {
    if (!cfgs || !bufs || !workspaces || K < 1 || K > REART_BATCH_MAX) return REART_ERR_INVALID_ARG;
    static_assert(sizeof(Batched<SearchArgs>) <= 3584, "kernel-argument segment");
    StepLaunch L[REART_BATCH_MAX] = {};
    for (int k = 0; k < K; ++k) {
        const int rc = relax_step_impl(&cfgs[k], &bufs[k], workspaces[k], workspace_bytes, stream, nullptr, false, &L[k]);
        if (rc != REART_OK) return rc;
        if (L[k].bwd_bytes != L[0].bwd_bytes || L[k].merged != L[0].merged || L[k].has_search != L[0].has_search ||
            L[k].has_fl != L[0].has_fl || L[k].has_aa != L[0].has_aa || L[k].has_cg != L[0].has_cg || L[k].nB != L[0].nB)
            return REART_ERR_INVALID_ARG;
        if (L[0].merged && L[k].post_blocks != L[0].post_blocks) return REART_ERR_INVALID_ARG;
        if (L[0].has_fl && (L[k].fgx != L[0].fgx || L[k].fgy != L[0].fgy)) return REART_ERR_INVALID_ARG;
        if (L[0].has_cg && L[k].ncg != L[0].ncg) return REART_ERR_INVALID_ARG;
        if (L[0].has_aa && L[k].aa.N != L[0].aa.N) return REART_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    BaseFwdArgs fa[REART_BATCH_MAX];
    SearchArgs sa[REART_BATCH_MAX];
    Batched<PostArgs> pa = {};
    BaseBwdArgs ba[REART_BATCH_MAX];
    FinalizeAdam ad[REART_BATCH_MAX];
    StepBook bk[REART_BATCH_MAX];
    void *wb[REART_BATCH_MAX];
    for (int k = 0; k < K; ++k) {
        fa[k] = L[k].fa; sa[k] = L[k].sa; pa.a[k] = L[k].pa; ba[k] = L[k].ba; ad[k] = L[k].ad; bk[k] = L[k].bk; wb[k] = L[k].ws_bwd;
        // Box slices per search workgroup: ONE instance wants three waves per workgroup (its launch is only a few wave
        // lifetimes long: shorter waves, more of them in flight); with K instances in the launch the chip is full anyway and
        // the launch is bound by the instructions it issues, of which every extra wave of a workgroup repeats the prologue and
        // the coarse rounds: measured for K = 6, aggregate it/s: 3 slices 22.7 k, 2 slices 23.8 k, 1 slice 24.5 k (K = 2: best with 2).  Same
        // results whatever the count; an explicit tune_slices is kept.
        const int auto_s = K >= 3 ? 1 : (K == 2 ? 2 : 3);      // measured grid (K x slices): profiles/r03_search_ab_runs.txt
        if (cfgs[k].tune_slices == 0) sa[k].S1 = auto_s;
        if (cfgs[k].tune_slices_flow == 0) sa[k].S3 = cfgs[k].tune_slices == 0 ? auto_s : sa[k].S1;
    }
    int rc = reart_base_forward_batch(fa, K, st);
    if (rc != REART_OK) return rc;
    if (L[0].has_search) {
        rc = reart_search_launch_batch(sa, K, st);
        if (rc != REART_OK) return rc;
    }
    if (L[0].merged) {
        if (K == 1) hipLaunchKernelGGL((post_kernel<true, false>), dim3(L[0].post_blocks), dim3(CG_BS), 0, st, pa);
        else hipLaunchKernelGGL((post_kernel<true, true>), dim3(L[0].post_blocks, K), dim3(CG_BS), 0, st, pa);
        REART_CHECK_LAUNCH();
    } else {
        // the unmerged iterations: Chamfer only (search + its consumer), assignment loss (the caller's pairs) [+ flow: its
        // search + consumer] -- the same separate consumers reart_relax_step launches, once for all K instances
        if (L[0].has_fl) {
            Batched<FlowArgs> fb = {};
            for (int k = 0; k < K; ++k) fb.a[k] = L[k].fl;
            hipLaunchKernelGGL(flow_blend_batch_kernel<true>, dim3(L[0].fgx, L[0].fgy, K), dim3(FLOW_BS), 0, st, fb);
            REART_CHECK_LAUNCH();
        }
        if (L[0].has_aa) {
            Batched<AssignArgs> ab = {};
            for (int k = 0; k < K; ++k) ab.a[k] = L[k].aa;
            hipLaunchKernelGGL(assign_grad_batch_kernel, dim3(reart_div_up(L[0].aa.N, CG_BS), L[0].nB, K), dim3(CG_BS), 0, st, ab);
            REART_CHECK_LAUNCH();
        }
        if (L[0].has_cg) {
            Batched<CGradArgs> cb = {};
            for (int k = 0; k < K; ++k) cb.a[k] = L[k].cg;
            hipLaunchKernelGGL(chamfer_grad_batch_kernel<true>, dim3(L[0].ncg, L[0].nB, K), dim3(CG_BS), 0, st, cb);
            REART_CHECK_LAUNCH();
        }
    }
    return reart_base_backward_batch(ba, ad, bk, wb, L[0].bwd_bytes, K, st);
}

// Same launch sequence with a hipEvent between phases, recorded on `stream`; synchronises the
// stream and ADDS the per-phase milliseconds to h_ms[REART_RELAX_PHASES] (host memory).
// Phases: 0 forward, 1 flow K=3 search, 2 flow blend, 3 Chamfer K=1 search, 4 Chamfer merge +
// gradient scatter, 5 model backward + Adam (2 launches), 6 unused, 7 bookkeeping (always serial).  Measurement aid for
// bench.py / profiling -- not graph-capturable.
extern "C" int reart_relax_step_timed(const reart_relax_config *cfg, const reart_relax_buffers *bufs,
                                      void *workspace, size_t workspace_bytes, void *stream,
                                      float *h_ms) {
    if (!h_ms) return REART_ERR_INVALID_ARG;
    hipEvent_t ev[REART_RELAX_PHASES + 1];
    for (int k = 0; k <= REART_RELAX_PHASES; ++k)
        if (hipEventCreate(&ev[k]) != hipSuccess) return REART_ERR_LAUNCH;
    int rc = relax_step_impl(cfg, bufs, workspace, workspace_bytes, stream, ev);
    if (rc == REART_OK) {
        if (hipEventSynchronize(ev[REART_RELAX_PHASES]) != hipSuccess) rc = REART_ERR_LAUNCH;
        for (int k = 0; k < REART_RELAX_PHASES && rc == REART_OK; ++k) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[k], ev[k + 1]) != hipSuccess) rc = REART_ERR_LAUNCH;
            h_ms[k] += ms;
        }
    }
    for (int k = 0; k <= REART_RELAX_PHASES; ++k) (void)hipEventDestroy(ev[k]);
    return rc;
}

// cfg.profile: the search launch of every iteration leaves per-workgroup wall-clock stamps, reduced on the device by
// the consumer launch (prof_body).  h_out[0] launches, [1] summed duration in SECONDS (last workgroup end - first
// workgroup start, constant-rate clock), [2] distance evaluations executed, [3] the clock rate in Hz, [4] the summed
// lifetimes of all search workgroups in seconds (divided by the launches and the workgroup slots of the chip: the
// duration a perfectly balanced launch would have).  Synchronises
// the stream; reset != 0 zeroes the accumulators afterwards.
extern "C" int reart_relax_profile(const reart_relax_config *cfg, void *workspace, size_t workspace_bytes,
                                   void *stream, double *h_out, int reset) {
    StepPlan p;
    if (!cfg || !h_out) return REART_ERR_INVALID_ARG;
    int rc = step_plan(cfg, &p);
    if (rc != REART_OK) return rc;
    if (!workspace || workspace_bytes < p.total) return REART_ERR_INVALID_ARG;
    if (!cfg->profile || !p.pruned) return REART_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long acc[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(acc, (char *)workspace + p.o_prof_acc, sizeof(acc), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return REART_ERR_LAUNCH;
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        return REART_ERR_LAUNCH;
    const double hz = 1e3 * (double)khz;
    h_out[0] = (double)acc[0]; h_out[1] = (double)acc[1] / hz; h_out[2] = (double)acc[2]; h_out[3] = hz;
    h_out[4] = (double)acc[3] / hz;
    if (reset && hipMemsetAsync((char *)workspace + p.o_prof_acc, 0, sizeof(acc), st) != hipSuccess) return REART_ERR_LAUNCH;
    return REART_OK;
}

extern "C" int reart_relax_forward(const reart_relax_config *cfg, const reart_relax_buffers *bufs,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    return relax_step_impl(cfg, bufs, workspace, workspace_bytes, stream, nullptr, true);
}


// ------------------------------------------------------------------------------ measurement aid: the launch chain's floor
// (include/reart_hip.h: reart_relax_step_floor)
__global__ void step_floor_init_kernel(int *chase, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) chase[i] = (int)(((unsigned)i * 40503u + 12345u) % (unsigned)n);     // a fixed scatter: every load lands on another line
}
__global__ void step_floor_kernel(const int *__restrict__ chase, int n, int loads, int barriers, int *sink) {
    extern __shared__ int fl_lds[];
    int idx = (int)((blockIdx.x * blockDim.x + threadIdx.x) % (unsigned)n);
    for (int l = 0; l < loads; ++l) idx = chase[idx];           // dependent: the next address is the value just loaded
    if (threadIdx.x == 0) fl_lds[0] = idx;
    for (int b_ = 0; b_ < barriers; ++b_) __syncthreads();
    if (idx < 0) sink[0] = fl_lds[0];                            // never true: keeps the chain alive
}
extern "C" int reart_relax_step_floor(const int *shape, int nk, int iters, void *workspace, size_t workspace_bytes, void *stream) {
    const int n = 16384;                                         // 64 KB of indices
    if (!shape || nk < 1 || nk > 8 || iters < 1 || !workspace || workspace_bytes < sizeof(int) * (size_t)n + 64) return REART_ERR_INVALID_ARG;
    for (int k = 0; k < nk; ++k)
        if (shape[5 * k] < 1 || shape[5 * k + 1] < 1 || shape[5 * k + 1] > 1024 || shape[5 * k + 2] < 4 || shape[5 * k + 2] > 64 * 1024 ||
            shape[5 * k + 3] < 0 || shape[5 * k + 4] < 0) return REART_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    int *chase = (int *)workspace, *sink = chase + n;
    hipLaunchKernelGGL(step_floor_init_kernel, dim3(n / 256), dim3(256), 0, st, chase, n);
    for (int it = 0; it < iters; ++it)
        for (int k = 0; k < nk; ++k)
            hipLaunchKernelGGL(step_floor_kernel, dim3(shape[5 * k]), dim3(shape[5 * k + 1]), (size_t)shape[5 * k + 2], st, chase, n,
                               shape[5 * k + 3], shape[5 * k + 4], sink);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
