// reart_amd/csrc/internal.h -- argument blocks shared between the operator files and the
// fused relaxation step (step.hip).  Not part of the C ABI.
#pragma once
#include "common.h"

struct BaseFwdArgs {
    const float *cano;
    const float *W1, *b1, *W2, *p6d, *pt;
    const float *gumbel;        // [N,P] injected noise, or NULL -> Philox(seed, *iter)
    const float *tau_ptr;       // device scalar, or NULL -> tau
    const int64_t *iter_ptr;    // device counter used as Philox offset (nullable -> 0)
    uint64_t seed;
    float tau;
    int N, P, B, H, Npad;
    float *out;                 // [B,N,3]
    float *out_soa;             // [B,3,Npad] nullable (+INF beyond N), for the K-NN kernels
    int64_t *seg_part;          // [N] nullable
    float *trans_list;          // [B,P,4,4] nullable
    float *yT;                  // [P,N] nullable
    float *hT;                  // [H,N] nullable
    int *hard_idx;              // [N] nullable
    float *rt_table;            // [B*P][12] nullable: [R|t] rows for the backward's scalar loads
    float *boxes;               // [B][Npad/64][8] nullable: AABB of every 64 output points per frame
};

struct BaseBwdArgs {
    const float *cano, *W2, *p6d, *pt;
    const float *yT, *hT;
    const int *hard_idx;
    const float *tau_ptr;
    float tau;
    const float *G;             // [B,N,3]
    const float *rt_table;      // [B*P][12] or NULL (built into the workspace)
    // fused step only: the tile load adds the flow-loss terms of the two adjacent frame pairs
    const float *gpf;           // [B,N,3] d(lambda*flow)/d pred_flow or NULL
    int cano_idx;
    int N, P, B, H;
    int nchunk;                 // ceil(N / cpts)
    int cpts;                   // points per backward workgroup: 64 or 32 (set by the launcher)
    float *partial;             // [nchunk][n_out]
    // finalize
    float *gW1, *gb1, *gW2, *g6d, *gt;
};

struct AdamSeg { float *p; const float *g; float *m; float *v; int n; float lr; };
struct AdamArgs { AdamSeg seg[8]; int nseg; float beta1, beta2, eps; const int64_t *step_ptr; int step; };

int reart_base_forward_ex(const BaseFwdArgs &a, hipStream_t st);
struct FinalizeAdam {
    int enabled;
    float *W1, *b1, *W2, *p6d, *pt;   // parameters (updated in place)
    float *m, *v;                     // moments, order W1|b1|W2|p6d|pt
    float seg_lr, trans_lr, beta1, beta2, eps;
    const int64_t *step_ptr;
    const double *bias_corr;          // device [2]: 1 - beta1^step, sqrt(1 - beta2^step) of the coming step
};

// Fused step only: end-of-iteration bookkeeping (loss log, iteration counter, next temperature and
// Adam bias corrections), done by the LAST finishing workgroup of the finalize kernel -- by then
// every other workgroup has read this iteration's counters, so no extra launch is needed.
struct StepBook {
    int enabled;
    const double *frame_loss; int n_frame_part;
    const double *flow_part; int n_flow_part;
    int64_t *iter; float *tau; float *losses; double *bias_corr;
    unsigned int *ticket;      // device counter, zero between launches
    int ring, n_iter;
    float lambda_flow, fixed_tau, end_tau, start_tau, beta1, beta2;
};
int reart_base_backward_ex(BaseBwdArgs a, const FinalizeAdam *adam, const StepBook *book, void *workspace,
                           size_t workspace_bytes, hipStream_t st);
#ifdef __HIPCC__
// cosine schedule of utils/model_utils.py:33-37 evaluated in double like the host code
__device__ __forceinline__ float reart_tau_schedule(long cur_iter, int n_iter, float end_t, float start_t) {
    const double c = cos(3.14159265358979323846 * (double)cur_iter / (double)n_iter);
    return (float)((double)end_t + ((double)start_t - (double)end_t) * (c + 1.0) * 0.5);
}
#endif
int reart_adam_ex(const AdamArgs &a, hipStream_t st);

// generic K-NN driver (knn.hip): njobs in {1,2}; job j searches q[j] ([N,P1[j],3] AoS) in t[j]
// ([N,P2[j],3] AoS); outputs dists/idx [N,P1[j],K].  Workspace: see knn_plan().
int reart_knn_run(int njobs, const float *const *q, const float *const *t,
                  const int64_t *const *lenq, const int64_t *const *lent, int N, const int *P1,
                  const int *P2, int K, int euclidean, float *const *dists, int64_t *const *idx,
                  void *workspace, size_t workspace_bytes, hipStream_t st);

// ---- K-NN internals (knn.hip) -----------------------------------------------------------
#define NN_BS 64   // threads per workgroup (one wave)
#define NN_UB 16   // targets per unrolled block; slice lengths are multiples of this
struct SoaJob {
    const float *src;
    const int64_t *len;  // nullable
    float *dst;
    int P, Ppad;
};
struct SoaArgs {
    SoaJob job[2];
};

struct KnnJob {
    const float *q;        // [N,P1,3] AoS queries
    const float *tsoa;     // [N,3,Ppad] SoA targets
    const int64_t *lenq;   // nullable, rows >= lenq[n] produce zeros
    const int64_t *lent;   // nullable, number of valid targets
    const float *q_alt;    // fused step: query cloud used where qmap[b] < 0
    const int *qmap;       // nullable per-batch query frame index into q
    const int *tlen;       // nullable per-batch target count (ragged SoA rows, stride Ppad)
    const float *boxes;    // nullable [N][Ppad/NN_BOX][8]: AABB (lo xyz, hi xyz, pad) of every NN_BOX targets
    const int *seed;       // pruned search only: [N][P1][KK] candidate neighbour indices (warm start)
    const int *border;     // pruned search only, nullable: [N * nqg] (batch, query group) pair = b * nqg + g handled at
                           // launch position k (heavy pairs first: the items of a launch are dealt in order, late
                           // heavy items make a long tail)
    unsigned int *cost;    // pruned search only, nullable: [items of this job] work done by each item (boxes tested
                           // and scanned, launch order), input of the next launch's order
    int P1, P2, Ppad, L;   // Ppad = S*L, L % NN_UB == 0
    int nqg;               // ceil(P1/64)
    float *pd;             // partial dists [S][N][P1][KK]   (S > 1)
    int *pi;               // partial idx   [S][N][P1][KK]
    float *dists;          // final [N,P1,K]
    int64_t *idx;          // final [N,P1,K]
};
struct KnnArgs {
    KnnJob job[2];
    int N, S, K, euclidean;
    int items0;            // work items belonging to job 0
    int items;             // total work items
    int sparse;            // prune.hip: boxes needed by <= sparse queries of a wave take the sparse scan (set by the launcher)
};

int reart_knn_launch_slices(const KnnArgs &a, int KK, hipStream_t st);
int reart_soa_launch(const SoaArgs &sa, int maxPpad, int N, int njobs, hipStream_t st);
int reart_knn_pick_split(long waves, int P2, int K);
#ifndef NN_BOX
#define NN_BOX 16   // targets per bounding box of the block-skip test (16, 32 or 64; measured 4545 / 4438 / 4321 it/s)
#endif
int reart_boxes_launch(const float *soa, int N, int Ppad, float *boxes, hipStream_t st);
// exact search with box pruning + warm start (prune.hip); partial lists only, boxes dealt round-robin to slices
int reart_knn_launch_pruned(const KnnArgs &a, int KK, hipStream_t st);
int reart_prune_pick_split(void);
int reart_prune_pick_split3(void);   // slices of the K = 3 (flow) search
// K = 1 (two jobs) and K = 3 (one job) pruned searches in one launch
// counters: 2 zero-initialised uints for the persistent form (NULL: one workgroup per item)
int reart_knn_launch_pruned_pair(const KnnArgs &k1, const KnnArgs &k3, unsigned int *counters, hipStream_t st);
// exact pruned search, one wave = 16 queries x 4 box slots (quad.hip); a.S must be 1 and
// items0 = N * ceil(P1 / 16) per job
int reart_knn_launch_quad(const KnnArgs &a, int KK, hipStream_t st);
int reart_knn_launch_quad_pair(const KnnArgs &k1, const KnnArgs &k3, hipStream_t st);
// exact search with per-query candidate lists, target cloud staged in LDS (lane.hip); a.S must be 1
int reart_knn_launch_lane(const KnnArgs &a, int KK, hipStream_t st);

// ---- exact grid search over static target sets (grid.hip) -----------------------------------
struct GridBuildArgs {
    const float *pts;         // AoS points; set e starts at pts + 3 * off(e)
    const int *offsets;       // [E+1] prefix offsets (ragged) or NULL -> e * N
    int N;                    // points per set when offsets == NULL; max set size otherwise
    int stride;               // row stride of the sorted SoA arrays (>= max set size)
    float *gx, *gy, *gz;      // [E][stride] coordinates sorted by cell
    int *gorig;               // [E][stride] original index of each sorted point
    int *cell_start;          // [E][GR_CELLS + 1]
    float *meta;              // [E][GR_META]
    int *scratch;             // [E][3 * stride] ints: cell id | sort ping | pong
};

struct GridQueryArgs {
    const float *q;           // [E][nq][3] AoS queries
    const float *q_alt;       // used where qmap[e] < 0
    const int *qmap;          // nullable per-set query frame index
    int nq, E, stride, euclid_unused;
    const float *gx, *gy, *gz;
    const int *gorig, *cell_start;
    const float *meta;
    float *od;                // [E][nq][KK] squared distances, ascending
    int *oi;                  // [E][nq][KK] original indices
};

size_t reart_grid_bytes(int E, int stride);
void reart_grid_layout(void *mem, int E, int stride, GridBuildArgs *b);
int reart_grid_build_launch(const GridBuildArgs &b, int E, hipStream_t st);
int reart_grid_query_launch(const GridQueryArgs &q, int K, hipStream_t st);
