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
    int pts;                    // points per forward workgroup: 64 or 32 (0: the default, 32)
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
    int cpts;                   // points per backward workgroup: 64, 32 or 16 (0: the default, 32)
    float *partial;             // [nchunk][n_out]
    // finalize
    float *gW1, *gb1, *gW2, *g6d, *gt;
};

struct AdamSeg { float *p; const float *g; float *m; float *v; int n; float lr; };
struct AdamArgs { AdamSeg seg[8]; int nseg; float beta1, beta2, eps; const int64_t *step_ptr; int step; };

// Several independent instances of one shape in ONE launch (relax batch): the kernels of the fused step take their
// argument block K times and pick theirs by blockIdx.y.  6 blocks of the largest (SearchArgs) stay within the 4 KB
// kernel-argument segment with room for the runtime's hidden arguments.
#define REART_BATCH_MAX 6
template <class A>
struct Batched { A a[REART_BATCH_MAX]; };
template <class A>
static inline Batched<A> reart_batched(const A *a, int K) {
    Batched<A> b = {};
    for (int k = 0; k < K && k < REART_BATCH_MAX; ++k) b.a[k] = a[k];
    return b;
}
int reart_base_forward_ex(const BaseFwdArgs &a, hipStream_t st);
int reart_base_forward_batch(const BaseFwdArgs *a, int K, hipStream_t st);          // same shapes, K <= REART_BATCH_MAX
struct FinalizeAdam {
    int enabled;
    float *W1, *b1, *W2, *p6d, *pt;   // parameters (updated in place)
    float *m, *v;                     // moments, order W1|b1|W2|p6d|pt
    float seg_lr, trans_lr, beta1, beta2, eps;
    float weight_decay;               // torch.optim.Adam's L2 form: g += weight_decay * p
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
int reart_base_backward_batch(const BaseBwdArgs *args, const FinalizeAdam *adam, const StepBook *book, void *const *workspaces,
                              size_t workspace_bytes, int K, hipStream_t st);
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
    const int *seed;       // pruned search only: [N][P1][KK] candidate neighbour indices (warm start); K = 1: may alias pi
    const int *border;     // pruned search only, nullable: [N * nqg] (batch, query group) pair = b * nqg + g handled at
                           // launch position k.  Position k runs on XCD k / ceil(G/8); inside an XCD's chunk heavy pairs
                           // come first (the items of a launch are dealt in order, late heavy items make a long tail)
    unsigned int *cost;    // pruned search only, nullable: [N * nqg] work done for the pair at each launch position
                           // (boxes tested and scanned, all waves), input of the next launch's order
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
};

int reart_knn_launch_slices(const KnnArgs &a, int KK, hipStream_t st);
int reart_soa_launch(const SoaArgs &sa, int maxPpad, int N, int njobs, hipStream_t st);
int reart_knn_pick_split(long waves, int P2, int K);
#ifndef NN_BOX
#define NN_BOX 16   // targets per bounding box of the block-skip test (16, 32 or 64; measured 4545 / 4438 / 4321 it/s)
#endif
int reart_boxes_launch(const float *soa, int N, int Ppad, float *boxes, hipStream_t st);
// exact search with box pruning + warm start (prune.hip): ONE launch for up to two K = 1 jobs (the Chamfer
// directions) and one K = 3 job (the flow search); one workgroup per (job, batch, query group), S waves each
struct SearchArgs {
    KnnJob k1[2]; int n1;          // K = 1 jobs (0, 1 or 2); results pd / pi [B][P1] (pi int32: also the next seed)
    KnnJob k3;    int n3;          // K = 3 job (0 or 1); results pd / pi [B][P1][3]: the three best 8-target BLOCKS
                                   // (block minimum, first index), rescanned with the exact key by the consumer
    int G;                         // (batch, query group) pairs per job = B * nqg (the same for every job)
    int per;                       // pairs per XCD chunk = ceil(G / 8) (set by the launcher)
    int S1, S3;                    // waves per workgroup of a K = 1 / K = 3 item (1..4)
    int sparse;                    // boxes needed by <= sparse queries of a wave go through the (query, box) queue (0: dense only)
    int share;                     // 1: every lane's bound also takes the seeds of the 15 other lanes of its row (neighbour seeds)
    int interleave;                // 0: XCD x runs the positions [x*per, (x+1)*per) (a run of frames per L2); 1: positions
                                   // x, x+8, ... (every XCD sees every frame: balanced whatever the frames cost)
    int cloud_resident;            // 1: when the target clouds fit in LDS, one workgroup of 16 waves per (job, batch, 16
                                   // query groups) with the cloud copied into LDS (knn_cloud_kernel); S1 / S3 then unused
    int cloud_slices;              // cloud-resident form: box slices per query group, 1 | 2 | 4 | 8 (0: the default, 4)
    int k_nqg;                     // query groups per batch (set by the launcher)
    unsigned long long *prof;      // nullable: [grid][2] wall-clock stamps of every workgroup (start, end)
    unsigned int *prof_pairs;      //           [grid] distance evaluations executed by the workgroup
};
int reart_search_launch(const SearchArgs &a, hipStream_t st);
int reart_search_launch_batch(const SearchArgs *a, int K, hipStream_t st);          // group form only, same shapes
int reart_search_grid(int n1, int n3, int G);   // workgroups of that launch (an upper bound for both forms)
int reart_search_grid_cloud(int n1, int n3, int G, int nqg, int slices);
int reart_search_workgroups(const SearchArgs &a);   // of the form reart_search_launch will pick for `a`

#ifdef __HIPCC__
// branch-free insertion of key (d, j) into an ascending top-3 list ordered by (distance, index)
__device__ __forceinline__ void reart_top3_insert(float (&kd)[3], int (&ki)[3], float d, int j) {
    const bool l0 = (d < kd[0]) | ((d == kd[0]) & (j < ki[0]));
    const bool l1 = (d < kd[1]) | ((d == kd[1]) & (j < ki[1]));
    const bool l2 = (d < kd[2]) | ((d == kd[2]) & (j < ki[2]));
    kd[2] = l1 ? kd[1] : (l2 ? d : kd[2]);
    ki[2] = l1 ? ki[1] : (l2 ? j : ki[2]);
    kd[1] = l0 ? kd[0] : (l1 ? d : kd[1]);
    ki[1] = l0 ? ki[0] : (l1 ? j : ki[1]);
    kd[0] = l0 ? d : kd[0];
    ki[0] = l0 ? j : ki[0];
}
#endif

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
