/*
 * reart_hip.h -- C ABI of libreart_hip.so: the MI355X (gfx950) implementation of
 * reart's per-iteration point-cloud hot path.
 *
 * Conventions:
 *   - plain pointers + sizes, no torch / pybind types; every pointer is a DEVICE
 *     pointer unless the parameter name starts with `h_`;
 *   - the library never allocates, frees or retains device memory: outputs and
 *     scratch are caller-owned (sizes from the *_workspace_bytes queries);
 *   - every entry point is asynchronous on the hipStream_t passed as `stream`
 *     (void* so that this header needs no HIP headers), re-entrant, does no host
 *     read of device data and is therefore hipGraph-capture safe;
 *   - return value: REART_OK (0) or a negative reart_status; never exit(), never
 *     throws.  (The reference's wrappers return 1 and exit(-1) on launch failure,
 *     networks/pointnet_lib/src/ball_query.cpp:25, ball_query_gpu.cu:62-66.)
 *   - fp32 data, int64 indices at the chamferdist / knn_cuda boundaries, int32 at
 *     the pointnet2_cuda boundary, exactly as the reference interfaces.
 *
 * Each entry point names the reference interface (file:line under the upstream
 * stevenlsw/reart tree) it replaces.
 */
#ifndef REART_HIP_H
#define REART_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    REART_OK = 0,
    REART_ERR_INVALID_ARG = -1,   /* null pointer, negative size, bad enum      */
    REART_ERR_UNSUPPORTED = -2,   /* e.g. D != 3, K > REART_MAX_K                */
    REART_ERR_LAUNCH = -3,        /* hipGetLastError() after a launch            */
    REART_ERR_NO_DEVICE = -4      /* no HIP device visible                       */
} reart_status;

#define REART_MAX_K 16            /* neighbours kept in registers per query      */

/* Library / device probes (host side, no device work). */
int reart_version(void);                       /* 100*major + minor              */
int reart_device_count(void);                  /* 0 when no GPU is visible       */
const char *reart_status_string(int status);

/* ------------------------------------------------------------------------ */
/* K-nearest neighbours / Chamfer                                            */
/* ------------------------------------------------------------------------ */

/* Replaces chamferdist._C.knn_points_idx(p1,p2,lengths1,lengths2,K,version)
 * (utils/chamfer.py:174; contract in the docstring :145-171).
 *   p1 [N,P1,3], p2 [N,P2,3] f32 contiguous; lengths1/2 [N] i64 or NULL (= full);
 *   dists [N,P1,K] f32 squared L2, idx [N,P1,K] i64 into p2, ascending by
 *   (distance, index); rows >= lengths1[n] and slots >= lengths2[n] are zero.
 * Distance contract: ((dx*dx)+(dy*dy))+(dz*dz) in fp32, no FMA, ties -> lowest j.
 * workspace (SoA target image + per-slice partial results):
 *   reart_knn_points_workspace_bytes(N,P1,P2,K) bytes. */
size_t reart_knn_points_workspace_bytes(int N, int P1, int P2, int K);
int reart_knn_points_idx(const float *p1, const float *p2,
                         const int64_t *lengths1, const int64_t *lengths2,
                         int N, int P1, int P2, int D, int K,
                         float *dists, int64_t *idx,
                         void *workspace, size_t workspace_bytes, void *stream);

/* Replaces chamferdist._C.knn_points_backward(p1,p2,lengths1,lengths2,idx,grad_dists)
 * (utils/chamfer.py:206-208).  grad_p1 [N,P1,3], grad_p2 [N,P2,3] are fully
 * written (zero where nothing contributes).  Deterministic: the scatter into
 * grad_p2 is a per-target gather over a counting sort of idx, summed in
 * ascending (i,k) order -- no float atomics.
 *   workspace: reart_knn_points_backward_workspace_bytes(N,P1,P2,K) bytes. */
size_t reart_knn_points_backward_workspace_bytes(int N, int P1, int P2, int K);
int reart_knn_points_backward(const float *p1, const float *p2,
                              const int64_t *lengths1, const int64_t *lengths2,
                              const int64_t *idx, const float *grad_dists,
                              int N, int P1, int P2, int D, int K,
                              float *grad_p1, float *grad_p2,
                              void *workspace, size_t workspace_bytes, void *stream);

/* Both directions of ChamferDistance.forward(bidirectional=True)
 * (utils/chamfer.py:78-123) in one launch: x,y [N,P,3];
 *   d_xy/i_xy: NN of each x point in y; d_yx/i_yx: NN of each y point in x. */
size_t reart_chamfer_bidir_workspace_bytes(int N, int P);
int reart_chamfer_bidir(const float *x, const float *y, int N, int P,
                        float *d_xy, int64_t *i_xy, float *d_yx, int64_t *i_yx,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Replaces knn_cuda.KNN(k, transpose_mode=True).forward(ref, query)
 * (run_robot.py:65-66,122; shape contract utils/model_utils.py:42):
 *   ref [B,nr,3], query [B,nq,3] -> dist [B,nq,k] ascending, idx [B,nq,k] i64.
 * euclidean != 0: dist = sqrt(squared distance) (upstream KNN_CUDA 0.2).
 * workspace: reart_knn_points_workspace_bytes(B, nq, nr, k). */
int reart_knn_cuda(const float *ref, const float *query, int B, int nr, int nq,
                   int D, int k, int euclidean, float *dist, int64_t *idx,
                   void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ */
/* Flow loss                                                                 */
/* ------------------------------------------------------------------------ */

/* Replaces blend_anchor_motion(query_loc, reference_loc, reference_flow, knn, return_mask=True)
 * (utils/flow_utils.py:147-170) including its knn_cuda.KNN(k) call (:158):
 *   query [nq,3], ref [nr,3], ref_flow [nr,3] -> flow [nq,3], mask [nq] (uint8 0/1). */
size_t reart_blend_anchor_motion_workspace_bytes(int nq, int nr, int k);
int reart_blend_anchor_motion(const float *query, const float *ref, const float *ref_flow,
                              int nq, int nr, int k, int euclidean,
                              float *flow, uint8_t *mask,
                              void *workspace, size_t workspace_bytes, void *stream);

/* The T-1 calls of blend_anchor_motion an iteration makes (run_robot.py:194-201: one per frame pair; utils/flow_utils.py:147-170)
 * as one: query [B,nq,3], ref / ref_flow [B,nr_max,3] (every frame's reference set padded to the longest), ref_len [B] int64 the
 * true lengths (NULL: all nr_max; each >= k) -> flow [B,nq,3], mask [B,nq].  Bit for bit what B calls of
 * reart_blend_anchor_motion return (one search over the batch with per-batch target lengths, one blend launch). */
size_t reart_blend_anchor_motion_batch_workspace_bytes(int B, int nq, int nr_max, int k);
int reart_blend_anchor_motion_batch(const float *query, const float *ref, const float *ref_flow, const int64_t *ref_len,
                                    int B, int nq, int nr_max, int k, int euclidean, float *flow, uint8_t *mask,
                                    void *workspace, size_t workspace_bytes, void *stream);

/* The kinematic projection's iteration between the assignment re-solve and the FK backward, as one call (csrc/kinpost.hip;
 * run_robot.py:165-209 for `--model kinematic --use_assign_loss [--use_flow_loss]`):
 *   assignment branch (:177-184)  loss = lambda_assign x sum |pc_src - tgt[cols]|^2 over the n sampled points of every frame, its
 *                                 gradient scattered to the sampled points' places in dL/d pc_trans (slot_of_point [N]: sample slot
 *                                 of canonical point p or -1; pc_src [B,n,3] = the sampled points of pc_trans as the solve saw them;
 *                                 tgt [B,n,3]; cols [B,n] the optimum) -- n = 0: no such branch;
 *   flow branch (:194-209)        comp = pc_trans[:cano_idx] | cano | pc_trans[cano_idx:]; blend_anchor_motion of every frame pair
 *                                 (reart_blend_anchor_motion_batch: ref / ref_flow [B,nr_max,3], ref_len [B] or NULL), flow_loss of
 *                                 comp[1:] - comp[:-1] against it (robust, smooth_weight), times lambda_flow; ref = NULL: none.
 * Outputs: G [B,N,3] = dL/d pc_trans of both branches (bit for bit what the reference's tensor expressions give through autograd's
 * operator order), matched [B,n,3] (nullable) the matched targets, losses [3] (device) = the assignment term, the flow term, their
 * sum.  workspace: reart_kin_post_workspace_bytes(B, N, nr_max or 0, k).  Nine launches on `stream`, no host synchronisation. */
size_t reart_kin_post_workspace_bytes(int B, int N, int nr_max, int k);
int reart_kin_post(const float *pc_trans, const float *cano, int B, int N, int cano_idx, const float *pc_src, const float *tgt,
                   const int32_t *cols, const int32_t *slot_of_point, int n, float lambda_assign, const float *ref,
                   const float *ref_flow, const int64_t *ref_len, int nr_max, int k, int euclidean, float lambda_flow, int robust,
                   float smooth_weight, float *G, float *matched, float *losses, void *workspace, size_t workspace_bytes,
                   void *stream);

/* Replaces flow_loss(gt_flow_list, pred_flow_list, flow_mask_list, robust, smooth_weight)
 * (networks/loss.py:10-21): gt, pred [B,N,3]; mask [B,N] uint8 or NULL (= all ones);
 * loss: device float scalar; grad_pred [B,N,3] or NULL = d loss / d pred. */
size_t reart_flow_loss_workspace_bytes(void);
int reart_flow_loss(const float *gt, const float *pred, const uint8_t *mask, int B, int N,
                    int robust, float smooth_weight, float *loss, float *grad_pred,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ */
/* Relaxation model (BaseModel) and rigid transforms                         */
/* ------------------------------------------------------------------------ */

/* Replaces BaseModel.forward(cano_pc, tau=...) (networks/model.py:39-70): seg head
 * 3->H->P (networks/blocks.py:99-118; W1 [H,3], b1 [H], W2 [P,H], last layer bias-free),
 * hard Gumbel-softmax with the -log(Exp(1)) noise `gumbel` [N,P] supplied by the caller
 * (F.gumbel_softmax draws it from torch's generator), rotation_6d_to_matrix
 * (screw_se3/geo_utils.py:632-651) and the per-part rigid apply.
 *   out [B,N,3]; seg_part [N] i64 = arg-max of the noise-free logits (:70);
 *   trans_list [B,P,4,4]; saved for backward: yT [P,N], hT [H,N], hard_idx [N] i32.
 *   seg_part / trans_list / yT / hT / hard_idx may be NULL.  P <= 32. */
int reart_base_forward(const float *cano, int N, int P, int B,
                       const float *W1, const float *b1, const float *W2, int H,
                       const float *prop6d, const float *propt,
                       const float *gumbel, float tau,
                       float *out, int64_t *seg_part, float *trans_list,
                       float *yT, float *hT, int32_t *hard_idx, void *stream);

/* The noise F.gumbel_softmax(logits, tau, hard=True) adds (networks/model.py:44: -log(Exp(1)) samples from torch's
 * generator) as the fused step draws it IN the forward kernel when reart_relax_buffers.gumbel is NULL: a Philox4x32-10
 * stream keyed by `seed`, counter (point, part / 4, iteration).  out [N,P] = exactly the samples iteration `iter` of an
 * engine with that seed uses, so the production path can be checked (distribution, independence) and replayed
 * (inject `out` as `gumbel`: same bits).  */
int reart_gumbel_noise(uint64_t seed, int64_t iter, int N, int P, float *out, void *stream);

/* Backward of the above (the reference relies on autograd): G = dL/d out [B,N,3] ->
 * gradients of W1, b1, W2, proposal_6d [B,P,6], proposal_t [B,P,3].  Deterministic
 * (fixed-order chunked reductions, no atomics). */
size_t reart_base_backward_workspace_bytes(int N, int P, int B, int H);
int reart_base_backward(const float *cano, int N, int P, int B,
                        const float *W1, const float *b1, const float *W2, int H,
                        const float *prop6d, const float *propt,
                        const float *yT, const float *hT, const int32_t *hard_idx, float tau,
                        const float *G,
                        float *gW1, float *gb1, float *gW2, float *g6d, float *gt,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Replaces compute_pc_transform(cano_pc, pose_list, cano_part) (utils/model_utils.py:54-67)
 * and the hard-label apply of KinematicModel.forward (networks/model.py:161-165):
 *   cano [N,3], pose [B,P,4,4], part [N] i64 -> out [B,N,3]. */
int reart_compute_pc_transform(const float *cano, const float *pose, const int64_t *part,
                               int N, int P, int B, float *out, void *stream);

/* Replaces rotation_6d_to_matrix (screw_se3/geo_utils.py:632-651): d6 [n,6] -> R [n,3,3]. */
int reart_rotation_6d_to_matrix(const float *d6, int n, float *R, void *stream);

/* One torch.optim.Adam step on one tensor (run_robot.py:145-151,219-221; amsgrad off,
 * weight_decay 0).  `step` is the 1-based step count. */
int reart_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                    int n, int step, float lr, float beta1, float beta2, float eps,
                    void *stream);

/* The same step on up to 8 tensors in ONE launch (run_robot.py:145-151: one optimizer over theta / axis / moment, each a
 * few hundred floats): param / grad / exp_avg / exp_avg_sq are HOST arrays of `count` device pointers, n and lr host arrays
 * of their sizes and learning rates, read at the call.  Same arithmetic per element as reart_adam_step. */
int reart_adam_step_multi(int count, float *const *param, const float *const *grad, float *const *exp_avg,
                          float *const *exp_avg_sq, const int *n, const float *lr, int step, float beta1, float beta2,
                          float eps, void *stream);

/* ------------------------------------------------------------------------ */
/* Fused relaxation iteration                                                */
/* ------------------------------------------------------------------------ */

/* One iteration of the reference's optimisation loop (run_robot.py:154-221, branch
 * "Chamfer reconstruction loss [+ flow loss]", i.e. i < assign_iter): BaseModel forward
 * with in-kernel Gumbel noise, recon_loss, blend_anchor_motion x (T-1) + flow_loss,
 * backward, Adam with the reference's two parameter groups, cosine temperature schedule.
 * Everything that changes between iterations (counter, RNG offset, tau, loss log) lives in
 * device memory, so one call enqueues a fixed launch sequence: capture it once in a
 * hipGraph / torch.cuda.CUDAGraph and replay. */
typedef struct reart_relax_config {
    int N, P, B, H;          /* points, parts (<=32), frames-1, seg-head width (128)      */
    int cano_idx;            /* position of the canonical frame in the T = B+1 sequence   */
    int use_flow;            /* --use_flow_loss                                           */
    int robust;              /* --use_robust_loss (Huber, delta 1)                        */
    int euclidean;           /* knn_cuda distance convention: 1 = sqrt (upstream)         */
    int flow_k;              /* neighbours blended; 3 (run_robot.py:66)                   */
    int M_max;               /* largest flow reference set                                */
    int M_total;             /* sum of the reference set sizes                            */
    int n_iter;              /* --n_iter: period of the cosine schedule                   */
    int ring;                /* rows in the loss log                                      */
    float lambda_flow;       /* --lambda_flow                                             */
    float smooth_weight;     /* flow_loss smooth_weight, 1e-2                             */
    float trans_lr, seg_lr;  /* --trans_lr, --seg_lr                                      */
    float beta1, beta2, eps; /* Adam defaults 0.9, 0.999, 1e-8                            */
    float start_tau, end_tau;/* --start_tau, --end_tau                                    */
    float fixed_tau;         /* > 0: frozen temperature (resume, run_robot.py:96-97)      */
    uint64_t seed;           /* Philox key of the Gumbel noise                            */
    int use_grid;            /* 1: exact grid search for the static targets (pc_list, flow */
                             /*    references); 0: brute force everywhere (same results)   */
    int use_boxes;           /* 1: bounding-box block-skip test in the brute-force searches */
                             /*    (exact; pays off when the clouds are stored in Morton order) */
    int use_assign;          /* 1: the assignment loss replaces the Chamfer loss (run_robot.py:164-187, */
                             /*    i >= assign_iter): lambda_assign * sum |x_src - y_assigned|^2 over   */
                             /*    the pairs of `assign_map`                                            */
    float lambda_assign;     /* --lambda_assign                                           */
    float weight_decay;      /* --weight_decay: torch.optim.Adam's L2 form g += wd * p (run_robot.py:146-148) */
    /* Tuning and measurement switches.  0 = the library default everywhere.  The library reads NO environment
     * variable in any call path: the caller decides once per instance (reart_amd/relax.py maps REART_* variables
     * to these fields when an engine is built), so the ranks of a job cannot diverge mid-run. */
    int search_mode;         /* 0: exact box-pruned warm-started search (default); 1: cold brute force (same results) */
    int tune_slices;         /* waves (box slices) per search workgroup, 1..4 (default 3)  */
    int tune_slices_flow;    /* the same for the K = 3 flow search (default: tune_slices)  */
    int tune_sparse;         /* boxes needed by <= n queries of a wave go through the (query, box) queue instead of a  */
                             /* 64-lane scan: 1..64 (default 40), < 0: dense scans only                                */
    int tune_fwd_pts;        /* points per forward workgroup: 64 | 32 (default 32)         */
    int tune_bwd_pts;        /* points per backward workgroup: 64 | 32 | 16 (default 32)   */
    int tune_reorder;        /* < 0: keep the static launch order of the search items      */
    int tune_cloud;          /* 1 | 2 | 4 | 8: cloud-resident form of the search (16-wave workgroups that copy their target   */
                             /* cloud into LDS, that many box slices per query group; clouds up to ~7000 points).  Exact and  */
                             /* tested, but measured slower than the default (targets from L2): the launch is bound by       */
                             /* instruction issue, not by the latency the LDS copy removes                                  */
    int tune_xcd;            /* > 0: every XCD runs one contiguous eighth of the (frame, query group) pairs (a run of */
                             /* frames per L2) instead of every 8th pair; measured slower: frames differ 10x in work  */
    int profile;             /* 1: the search launch of every iteration records per-workgroup wall-clock stamps and  */
                             /*    its executed distance evaluations, reduced on the device (reart_relax_profile)    */
    int tune_share;          /* < 0: a query's search bound comes from its own seeds only; default: also from the    */
                             /*    seeds of the 15 neighbouring queries of its row (same results, fewer boxes), the  */
                             /*    candidates split over the waves of the search workgroup; 1: every wave all of them */
} reart_relax_config;

typedef struct reart_relax_buffers {
    const float *cano;       /* [N,3]                                                     */
    const float *pc_list;    /* [B,N,3] observed frames without the canonical one         */
    const float *ref_loc;    /* [M_total,3] flow reference points, sets concatenated      */
    const float *ref_flow;   /* [M_total,3] their flow vectors                            */
    const int *ref_off;      /* [B+1] int32 prefix offsets of the sets                    */
    const float *gumbel;     /* NULL (in-kernel Philox) or injected noise [N,P] (tests)   */
    float *W1, *b1, *W2;     /* seg head [H,3], [H], [P,H]; updated in place              */
    float *p6d, *pt;         /* proposal_6d [B,P,6], proposal_t [B,P,3]                   */
    float *adam_m, *adam_v;  /* [3H+H+PH+9BP] each, order W1|b1|W2|p6d|pt, start at zero   */
    int64_t *iter;           /* device scalar: completed iterations (caller initialises)  */
    float *tau;              /* device scalar: temperature of the next iteration          */
    float *losses;           /* [ring][4]: recon, lambda*flow, total, tau; row iter%ring  */
    float *pc_trans;         /* [B,N,3] forward output of the last iteration              */
    int64_t *seg_part;       /* [N] or NULL                                               */
    float *trans_list;       /* [B,P,4,4] or NULL                                         */
    /* optional fork/join: with all three set, the flow branch (K=3 search + blend) runs on      */
    /* aux_stream concurrently with the Chamfer search; the hipEvent_t's are caller-owned.       */
    void *aux_stream, *ev_fork, *ev_join;
    /* use_assign: [B,N] int32, assign_map[b][i] = index into pc_list[b] of the point assigned to     */
    /* pc_trans[b][i], or -1 when point i is not among the sampled sources (the caller refreshes it   */
    /* every assign_gap iterations from FPS + linear assignment, run_robot.py:165-178)                */
    const int *assign_map;
} reart_relax_buffers;

size_t reart_relax_workspace_bytes(const reart_relax_config *cfg);
/* once per problem: static SoA images of pc_list / reference sets, tau(iter) */
int reart_relax_prepare(const reart_relax_config *cfg, const reart_relax_buffers *bufs,
                        void *workspace, size_t workspace_bytes, void *stream);
/* forward only, for the CURRENT iteration (same temperature and Gumbel noise the next reart_relax_step
 * will use): fills pc_trans / seg_part / trans_list and changes nothing else.  The assignment loss needs
 * the transformed clouds of iteration i to compute the assignment used in iteration i. */
int reart_relax_forward(const reart_relax_config *cfg, const reart_relax_buffers *bufs,
                        void *workspace, size_t workspace_bytes, void *stream);
/* enqueue one iteration (5 launches in the default configuration, no host sync, no environment lookups) */
int reart_relax_step(const reart_relax_config *cfg, const reart_relax_buffers *bufs,
                     void *workspace, size_t workspace_bytes, void *stream);
/* K independent instances of ONE shape (same N, B, P, H, M_max and switches in every cfgs[k]; clouds, parameters,
 * canonical index, seed and learning rates are each instance's own) advance one iteration in the five launches of a
 * single instance: every kernel runs once with K argument blocks, instance k on row k of its grid.  cfgs / bufs are
 * arrays of K, workspaces[k] is instance k's workspace (each prepared with reart_relax_prepare).  Each instance
 * computes exactly what reart_relax_step computes for it.  This is how a sweep over canonical frames
 * (/root/reference/README.md:60, run_robot.py: one process per cano_idx) fills the chip: one instance occupies a fraction
 * of the 256 compute units and is a chain of dependent launches.  Box-pruned search paths: Chamfer + flow loss (five
 * launches), Chamfer only, and the assignment loss with or without flow (run_robot.py:164-192: the pairs in each
 * instance's assign_map); every instance in the same mode, 1 <= K <= 6; REART_ERR_UNSUPPORTED / REART_ERR_INVALID_ARG
 * otherwise. */
int reart_relax_step_batch(const reart_relax_config *cfgs, const reart_relax_buffers *bufs, void *const *workspaces,
                           size_t workspace_bytes, int K, void *stream);
/* measurement aid: the same sequence with hipEvents between phases on `stream`; synchronises
 * and ADDS per-phase milliseconds to the HOST array h_ms[REART_RELAX_PHASES]:
 * 0 forward, 1 flow K=3 search, 2 flow blend, 3 Chamfer K=1 search, 4 Chamfer merge + gradient,
 * 5 model backward + Adam + bookkeeping, 6-7 unused.  In the default configuration (pruned searches,
 * flow loss on) both searches share ONE launch and both consumers share ONE launch: phases 1-2 are
 * then empty, 3 = the search launch, 4 = the consumer launch.  Always serial (no fork/join); event
 * pairs around single launches over-read short kernels by a few microseconds. */
#define REART_RELAX_PHASES 8
int reart_relax_step_timed(const reart_relax_config *cfg, const reart_relax_buffers *bufs,
                           void *workspace, size_t workspace_bytes, void *stream, float *h_ms);
/* measurement aid (cfg->profile = 1, default search path with the flow loss): the search launch of every
 * reart_relax_step -- eager or replayed from a graph -- leaves per-workgroup wall-clock stamps that the consumer
 * launch reduces on the device.  h_out[0] = launches since the last reset, [1] = their summed duration in seconds
 * (first workgroup start to last workgroup end, constant-rate clock), [2] = query-target distance evaluations they
 * executed, [3] = the clock rate in Hz, [4] = the summed lifetimes of the launches' workgroups in seconds.
 * h_out holds 5 doubles.  Synchronises `stream`; reset != 0 clears the accumulators. */
int reart_relax_profile(const reart_relax_config *cfg, void *workspace, size_t workspace_bytes, void *stream,
                        double *h_out, int reset);

/* ------------------------------------------------------------------------ */
/* PointNet++ sampling / grouping (the live kernels of pointnet2_cuda)       */
/* ------------------------------------------------------------------------ */

/* Replaces furthest_point_sampling_wrapper(b,n,m, points, temp, idx)
 * (networks/pointnet_lib/src/sampling.cpp:38-49 -> sampling_gpu.cu:93-209) and the CPU
 * fallback farthest_point_sample (networks/pointnet2_utils.py:74-99).
 *   xyz [B,N,3]; start [B] i32 first index per cloud or NULL (= 0, the CUDA kernel's rule;
 *   the CPU fallback draws it from torch's RNG: inject it);  cuda_mode = 0: arg-max = first
 *   maximum (torch.max), 1: the CUDA kernel's block-tree tie rule.  No `temp` buffer is
 *   needed: the running minimum distances live in registers.
 *   idx32 [B,npoint] i32 and/or idx64 [B,npoint] i64 (either may be NULL).  N <= 12288. */
int reart_fps(const float *xyz, int B, int N, int npoint, const int32_t *start, int cuda_mode,
              int32_t *idx32, int64_t *idx64, void *stream);

/* Replaces ball_query_wrapper(b,n,m,radius,nsample,new_xyz,xyz,idx)
 * (networks/pointnet_lib/src/ball_query.cpp:15-26 -> ball_query_gpu.cu:9-45) and the CPU
 * fallback query_ball_point (networks/pointnet2_utils.py:102-140).
 *   cuda_mode = 0: d2 <= float32(radius^2), first nsample in index order, padded with the
 *   nearest point;  1: d2 < float(radius)*float(radius), padded with the first hit (0 if none).
 *   Distance: direct difference ((dx*dx)+(dy*dy))+(dz*dz).
 *   idx32 [B,S,nsample] i32 and/or idx64 [B,S,nsample] i64. */
int reart_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S,
                     double radius, int nsample, int cuda_mode,
                     int32_t *idx32, int64_t *idx64, void *stream);

/* The channel-major operators of the reference's pybind module `pointnet2_cuda` that nothing in the reference calls
 * (networks/pointnet_lib/src/pointnet2_api.cpp:14-25; only pointnet2_modules.py reaches them, and nothing imports it) --
 * here so that the module is whole (reart_amd/pointnet2_cuda.py).  int32 indices, caller-allocated outputs.
 *   reart_pn2_gather_points:       out[b][c][m] = points[b][c][idx[b][m]]; points [B,C,N], idx [B,M], out [B,C,M]
 *       (gather_points_wrapper, sampling_gpu.cu:8-24; group_points_wrapper, group_points_gpu.cu:39-54, is the same map with
 *       idx [B, npoints * nsample] and out [B,C,npoints,nsample]);
 *   reart_pn2_gather_points_grad:  grad_points[b][c][idx[b][m]] += grad_out[b][c][m] (float atomics, as sampling_gpu.cu:46-63 /
 *       group_points_gpu.cu:8-21; the caller zeroes grad_points, pointnet_lib/pointnet2_utils.py:70,231);
 *   reart_pn2_three_interpolate:   out[b][c][n] = (w0 p[i0] + w1 p[i1]) + w2 p[i2], p = points[b][c][:]; points [B,C,M],
 *       idx / weight [B,N,3], out [B,C,N] (interpolate_gpu.cu:149-169);
 *   reart_pn2_three_interpolate_grad: grad_points[b][c][i_j] += grad_out[b][c][n] * w_j (interpolate_gpu.cu:192-214). */
int reart_pn2_gather_points(const float *points, const int32_t *idx, int B, int C, int N, int M, float *out, void *stream);
int reart_pn2_gather_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int M, float *grad_points,
                                 void *stream);
int reart_pn2_three_interpolate(const float *points, const int32_t *idx, const float *weight, int B, int C, int M, int N,
                                float *out, void *stream);
int reart_pn2_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight, int B, int C, int N, int M,
                                     float *grad_points, void *stream);

/* ------------------------------------------------------------------------ */
/* Projection model: screw-joint forward kinematics                          */
/* ------------------------------------------------------------------------ */

/* Replaces fk(paths_to_base, reverse_topo, edge_index, axis_list, moment_list, theta_list,
 * distance_list) (utils/kinematic_utils.py:151-198) including
 * screw_param_to_exponential_coordinates / transform_from_exponential_coordinates
 * (screw_se3/screw_utils.py:6-30) and se3_exp_map (screw_se3/geo_utils.py:147-222).
 * The joint tree is passed as arrays instead of the reference's dicts:
 *   parent[c] (-1 = root), edge_of_part[c] (index of edge "c_parent" in edge_index),
 *   order = reverse_topo (parts from root to leaf), all i32 [P];
 *   axis, moment [E,3]; theta [B,E]; distance [B,E] or NULL (= 1e-6, :176);
 *   trans [B,P,4,4] out. */
int reart_fk_forward(const int32_t *parent, const int32_t *edge_of_part, const int32_t *order,
                     int P, const float *axis, const float *moment, const float *theta,
                     const float *distance, int B, int E, float *trans, void *stream);

/* Backward of `fk` + the hard-label rigid apply of KinematicModel.forward
 * (networks/model.py:161-165): x [N,3], part [N] i64, G = dL/d pc_trans [B,N,3], trans = the
 * forward's output -> g_axis [E,3], g_moment [E,3], g_theta [B,E], g_distance [B,E] or NULL.
 * Deterministic (ordered reductions).  P <= 64. */
size_t reart_fk_backward_workspace_bytes(int P, int B, int E);
int reart_fk_backward(const float *x, const int64_t *part, const float *G, int N,
                      const int32_t *parent, const int32_t *edge_of_part, const int32_t *order,
                      int P, const float *axis, const float *moment, const float *theta,
                      const float *distance, int B, int E, const float *trans,
                      float *g_axis, float *g_moment, float *g_theta, float *g_distance,
                      void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ */
/* PointNet++ correspondence extractor: dense layers and interpolation       */
/* ------------------------------------------------------------------------ */

/* One fused layer  Y = relu(X Wt + bias) [max-pooled over pool_k consecutive rows]  on the fp32
 * matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).  Replaces Conv2d/Conv1d(1x1) + BatchNorm
 * (eval, folded into Wt/bias by the caller) + ReLU and the max over nsample of
 * PointNetSetAbstractionMsg / PointNetSetAbstraction / PointNetFeaturePropagation
 * (networks/pointnet2_utils.py:209-235, 257-295, 309-348).
 *   Plain input: X [rows, ldx].  Gathered input (gather_idx != NULL; replaces index_points +
 *   centre subtraction + cat of :270-281 / sample_and_group_all :186-188 without materialising
 *   the grouped tensor): row r takes point gather_idx[r] (i64, [B,S,K] flattened) of cloud
 *   r / (S*K); columns = [F (D) | Q - C[r / K]] (xyz_first = 0) or [Q | F] (xyz_first = 1,
 *   C may be NULL); F [B*Npts, D], Q [B*Npts, 3], C [B*S, 3]; Cin must equal D + 3.
 *   Wt [Cin, Cout] (transposed conv weight), bias [Cout] or NULL; pool_k in {0, 32, 64, 128};
 *   Y [rows (/pool_k), ldy], written at columns ycol0 .. ycol0 + Cout. */
int reart_mlp_layer(const float *X, int ldx, const int64_t *gather_idx, int K, int S, int Npts,
                    const float *F, int D, const float *Q, const float *C, int xyz_first,
                    const float *Wt, const float *bias, int rows, int Cin, int Cout, int relu,
                    int pool_k, float *Y, int ldy, int ycol0, void *stream);

/* Three such layers in ONE launch for a scale of the first set-abstraction level (PointNetSetAbstractionMsg,
 * networks/pointnet2_utils.py:257-295): gathered input [F (3) | Q - C] (6 columns) -> C1 -> C2 -> C3 (each with bias and
 * ReLU) -> max over the K rows of a group; the activations between the layers never leave the compute unit (LDS).
 * Same arguments as the gathered form of reart_mlp_layer (D = 3, xyz_first = 0, pool_k = K); rows = B*S*K.
 * Bit-identical to three reart_mlp_layer calls.  Built for the extractor's three scales, (C1, C2, C3, K) =
 * (32, 32, 64, 32), (64, 64, 128, 64), (64, 96, 128, 128) (networks/feature_extractor.py:19-21); anything else returns
 * REART_ERR_UNSUPPORTED (call the layers one by one). */
int reart_mlp_chain3(const int64_t *gather_idx, int K, int S, int Npts, const float *F, const float *Q, const float *C,
                     const float *W1t, const float *b1, int C1, const float *W2t, const float *b2, int C2,
                     const float *W3t, const float *b3, int C3, int rows, float *Y, int ldy, int ycol0, void *stream);

/* The same fusion for a scale of the SECOND set-abstraction level: gathered input [F (D) | Q - C] (D + 3 columns, D % 4 == 0,
 * F 16-byte aligned) -> C1 -> C2 -> C3 -> max over the K rows of a group.  The weights stream through LDS in 16-row slabs,
 * a workgroup carries 128 rows through all three layers (rows % 128 == 0).  Bit-identical to three reart_mlp_layer calls.
 * workspace (16-byte aligned, reart_mlp_chain3_wide_workspace_bytes(D, C1, C2, C3)): every call first writes an image of
 * the three weight matrices there in the order the matrix cores' B fragments are read (one small launch).
 * Built for (C1, C2, C3, K) = (128, 128, 256, 64), (128, 196, 256, 128) (networks/feature_extractor.py:22-23); anything
 * else returns REART_ERR_UNSUPPORTED. */
size_t reart_mlp_chain3_wide_workspace_bytes(int D, int C1, int C2, int C3);
int reart_mlp_chain3_wide(const int64_t *gather_idx, int K, int S, int Npts, const float *F, int D, const float *Q,
                          const float *C, const float *W1t, const float *b1, int C1, const float *W2t, const float *b2,
                          int C2, const float *W3t, const float *b3, int C3, int rows, float *Y, int ldy, int ycol0,
                          void *workspace, size_t workspace_bytes, void *stream);

/* 3-NN inverse-distance interpolation of PointNetFeaturePropagation
 * (networks/pointnet2_utils.py:326-336): xyz1 [B,N,3], xyz2 [B,S2,3], points2 [B,S2,D] ->
 * out [B*N, ldo] columns col0 .. col0 + D (so the cat with the skip features, :338-342, is free).
 * Distances: the reference's matmul expansion (square_distance, :33-55) with torch's CPU rounding,
 * d = ((-2*fma(qz,tz,fma(qy,ty,qx*tx))) + |q|^2) + |t|^2 -- bit-equal to the reference on its CPU path
 * (tests/golden/three_interp.npz); the three smallest by (d, index).
 * reart_three_nn alone: dist3 f32 [B,N,3], idx3 i64 [B,N,3] (= square_distance(...).sort()[:3]). */
int reart_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S2, float *dist3,
                   int64_t *idx3, void *stream);
size_t reart_three_interpolate_workspace_bytes(int B, int N, int S2);
int reart_three_interpolate(const float *xyz1, const float *xyz2, const float *points2, int B,
                            int N, int S2, int D, float *out, int ldo, int col0,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ */
/* Exact K-NN against static target sets (uniform grid)                      */
/* ------------------------------------------------------------------------ */

/* Same result, bit for bit, as reart_knn_points_idx / reart_knn_cuda (same distance expression,
 * ties -> lowest index) for K in {1, 3}, computed through a 16^3 uniform grid over each target
 * set: ~30-50x fewer distance evaluations when the targets do not change between calls (the
 * observed frames and the flow reference sets of the relaxation loop).  This entry builds the grid
 * and queries it in one call; the fused step (reart_relax_prepare/step) builds once.
 *   targets: [E,Nt_max,3] (offsets NULL) or ragged sets concatenated with offsets [E+1] i32;
 *   queries [E,nq,3]; dists [E,nq,K] squared, ascending; idx [E,nq,K] i32.  Every set needs >= K points. */
size_t reart_grid_knn_workspace_bytes(int E, int Nt_max);
int reart_grid_knn(const float *targets, const int32_t *offsets, int E, int Nt_max,
                   const float *queries, int nq, int K, float *dists, int32_t *idx,
                   void *workspace, size_t workspace_bytes, void *stream);

/* Warm-started exact K-NN for repeated searches on slowly moving clouds (the relaxation loop repeats
 * the searches of utils/chamfer.py:78-94 and utils/flow_utils.py:158 every iteration).  Same result,
 * bit for bit, as reart_knn_points_idx for K in {1, 3} (full lengths), whatever `seed` holds: the
 * seeds (neighbour indices of an earlier call, -1 / out-of-range / repeated entries are tolerated)
 * only bound the search so that bounding boxes of 16 consecutive targets can be skipped, which pays
 * when both clouds are stored in a spatially coherent order (k-d tree leaf or Morton order).
 *   p1 [N,P1,3] queries, p2 [N,P2,3] targets; seed [N,P1,K] i32 in/out (receives idx);
 *   dists [N,P1,K] squared, ascending; idx [N,P1,K] i64.  The fused step uses the same kernels. */
size_t reart_knn_points_warm_workspace_bytes(int N, int P1, int P2, int K);
int reart_knn_points_idx_warm(const float *p1, const float *p2, int N, int P1, int P2, int K,
                              int32_t *seed, float *dists, int64_t *idx, void *workspace,
                              size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ */
/* Batched linear assignment (assignment loss)                               */
/* ------------------------------------------------------------------------ */

/* Replaces `[linear_sum_assignment(c) for c in cost]` / `parallel_lap(cost)` of the assignment loss
 * (run_robot.py:172-176, utils/model_utils.py:85-89) for square cost matrices.
 *   cost [B,n,n] fp32, n <= 4096; col4row [B,n] i32 = column assigned to row i (minimum total cost).
 * epsilon-scaling auction + an exact dual certificate in fp64: certified[b] = 1 means the assignment of matrix
 * b is optimal (equal to scipy's whenever the optimum is unique); certified[b] = 0 means the certificate
 * did not close and the caller must solve that matrix with the host solver (the Python wrapper does).
 * price_in (nullable, [B,n] f64): column potentials returned by an earlier call on a similar batch (the loop
 * re-solves slowly moving matrices every assign_gap iterations): warm start; price_out (nullable, [B,n] f64)
 * receives the potentials of this batch (may alias price_in). */
size_t reart_lap_workspace_bytes(int B, int n);
int reart_lap_auction(const float *cost, int B, int n, int32_t *col4row, int32_t *certified,
                      const double *price_in, double *price_out, void *workspace, size_t workspace_bytes,
                      void *stream);

/* The same solve for Euclidean costs between two point sets: cost must be reart_cdist(src, tgt) (src, tgt [B,n,3]) -- what
 * /root/reference/utils/model_utils.py:92-104 (compute_ass_err) and run_robot.py:170-176 build with torch.cdist.  The long
 * single-bidder chains at the end of every phase then recompute their rows from the points instead of reading them
 * (a dependent row read per link otherwise).  Result and potentials identical to reart_lap_auction on the same matrix. */
int reart_lap_auction_points(const float *cost, const float *src, const float *tgt, int B, int n, int32_t *col4row,
                             int32_t *certified, const double *price_in, double *price_out, void *workspace,
                             size_t workspace_bytes, void *stream);

/* A cold solve as a RACE over epsilon schedules: `racers` (1..13) workgroups per matrix run the auction with different
 * (first epsilon, shrink factor) pairs on compute units that would otherwise idle (T-1 = 19 matrices on 256 units); the first
 * racer whose certificate closes publishes its result, the others stop at their next look at the flag.  The fastest schedule
 * depends on the matrix: over five schedules the slowest matrix of a batch is done 20 % earlier than under the best single
 * one.  The ASSIGNMENT is the optimum whichever racer wins; price_out holds the winner's potentials (valid duals, but not
 * reproducible from run to run).  src / tgt as in reart_lap_auction_points, or both NULL; workspace:
 * reart_lap_race_workspace_bytes.  Used where only the assignment matters: /root/reference/utils/model_utils.py:92-104
 * (compute_ass_err) and the assignment refresh of run_robot.py:165-178. */
size_t reart_lap_race_workspace_bytes(int B, int n, int racers);
int reart_lap_auction_race(const float *cost, const float *src, const float *tgt, int B, int n, int racers,
                           int32_t *col4row, int32_t *certified, double *price_out, void *workspace,
                           size_t workspace_bytes, void *stream);

/* The race with WARM racers in the field: price_in / col4row_in are the potentials and the assignment an earlier solve of a
 * similar batch returned; up to three of the `racers` (2..16, at most 13 of them cold) start from them, the others cold.  A loop that re-solves every
 * few iterations (run_robot.py:165-178) need not know whether its matrices moved little -- a warm racer is then done in a
 * fraction of a cold solve -- or jumped (a cold one wins).  price_out != price_in, col4row != col4row_in. */
int reart_lap_auction_race_warm(const float *cost, const float *src, const float *tgt, int B, int n, int racers,
                                const int32_t *col4row_in, const double *price_in, int32_t *col4row, int32_t *certified,
                                double *price_out, void *workspace, size_t workspace_bytes, void *stream);

/* The same solve warm-started from an earlier solve of a similar batch (the loop re-solves every assign_gap
 * iterations): on entry col4row holds that solve's assignment and price_in (required) its potentials; pairs that are
 * still epsilon-tight under the new costs are kept.  Certified like a cold solve.  Use when the costs move smoothly
 * (KinematicModel); with BaseModel's resampled part labels a cold solve is faster. */
int reart_lap_auction_warm(const float *cost, int B, int n, int32_t *col4row, int32_t *certified,
                           const double *price_in, double *price_out, void *workspace, size_t workspace_bytes,
                           void *stream);

/* Re-solve of a slowly moving batch (the kinematic projection, README.md:125: --assign_gap=1 re-solves its T-1 matrices
 * after every Adam step, run_robot.py:165-178): shortest augmenting paths (Jonker-Volgenant) started from the previous
 * solve's assignment (col4row on entry) and potentials (price_in, required) -- pairs that still attain their row's
 * minimum are kept, every other row costs one Dijkstra search over the reduced costs.  Same outputs, same exact
 * certificate as reart_lap_auction; the first solve of a sequence comes from reart_lap_auction.  For n >= 512 the call is
 * four launches on `stream`: the rows' (min, arg-min) under the incoming potentials on the whole chip, the sequential
 * part with one workgroup per matrix, the first certificate round on the whole chip, and the remaining certificate
 * rounds for a matrix that needs them (normally none); the intermediate arrays live in `workspace`. */
int reart_lap_resolve(const float *cost, int B, int n, int32_t *col4row, int32_t *certified,
                      const double *price_in, double *price_out, void *workspace, size_t workspace_bytes,
                      void *stream);

/* Measurement aid for the latency roofline of reart_lap_resolve_points (no reference counterpart): B workgroups run
 * `steps` path-search steps of the re-solve stripped to what cannot be removed -- the workgroup-wide (distance, column)
 * arg-min over n <= 2048 labels held in registers and its one barrier, with the solver's own primitives -- and
 * *h_us_per_step (HOST pointer) receives the slowest workgroup's microseconds per step on the GPU's constant-rate
 * clock.  A re-solve of S sequential steps cannot take less than S x this.  Synchronises `stream`.  workspace: 16 B bytes. */
int reart_lap_step_floor(int B, int n, int steps, void *workspace, size_t workspace_bytes, double *h_us_per_step, void *stream);

/* Measurement aid for the headline iteration (run_robot.py:154-221 as five dependent launches; no reference counterpart):
 * enqueues `iters` times a chain of `nk` <= 8 launches of a kernel that does NO arithmetic.  shape [nk][5] (HOST ints): grid
 * size, block size, dynamic LDS bytes, dependent global loads per thread (a pointer chase through `workspace`, 64 KB of it,
 * L2-resident), workgroup barriers.  Launched with the shapes and dependent-access counts of the step's kernels, the chain's
 * time per iteration is what those launches cost when their arithmetic is free: dispatch, drain and the latency of the
 * dependent chain -- a floor of the five-launch iteration (bench.py reports it as roofline.step_floor_us).  Asynchronous and
 * graph-capturable: the caller times it.  workspace: >= 64 KB + 64 B, initialised by the first call's own set-up launch. */
int reart_relax_step_floor(const int *shape, int nk, int iters, void *workspace, size_t workspace_bytes, void *stream);

/* The same re-solve for Euclidean costs between two point sets, without a cost matrix: src, tgt [B,n,3], n <= 2048;
 * c_ij is the value reart_cdist(src, tgt) would hold (same fp32 expression), recomputed from LDS copies of both sets
 * wherever the solver needs a cost -- a path-search step then reads no memory beyond LDS.  Result identical to
 * reart_lap_resolve on reart_cdist's matrix. */
int reart_lap_resolve_points(const float *src, const float *tgt, int B, int n, int32_t *col4row,
                             int32_t *certified, const double *price_in, double *price_out, void *workspace,
                             size_t workspace_bytes, void *stream);

/* reart_lap_resolve_points with `racers` (2..13) workgroups per problem on otherwise idle compute units (a re-solve keeps
 * one workgroup per problem busy: 19 of 256 compute units at README.md:125's 20 frames).  Every racer runs the same exact
 * algorithm from the same start and differs only in the order in which it takes the free rows (ascending, descending,
 * fixed pseudo-random permutations); the first to finish publishes, the others leave at their next step.  The assignment is the optimum either way; the potentials written to price_out -- and, between optima
 * of exactly equal cost, the assignment -- are the winner's, so not reproducible run to run.  stats[b][0] carries the
 * winning racer in bits 16+.  workspace: reart_lap_race_workspace_bytes(B, n, racers).  n < 512 runs the plain re-solve. */
int reart_lap_resolve_points_race(const float *src, const float *tgt, int B, int n, int racers, int32_t *col4row,
                                  int32_t *certified, const double *price_in, double *price_out, void *workspace,
                                  size_t workspace_bytes, void *stream);

/* reart_lap_resolve_points[_race] with one search per WAVE (csrc/lap_mw.hip): the free rows a refresh leaves (run_robot.py:164-187
 * re-solves every assign_gap iterations) are mostly independent of each other, so every wave of a problem's workgroup follows
 * its own free row -- row-reduction chain, then a shortest augmenting path -- on columns it holds in registers, and commits
 * under a workgroup lock after checking that the columns it is about to write are still as it saw them.  512 <= n <= 2048
 * (REART_ERR_UNSUPPORTED otherwise: call reart_lap_resolve_points_race).  racers >= 1 workgroups per problem (free rows taken
 * in different orders); workspace: reart_lap_race_workspace_bytes(B, n, racers).  Outputs, certificate and the caveat on the
 * potentials as reart_lap_resolve_points_race; stats[b] = released rows (+ winner << 16), rows left for the path searches |
 * commit conflicts << 16, path-search steps, 1 + 256 * row-reduction steps. */
int reart_lap_resolve_points_mw(const float *src, const float *tgt, int B, int n, int racers, int32_t *col4row,
                                int32_t *certified, const double *price_in, double *price_out, void *workspace,
                                size_t workspace_bytes, void *stream);

/* reart_lap_resolve_points_mw with every problem's row reduction spread over `arr_wgs` workgroups (1..256; csrc/lap_mw.hip:
 * lap_mc_arr_kernel -- the chains only meet in the column they commit on, so their state lives in memory and a commit is a
 * lock-free compare-and-swap on the column's owner), the path searches then one workgroup per problem and racer.  Three
 * launches between the two whole-chip passes.  512 <= n <= 2048; workspace: reart_lap_mc_workspace_bytes(B, n, racers).
 * Outputs, certificate, statistics and the caveat on the potentials as reart_lap_resolve_points_mw; certified [B] and the
 * statistics need no clearing by the caller (the set-up launch defines them: no fill launches in front of a refresh). */
size_t reart_lap_mc_workspace_bytes(int B, int n, int racers);
int reart_lap_resolve_points_mc(const float *src, const float *tgt, int B, int n, int racers, int arr_wgs, int32_t *col4row,
                                int32_t *certified, const double *price_in, double *price_out, void *workspace,
                                size_t workspace_bytes, void *stream);

/* Is the optimum a solve returned the ONLY optimal assignment?  (csrc/ties.hip; reference: the refresh is scipy's
 * linear_sum_assignment, a pure function of the cost matrix, run_robot.py:172-176, so a run repeats under --manual_seed,
 * run_robot.py:37-49; the raced solvers above return SOME optimum.)  Inputs: the points of reart_lap_resolve_points, the
 * optimum col4row [B,n] and its column potentials price [B,n] as those calls leave them; n <= 4096.  Outputs (caller-owned):
 *   cols [B,n,K]   per row the columns j != s(i) whose pair is tight under the potentials,
 *                  c_ij + p_j - (c_i,s(i) + p_s(i)) <= 1e-13 x cost scale (the certificate's tolerance), costs by reart_cdist's
 *                  expression -- the first K of them (1 <= K <= 32); cnt [B,n] how many the row has (may exceed K);
 *   tie [B]        0: no alternating cycle among the tight pairs -- the optimum is unique; 1: there is one -- other optima of
 *                  the same cost exist (the host chooses: reart_amd/utils/lap.py canonical_among_ties); 2: a row has more
 *                  than K tight pairs (the host lists them itself); 3: col4row is not an assignment.
 * Two launches on `stream`, no workspace, no host synchronisation. */
int reart_lap_ties(const float *src, const float *tgt, int B, int n, const int32_t *col4row, const double *price,
                   int32_t *tie, int32_t *cols, int32_t *cnt, int K, void *stream);

/* reart_lap_resolve_points_mc followed by the tie check of reart_lap_ties, with the tight pairs listed by the solve's own
 * certificate pass (its scan meets exactly the pairs within the tolerance of the row's minimum: no second pass over the costs).
 * Same inputs / outputs as the two calls; n >= 512.  A problem whose certificate needed more rounds than the pass (its
 * potentials moved afterwards) comes back with tie = 2: the host lists its pairs itself.  Reference: run_robot.py:172-176 (the
 * refresh as a function of the cost matrix), run_robot.py:37-49 (runs repeat under --manual_seed). */
int reart_lap_resolve_points_mc_ties(const float *src, const float *tgt, int B, int n, int racers, int arr_wgs, int32_t *col4row,
                                     int32_t *certified, const double *price_in, double *price_out, int32_t *tie, int32_t *cols,
                                     int32_t *cnt, int K, void *workspace, size_t workspace_bytes, void *stream);

/* Device-side glue of an assignment refresh (run_robot.py:165-178), so that a loop which re-solves on the GPU touches the host
 * only for the B certificate flags (csrc/assign.hip):
 *   reart_gather_points: out[b][r] = pc[b][index[r]] -- `index_points(pc_trans_list, fps_idx)` (run_robot.py:169) for ONE sample
 *     shared by all frames; pc [B,N,3], index [n] (0 <= index[r] < N), out [B,n,3].
 *   reart_assign_pairs: the solved columns as the fused step's pair map (run_robot.py:177-178: `pc_tgt` gathered by the
 *     solver's columns, paired with the sampled source points in order): assign_map[b][p] = -1 where slot_of_point[p] < 0, else
 *     tgt_index[b][col4row[b][slot_of_point[p]]].  col4row [B,n], slot_of_point [N] (sample slot of canonical point p or -1),
 *     tgt_index [B,n] (index of the r-th sampled target point in frame b's cloud), assign_map [B,N] (reart_relax_buffers). */
int reart_gather_points(const float *pc, const int32_t *index, int B, int N, int n, float *out, void *stream);
/*   reart_publish_words: host_out = a[0..na) | b[0..nb) | c[0..nc) (int32 words in device memory; b, c may be empty), written by one
 *     launch into PINNED host memory (`host_out`: the pointer hipHostMalloc / torch's pin_memory returned) -- the certificate
 *     flags, tie flags and statistics the host reads after a refresh, without a copy launch each. */
int reart_publish_words(const int32_t *a, int na, const int32_t *b, int nb, const int32_t *c, int nc, int32_t *host_out, void *stream);
int reart_assign_pairs(const int32_t *col4row, const int32_t *slot_of_point, const int32_t *tgt_index, int B, int N, int n,
                       int32_t *assign_map, void *stream);

/* Cost matrices for the above: replaces `torch.cdist(pc_src, pc_tgt)` (run_robot.py:171, utils/model_utils.py:93).
 *   a [B,n,3], b [B,m,3] -> out [B,n,m] = Euclidean distance, sqrt(((dx*dx)+(dy*dy))+(dz*dz)) in fp32. */
int reart_cdist(const float *a, const float *b, int B, int n, int m, float *out, void *stream);

/* ------------------------------------------------------------------------ */
/* Correspondence matching on the extractor's descriptors                    */
/* ------------------------------------------------------------------------ */

/* Replaces match_smnn(desc1, desc2, th=0.9) (utils/flow_utils.py:48-100; called once per frame pair
 * by compute_corr_list_filter :116-143): nearest / second-nearest descriptor ratio test in both
 * directions and the mutual filter, without the [N1,N2] distance matrix.
 *   desc1 [E,N1,64], desc2 [E,N2,64] (point-major rows);  keep [E,N1] u8 = 1 where point i of desc1
 *   has a mutual match; tgt [E,N1] i64 = its nearest descriptor in desc2 (valid where keep).
 *   The matches of pair e, sorted by source index, are {(i, tgt[e,i]) : keep[e,i]}.
 *   th < 0 skips the ratio test: plain mutual nearest neighbours = the reference's matching="mnn" branch
 *   (k = 1 KNN in both directions + find_mutual_correspondences, utils/flow_utils.py:102-113, 126-137). */
size_t reart_match_smnn_workspace_bytes(int E, int N1, int N2);
int reart_match_smnn(const float *desc1, const float *desc2, int E, int N1, int N2, int D, float th,
                     uint8_t *keep, int64_t *tgt, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ */
/* End-of-run structure extraction (run_robot.py:224-330)                     */
/* ------------------------------------------------------------------------ */

/* Screw decomposition and joint-type costs of relative part motions, one launch.  Replaces
 * compute_relative_trans + compute_geo_cost (utils/graph_utils.py:170-186, :131-167), compute_screw_trans /
 * compute_screw_cost (:235-292), compute_mean_screw_param (:207-232), frobenius_cost (:189-196), the identity
 * cost of merge_graph (:338-342) and the per-edge screw parameters of build_graph
 * (utils/kinematic_utils.py:84-99), with transform_to_dq / dq_to_screw (screw_se3/dq_utils.py:129-182).
 *   trans [T,P,4,4]; pairs [E,2] i32 (src, tgt): the relative motion of edge e in frame t is
 *   inv(trans[t,src]) * trans[t,tgt].  pairs == NULL: trans IS the relative motion, [T,E,4,4] (P ignored).
 *   plain_mean != 0: mean axis / moment over all frames (build_graph's single-edge call); 0: over the frames
 *   that are not a unit transform (compute_mean_screw_param), all frames when every frame is one or E <= 1.
 * Outputs (nullable unless noted): screw [T,E,8] = axis(3), moment(3), theta, distance;  rel [T,E,4,4];
 * mean [E,6] = mean axis, mean moment;  recon [T,E,4,4] = reconstruction with the cheaper joint type;
 * cost [E,4] (required) = revolute cost, prismatic cost, their minimum, mean over frames of |rel - I|^2;
 * mean_cost = mean_e(minimum) / T (compute_screw_cost's scalar).  workspace is needed when mean == NULL. */
size_t reart_screw_fit_workspace_bytes(int T, int E);
int reart_screw_fit(const float *trans, int T, int P, const int32_t *pairs, int E, int plain_mean,
                    float *screw, float *rel, float *mean, float *recon, float *cost, float *mean_cost,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Replaces fps_sample_cano (utils/graph_utils.py:37-52): farthest point sampling inside every part, one
 * workgroup per part.  cano [N,3], seg [N] i64, labels [Ps] i64 -> idx [Ps,num_fps] i64 (indices into cano,
 * start = the part's first point like the reference's CUDA FPS; -1 when the part has fewer than num_fps
 * points), count [Ps] i32 = part sizes.  cuda_mode as in reart_fps. */
int reart_part_fps(const float *cano, const int64_t *seg, int N, const int64_t *labels, int Ps, int num_fps,
                   int cuda_mode, int64_t *idx, int32_t *count, void *stream);

/* Replaces compute_spatial_cost (utils/graph_utils.py:70-84) and compute_joint_cost (:87-100) over all
 * ordered part pairs.  cano_fps [Ps,F,3]; frame_fps [T,Ps,F,3] (nullable: the same points in the predicted
 * frames) -> cano_dist [Ps,Ps] = squared distance of the closest pair (i -> j), pair [Ps,Ps,2] i64 = that
 * pair's (source, target) FPS slots (first minimum), joint [Ps,Ps] = its squared distance summed over frames. */
int reart_part_pair_cost(const float *cano_fps, const float *frame_fps, int T, int Ps, int F, float *cano_dist,
                         int64_t *pair, float *joint, void *stream);

/* Replaces compute_group_temporal_err (utils/model_utils.py:107-118).  pcs [T,N,3], seg [N] i64,
 * labels [Ps] i64 -> per_part [Ps], worst = max over parts. */
int reart_group_temporal_err(const float *pcs, int T, int N, const int64_t *seg, const int64_t *labels, int Ps,
                             float *per_part, float *worst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* REART_HIP_H */
