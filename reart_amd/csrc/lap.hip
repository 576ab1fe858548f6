// reart_amd/csrc/lap.hip -- batched linear assignment on the GPU for the assignment loss
// (reference run_robot.py:164-187: `scipy.optimize.linear_sum_assignment` on (T-1) cost matrices
// `cdist(pc_src, pc_tgt)` of num_fps x num_fps, recomputed every `assign_gap` iterations; the reference
// ships them to a CPU process pool, utils/model_utils.py:85-103 -- SURVEY.md 8f-2).
//
// scipy's solver is a sequential shortest-augmenting-path method (~150 k dependent steps for n = 1024):
// latency-bound on a GPU.  Here:
//   1. forward AUCTION with epsilon scaling (Bertsekas), Jacobi rounds, one workgroup per matrix: every
//      unassigned row bids for its cheapest column at the current prices; a column takes the highest bid.
//      Ends with an assignment whose cost is within n * eps_final of the optimum.
//   2. CERTIFICATE: starting from the auction's prices, Bellman-Ford rounds on the column potentials d
//      (d_sigma(i) <- min_k (c_ik + d_k) - c_i,sigma(i)) until every row's assigned column is an exact
//      arg-min of c_ik + d_k.  Then (u, d) is a dual solution tight on the assignment: the assignment is
//      OPTIMAL (same optimum as scipy's; the permutation is the same whenever the optimum is unique).
//      If the rounds do not settle (the auction's result was not optimal, or an exact tie cycles within
//      rounding), the matrix is reported uncertified and the host falls back to scipy for it.
// All arithmetic on prices / potentials is fp64 on the fp32 costs.  Deterministic: bids meet through
// integer atomics on ordered keys (max bid, then lowest row).
#include "common.h"
#include "internal.h"
#include "lap_dev.h"
#include <math.h>

#define LAP_BS 1024
// lap_auction_kernel: from phase LAP_SEARCH_PHASE on, the last LAP_SEARCH_NU free rows of a phase get one augmenting-path
// search each instead of a bidding chain.  Measured on the tail's 19 x 4096^2 problems (tools/lap_cold.py): the early
// phases' chains are long (up to 12 k links) but they are the price war that settles the duals -- cut short by a search
// (which raises the prices by the least possible amount) the NEXT phase pays with 3-5x the rounds; from the 7th phase
// on (eps <= 3e-6 of the largest cost) nothing is left to settle, the search is simply the shorter way to place the last
// row, and its tighter prices halve the certificate's rounds (36 -> 19): 440 -> 395 ms.
#define LAP_CW 8            // waves that compute in the points form of a single-bidder chain (of LAP_BS / 64 = 16)
#define LAP_SEARCH_PHASE 7
#define LAP_SEARCH_NU 1
#define LAP_NLDS 2048   // up to here the rows' bids live in LDS too; above, in the workspace (36 B of LDS per row/column)

#define LAP_EPS0 0.125      // first epsilon of a cold solve, as a fraction of the largest cost
#define LAP_THETA 6.0       // epsilon shrinks by this factor from phase to phase
struct LapArgs {
    const float *cost;     // [B][n][n]
    int B, n;
    int *col4row;          // [B][n]
    int *certified;        // [B]
    double *price_out;     // [B][n] final potentials (workspace)
    double *pbval_ws;      // [B][n] rows' bids when n > LAP_NLDS (LDS holds the rest of the state)
    int max_rounds_cert;
    int *stats;            // nullable [B][4]: phases, auction rounds, bids, certificate rounds
    double eps0, theta_inv, eps_final;   // first epsilon and final epsilon as fractions of the largest cost, 1 / scaling factor
    const double *price_in; // nullable [B][n]: potentials of an earlier, similar problem (warm start)
    int warm_assign;        // with price_in: col4row holds that problem's assignment; pairs that still satisfy eps-CS are kept
    // nullable [B][n][3]: the point sets whose Euclidean distances `cost` holds (cost == reart_cdist(src, tgt), bit for bit).
    // With them a single-bidder chain recomputes its rows from the points instead of reading them (lap_auction_kernel).
    const float *src, *tgt;
    // race (reart_lap_auction_race): gridDim.y workgroups solve the SAME matrix with different epsilon schedules on compute
    // units that would idle; the first to finish certified publishes its result, the others stop when they see `done`.
    int *done;             // nullable [B]: 0 until a racer has published matrix b
    int warm_racers;       // the last `warm_racers` of gridDim.y start from price_in / col_in (c_lap_race_warm)
    const int *col_in;     // race: the earlier assignment the warm racers read (col4row is the winner's output)
};

// epsilon schedules of the racers (first epsilon as a fraction of the largest cost, shrink factor); racer 0 is the default
#define LAP_SEARCH_ABORTED (-2147483647 - 1)
#define LAP_RACE_COLD 13    // measured with 5 / 8 / 12 cold racers: 4096^2 189 / 190 / 195 ms, 2048^2 73 / 64 / 66 ms, 1024^2 24.6 / 24.7 / 23.2 ms
#define LAP_RACE_MAX 16     // cold + warm racers of one launch
__constant__ double c_lap_race[LAP_RACE_COLD][2] = {{LAP_EPS0, LAP_THETA}, {0.125, 4.0}, {0.03, 6.0}, {0.01, 4.0}, {0.06, 5.0},
                                                     {0.25, 6.0}, {0.125, 8.0}, {0.06, 4.0}, {0.02, 6.0}, {0.03, 5.0},
                                                     {0.5, 5.0}, {0.015, 5.0}, {0.125, 5.0}};
// With the potentials (and assignment) of an earlier, similar batch the LAST racers start warm: (first epsilon, shrink factor,
// 1 = potentials only, 2 = potentials + assignment).  Whether a warm start pays depends on how far the matrices moved, which
// the caller cannot know (the base model's resampled labels make them jump, a settled optimisation does not): racing decides.
#define LAP_RACE_WARM 3
__constant__ double c_lap_race_warm[LAP_RACE_WARM][3] = {{1e-2, 6.0, 2.0}, {1e-3, 6.0, 2.0}, {1e-3, 6.0, 1.0}};


// smallest (value, column) and second smallest value of row i under prices p over the columns [jb, je); all lanes
// get the result (an empty range gives +inf).  Exact selections only, so any split of a row into ranges followed by
// lap_merge_top2 gives the same triple as one scan of the whole row.
__device__ __forceinline__ void lap_row_top2_range(const float *__restrict__ row, const double *__restrict__ p, int jb, int je,
                                                   int lane, double &v1, int &j1, double &v2) {
    v1 = INFINITY; v2 = INFINITY; j1 = 0x7fffffff;
    if ((((uintptr_t)(row + jb)) & 15) == 0) {
        // 16-byte loads, up to 8 per lane in flight (2048 columns per pass): the scan is a dependent global read and
        // its latency, not its bandwidth, is what a bid costs.  Each lane still meets its columns in ascending order.
        const int je4 = jb + ((je - jb) & ~3);
        for (int j0 = jb + 4 * lane; j0 < je4; j0 += 256 * 8) {
            float4 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = *(const float4 *)(row + (j0 + 256 * u < je4 ? j0 + 256 * u : jb));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 256 * u;
                if (j < je4) {
                    const float rr[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        lap_top2_push((double)rr[c] + p[j + c], j + c, v1, j1, v2);
                    }
                }
            }
        }
        jb = je4;     // a tail of at most three columns follows
    }
    for (int j0 = jb + lane; j0 < je; j0 += 64 * 8) {
        float r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = row[j0 + 64 * u < je ? j0 + 64 * u : jb];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + 64 * u;
            if (j < je) {
                lap_top2_push((double)r[u] + p[j], j, v1, j1, v2);
            }
        }
    }
    int pay = 0;
    lap_wave_top2_fast(v1, j1, v2, pay);
}

// the row's minimum of c_ik + p_k alone (the certificate needs nothing else): a third of the instructions of the
// (min, arg-min, second-min) scan, which bound the one-workgroup pass (10 VALU instructions per element)
__device__ __forceinline__ double lap_row_min(const float *__restrict__ row, const double *__restrict__ p, int n, int lane) {
    double m = INFINITY;
    int jb = 0;
    if ((((uintptr_t)row) & 15) == 0) {
        const int n4 = n & ~3;
        for (int j0 = 4 * lane; j0 < n4; j0 += 256 * 8) {
            float4 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = *(const float4 *)(row + (j0 + 256 * u < n4 ? j0 + 256 * u : 0));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 256 * u;
                if (j < n4) {
                    m = fmin(m, fmin(fmin((double)r[u].x + p[j], (double)r[u].y + p[j + 1]),
                                     fmin((double)r[u].z + p[j + 2], (double)r[u].w + p[j + 3])));
                }
            }
        }
        jb = n4;
    }
    for (int j = jb + lane; j < n; j += 64) m = fmin(m, (double)row[j] + p[j]);
    return lap_wave_min_d(m);
}

__device__ __forceinline__ void lap_row_top2(const float *__restrict__ row, const double *__restrict__ p, int n, int lane,
                                             double &v1, int &j1, double &v2) {
    lap_row_top2_range(row, p, 0, n, lane, v1, j1, v2);
}

__device__ __forceinline__ void lap_merge_top2(double ov1, int oj1, double ov2, double &v1, int &j1, double &v2) {
    const bool take = (ov1 < v1) || (ov1 == v1 && oj1 < j1);
    const double lose = take ? v1 : ov1;
    v2 = fmin(fmin(v2, ov2), lose);
    v1 = take ? ov1 : v1;
    j1 = take ? oj1 : j1;
}

// largest entry seen by this thread of the workgroup's matrix: 16-byte loads, eight in flight per thread (one pass
// over n*n floats; a scalar strided loop here cost more than the certificate pass)
template <int BS>
__device__ __forceinline__ float lap_matrix_max(const float *__restrict__ C, size_t total, int tid) {
    float m = 0.f;
    const size_t n4 = ((((uintptr_t)C) & 15) == 0) ? total / 4 : 0;
    const float4 *C4 = (const float4 *)C;
    for (size_t e0 = tid; e0 < n4; e0 += (size_t)BS * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = C4[e0 + (size_t)u * BS < n4 ? e0 + (size_t)u * BS : e0];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
    }
    for (size_t e = 4 * n4 + tid; e < total; e += BS) m = fmaxf(m, C[e]);
    return m;
}

#ifdef REART_PRUNE_PHASE   // diagnostic build only (make -C reart_amd/csrc phase): s_memtime ticks of the path-search loop
// [workgroup (first 32)][phase]: 0..5 the path search (arg-min | barrier | merge | row costs | flip | set-up), 6 = everything
// before the row reduction, 7 = the row reduction, 8 = certificate + outputs
__device__ unsigned long long g_jv_phase[32 * 10];
extern "C" int reart_debug_jv_phase(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_jv_phase), sizeof(g_jv_phase)) != hipSuccess) return REART_ERR_LAUNCH;
    if (reset) { static unsigned long long z[32 * 10]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_jv_phase), z, sizeof(z)); }
    return REART_OK;
}
// path-search lengths of lap_jv_kernel (racer 0 of every problem): [log2 bin of the steps][searches | steps]
__device__ unsigned long long g_jv_hist[16 * 2];
extern "C" int reart_debug_jv_hist(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_jv_hist), sizeof(g_jv_hist)) != hipSuccess) return REART_ERR_LAUNCH;
    if (reset) { static unsigned long long z[16 * 2]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_jv_hist), z, sizeof(z)); }
    return REART_OK;
}
#define JPH_HIST(steps_) do { if (threadIdx.x == 0 && blockIdx.y == 0) { int b_ = 0; while ((1 << b_) < (steps_) && b_ < 15) ++b_; \
    atomicAdd(&g_jv_hist[2 * b_], 1ull); atomicAdd(&g_jv_hist[2 * b_ + 1], (unsigned long long)(steps_)); } } while (0)
__device__ unsigned long long g_auc_phase[32 * 10];   // the auction's: see tools/lap_cold.py for the phase names
extern "C" int reart_debug_auction_phase(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_auc_phase), sizeof(g_auc_phase)) != hipSuccess) return REART_ERR_LAUNCH;
    if (reset) { static unsigned long long z[32 * 10]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_auc_phase), z, sizeof(z)); }
    return REART_OK;
}
__device__ int g_auc_trace[32 * 4];   // workgroup 0, per phase: released rows | rounds | bids | search steps
extern "C" int reart_debug_auction_trace(int *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_auc_trace), sizeof(g_auc_trace)) == hipSuccess ? REART_OK : REART_ERR_LAUNCH;
}
#define APH_TRACE(ph, slot, val) do { if (threadIdx.x == 0 && blockIdx.x == 0 && (ph) < 32) g_auc_trace[(ph) * 4 + (slot)] = (val); } while (0)
#define APH_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x < 32) for (int k_ = 0; k_ < 10; ++k_) g_auc_phase[blockIdx.x * 10 + k_] += jph[k_]; } while (0)
#define JPH_COUNT(k, v) do { jph[k] += (v); } while (0)
#define JPH_DECL unsigned long long jph_t = __builtin_amdgcn_s_memtime(), jph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define JPH(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); jph[k] += n_ - jph_t; jph_t = n_; } while (0)
#define JPH_FLUSH() do { if (threadIdx.x == 0 && blockIdx.x < 32) for (int k_ = 0; k_ < 10; ++k_) g_jv_phase[blockIdx.x * 10 + k_] += jph[k_]; } while (0)
#else
#define JPH_DECL do { } while (0)
#define JPH(k) do { } while (0)
#define JPH_FLUSH() do { } while (0)
#define APH_FLUSH() do { } while (0)
#define APH_TRACE(ph, slot, val) do { } while (0)
#define JPH_COUNT(k, v) do { } while (0)
#define JPH_HIST(steps_) do { } while (0)
#endif

__global__ __launch_bounds__(LAP_BS) void lap_auction_kernel(LapArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    const int n = a.n, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = LAP_BS / 64;
    double *price = (double *)lsm;                                  // [n] column prices
    unsigned long long *bidval = (unsigned long long *)(price + n); // [n] highest bid (ordered key)
    const bool bids_in_lds = n <= LAP_NLDS;
    const int racer = blockIdx.y;                                   // 0 unless this is a race
    double *pbval = bids_in_lds ? (double *)(bidval + n) : a.pbval_ws + ((size_t)racer * a.B + b) * n;   // [n] row's bid
    int *owner = (int *)(bidval + n + (bids_in_lds ? n : 0));       // [n] column -> row
    int *assigned = owner + n;                                      // [n] row -> column
    int *bidder = assigned + n;                                     // [n] winning row of the round
    int *pbobj = bidder + n;                                        // [n] row's bid column
    int *ulist = pbobj + n;                                         // [n] unassigned rows
    __shared__ int s_cnt, s_flag, s_next, s_abort;
    // Loops with ONE barrier per step publish the race's abort decision through a slot per step parity: thread 0 writes slot
    // (step & 1) before the barrier, every wave reads that slot after it, and the slot is not written again before the barrier
    // of step + 1 -- which no wave passes without having read.  (A single word rewritten by thread 0 for the next step could be
    // seen early by a slow wave, which would then leave one barrier before its siblings.)
    __shared__ int s_abp[2];
    __shared__ double s_red[NW], s_red2[NW], s_pv1[NW], s_pv2[NW];
    __shared__ int s_pj1[NW];
    const float *C = a.cost + (size_t)b * n * n;
    JPH_DECL;
    // a race is over for this workgroup once another racer has published the matrix (device-scope atomic load: the flag lives in L2)
    auto lost = [&]() -> int { return a.done ? __hip_atomic_load(a.done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; };
    const bool race = a.done != nullptr && gridDim.y > 1;
    if (tid == 0) { s_abort = 0; s_abp[0] = 0; s_abp[1] = 0; }

    // largest cost
    double mx = (double)lap_matrix_max<LAP_BS>(C, (size_t)n * n, tid);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) s_red[wv] = mx;
    // warm start: potentials of an earlier problem, shifted to be non-negative (bids are ordered as unsigned keys)
    const int warm_slot = (int)blockIdx.y - ((int)gridDim.y - a.warm_racers);          // >= 0: this racer starts warm
    const int warm_mode = a.done ? (warm_slot >= 0 ? (int)c_lap_race_warm[warm_slot][2] : 0) : (a.price_in ? (a.warm_assign ? 2 : 1) : 0);
    const double *price_in = warm_mode ? a.price_in : nullptr;
    double pmin = 0.0;
    if (price_in) {
        pmin = INFINITY;
        for (int j = tid; j < n; j += LAP_BS) pmin = fmin(pmin, price_in[(size_t)b * n + j]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) pmin = fmin(pmin, __shfl_xor(pmin, o, 64));
        if (lane == 0) s_red2[wv] = pmin;
    }
    __syncthreads();
    if (price_in) {
        pmin = INFINITY;
        for (int w = 0; w < NW; ++w) pmin = fmin(pmin, s_red2[w]);
    }
    __syncthreads();
    for (int j = tid; j < n; j += LAP_BS) {
        price[j] = price_in ? price_in[(size_t)b * n + j] - pmin : 0.0;
        owner[j] = -1; assigned[j] = -1; bidval[j] = 0ull; bidder[j] = 0x7fffffff;
    }
    __syncthreads();
    const bool warm_assign = warm_mode == 2;
    if (warm_assign) {
        const int *cin = a.col_in ? a.col_in : a.col4row;
        for (int i = tid; i < n; i += LAP_BS) {
            const int j = cin[(size_t)b * n + i];
            if (j >= 0 && j < n) { assigned[i] = j; owner[j] = i; }
        }
        __syncthreads();
        for (int i = tid; i < n; i += LAP_BS) {        // a repeated column would leave two rows on it: keep the owner only
            const int j = assigned[i];
            if (j >= 0 && owner[j] != i) assigned[i] = -1;
        }
        __syncthreads();
    }
    mx = 0.0;
    for (int w = 0; w < NW; ++w) mx = fmax(mx, s_red[w]);
    if (!(mx > 0.0)) mx = 1.0;
    const double eps_final = mx * a.eps_final;

    int st_phases = 0, st_rounds = 0, st_bids = 0, st_cert = 0;
    // The last free rows of a phase: one shortest augmenting path each instead of a bidding chain.  A lone bidder raises
    // one price by its bid increment, the displaced row bids next, and so on until an unowned column is hit: thousands of
    // dependent row scans per phase, most of them on the same few columns (a price war).  The path search finds where
    // that walk must end and by how much every price on the way must rise in ONE sweep: Dijkstra from the free row over
    //     r'_ik = c_ik + p_k - (c_i,s(i) + p_s(i)) + eps   (>= 0 by epsilon-complementary slackness; 0 back along a pair)
    // fixes the closest column, relaxes its owner's row, until the closest column is unowned; prices of the fixed
    // columns rise by (mu - d_j) and the pairs along the path are flipped.  Every pair still satisfies epsilon-CS
    // afterwards (each row's new column is its exact arg-min), which is all the next phase and the certificate rely on.
    // Thread t holds the labels of the columns t, t + 1024, ... in registers; a step is one coalesced row read and one
    // workgroup arg-min (the same loop as lap_jv_kernel's, with the eps offset).
    constexpr int CPT = LAP_NMAX / LAP_BS;
    __shared__ double s_rv[2][NW];
    __shared__ int s_rj[2][NW];
    __shared__ double s_cv1[2][NW], s_cv2[2][NW];
    __shared__ int s_cj1[2][NW], s_ci0[2][NW];
    auto eps_search = [&](int i0, double eps) -> int {
        int *pred = pbobj;                       // a row's bid column: only live inside a bidding round
        double d[CPT];
        unsigned scanned = 0u, freecol = 0u;
        {
            const float *row = C + (size_t)i0 * n;
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int j = tid + k * LAP_BS;
                d[k] = j < n ? (double)row[j] + price[j] : INFINITY;
                if (j < n) { pred[j] = i0; if (owner[j] < 0) freecol |= 1u << k; }
                else scanned |= 1u << k;
            }
        }
        double mu = 0.0;
        int sink = -1, steps = 0;
        for (int it = 0; ; ++it) {
            double bv = INFINITY;
            int bj = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < CPT; ++k)
                if (!((scanned >> k) & 1u)) {
                    const int key = (tid + k * LAP_BS) | (((freecol >> k) & 1u) ? 0 : (1 << 30));   // unowned columns first
                    if (d[k] < bv || (d[k] == bv && key < bj)) { bv = d[k]; bj = key; }
                }
            lap_wave_argmin_fast(bv, bj);
            const int par = it & 1;
            if (lane == 0) { s_rv[par][wv] = bv; s_rj[par][wv] = bj; }
            if (race && tid == 0) s_abp[par] = ((it & 63) == 0) ? lost() : s_abp[par ^ 1];
            __syncthreads();
            if (race && s_abp[par]) return LAP_SEARCH_ABORTED;
            // the waves' minima meet in the first NW <= 16 lanes of every wave: four butterfly steps, then a broadcast
            bv = lane < NW ? s_rv[par][lane] : INFINITY; bj = lane < NW ? s_rj[par][lane] : 0x7fffffff;
            lap_lanes_argmin<(NW <= 2 ? 1 : (NW <= 4 ? 2 : (NW <= 8 ? 3 : 4)))>(bv, bj);
            ++steps;
            mu = bv;
            if (bj == 0x7fffffff || !(bv < INFINITY)) break;          // non-finite costs only
            const int jstar = bj & ~(1 << 30);
            if ((jstar & (LAP_BS - 1)) == tid) scanned |= 1u << (jstar / LAP_BS);
            const int i = owner[jstar];
            if (i < 0) { sink = jstar; break; }
            const float *row = C + (size_t)i * n;
            float rc[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) rc[k] = row[tid + k * LAP_BS < n ? tid + k * LAP_BS : 0];
            const double base = mu - ((double)row[jstar] + price[jstar]) + eps;
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int j = tid + k * LAP_BS;
                if (!((scanned >> k) & 1u)) {
                    const double nd = base + ((double)rc[k] + price[j]);
                    if (nd < d[k]) { d[k] = nd; pred[j] = i; }
                }
            }
        }
        __syncthreads();                         // every thread has read the prices it needs
        if (sink >= 0) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const int j = tid + k * LAP_BS;
                if (j < n && ((scanned >> k) & 1u) && j != sink) price[j] += mu - d[k];
            }
            if (tid == 0) {
                int j = sink;
                for (;;) {
                    const int i = pred[j];
                    const int jn = assigned[i];
                    assigned[i] = j; owner[j] = i;
                    if (i == i0) break;
                    j = jn;
                }
            }
        }
        __syncthreads();
        return sink >= 0 ? steps : -steps - 1;
    };
    JPH(0);
    const double eps0 = !race ? a.eps0 : (warm_slot >= 0 ? c_lap_race_warm[warm_slot][0] : c_lap_race[racer][0]);
    const double theta_inv = !race ? a.theta_inv : 1.0 / (warm_slot >= 0 ? c_lap_race_warm[warm_slot][1] : c_lap_race[racer][1]);
    for (double eps = mx * eps0; ; eps = fmax(eps * theta_inv, eps_final)) {
        ++st_phases;
        // a phase keeps the prices and every pair that already satisfies the new, tighter epsilon-complementary
        // slackness  c_i,s(i) + p_s(i) <= min_k (c_ik + p_k) + eps ; the other rows are released and bid again
        // The scan that finds the rows to release also is their first bid of the phase (prices do not move in between):
        // the released rows bid right here and the first round below starts at its resolution step.
        const bool pre_bid = st_phases > 1 || warm_assign;
        if (pre_bid) {
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            for (int i = wv; i < n; i += NW) {
                double v1, v2;
                int j1;
                lap_row_top2(C + (size_t)i * n, price, n, lane, v1, j1, v2);
                if (lane == 0) {
                    const int j = assigned[i];
                    if (j < 0 || (double)C[(size_t)i * n + j] + price[j] > v1 + eps) {
                        if (j >= 0) owner[j] = -1;
                        assigned[i] = -1;
                        if (!(v2 < INFINITY)) v2 = v1;
                        const double bid = price[j1] + (v2 - v1) + eps;
                        pbobj[i] = j1; pbval[i] = bid;
                        atomicMax(&bidval[j1], lap_key(bid));
                        ulist[atomicAdd(&s_cnt, 1)] = i;
                    }
                }
            }
        }
        __syncthreads();
        JPH(1);
        [[maybe_unused]] const int tr_rounds0 = st_rounds, tr_bids0 = st_bids;
        [[maybe_unused]] int tr_search = 0;
        APH_TRACE(st_phases - 1, 0, pre_bid ? s_cnt : n);
        bool first = pre_bid;
        for (;;) {
            if (!first) {
                if (tid == 0) { s_cnt = 0; if (race) s_abort = lost(); }
                __syncthreads();
                if (s_abort) return;                                   // uniform: every thread reads the same LDS word
                for (int i = tid; i < n; i += LAP_BS)
                    if (assigned[i] < 0) ulist[atomicAdd(&s_cnt, 1)] = i;
                __syncthreads();
            }
            const int nu = s_cnt;
            JPH(2);
            if (nu == 0) break;
            ++st_rounds; st_bids += nu;
            const bool skip_bids = first;      // the phase's first round: the bids were placed by the scan above
            first = false;
            if (!skip_bids) {
            // One wave covers 1024 columns with four 16-byte loads per lane in flight; longer rows are shared by n / 1024
            // waves whose (min, arg-min, second-min) triples one wave merges.
            const int wmax = n >= 2048 ? (n / 1024 < NW ? n / 1024 : NW) : 1;
            if (nu <= LAP_SEARCH_NU && st_phases >= LAP_SEARCH_PHASE) {
                JPH_COUNT(7, nu);
                int done = 0;
                for (; done < nu; ++done) {
                    const int r = eps_search(ulist[done], eps);
                    if (r == LAP_SEARCH_ABORTED) return;
                    st_bids += r >= 0 ? r : -r - 1;
                    JPH_COUNT(8, r >= 0 ? r : -r - 1);
                    tr_search += r >= 0 ? r : -r - 1;
                    if (r < 0) break;            // non-finite costs: leave the row to the bids below
                }
                JPH(4);
                if (done > 0) continue;          // the list is rebuilt (empty unless a search gave up)
            }
            if (nu == 1) {
                // a single bidder: no conflicts are possible, so the chain (the row bids, takes the column, the displaced
                // owner bids next, ...) is followed without rebuilding the bidder list until nobody is displaced.  The end
                // of a phase is mostly such chains; every link is one row scan, a dependent read.
                int i = ulist[0];
                [[maybe_unused]] const int st_bids0 = st_bids;
                if (a.src && a.tgt && wmax > 1) {       // n >= 2048; below, one wave walks the chain without any barrier (faster: measured)
                    // Points form: the chain's rows are recomputed, not read.  Every link of the matrix form is a dependent
                    // row read (3.2 us at n = 4096; a phase's last row walks thousands of links).  Here the source points
                    // are staged once per chain into LDS arrays that are idle between rounds (pbobj | ulist | bidder);
                    // LAP_CW of the 16 waves each keep the target points, prices and owners of n / (64 LAP_CW) columns
                    // per lane in registers (with all 16 waves a link is bound by the VALU work of its reductions: 3.0 us).
                    // A link: the row's distances (reart_cdist's expression: the same bits as the matrix), the wave's
                    // minimum by an fp64 butterfly, the arg-min's column / owner read from the ONE lane that holds the
                    // minimum (exact ties take the full (min, arg-min, second-min) butterfly instead), the second minimum by
                    // another butterfly, ONE barrier, the same meeting of the waves' results in the first lanes of every
                    // wave, the update by the arg-min's thread.  Same bids, prices and assignment as the matrix form.
                    constexpr int CTH = 64 * LAP_CW, CPC = LAP_NMAX / CTH;
                    float *sx = (float *)pbobj, *sy = (float *)ulist, *sz = (float *)bidder;
                    const float *S3 = a.src + (size_t)b * n * 3, *T3 = a.tgt + (size_t)b * n * 3;
                    __syncthreads();                                         // everyone has read ulist[0]
                    for (int r = tid; r < n; r += LAP_BS) { sx[r] = S3[3 * r]; sy[r] = S3[3 * r + 1]; sz[r] = S3[3 * r + 2]; }
                    const bool act = wv < LAP_CW;
                    float tcx[CPC], tcy[CPC], tcz[CPC];
                    double pr[CPC];
                    int own[CPC];
#pragma unroll
                    for (int k = 0; k < CPC; ++k) {
                        const int j = tid + k * CTH;
                        const bool ok = act && j < n;
                        tcx[k] = ok ? T3[3 * j] : 0.f; tcy[k] = ok ? T3[3 * j + 1] : 0.f; tcz[k] = ok ? T3[3 * j + 2] : 0.f;
                        pr[k] = ok ? price[j] : INFINITY;
                        own[k] = ok ? owner[j] : -1;
                    }
                    __syncthreads();
                    int par = 0;
                    for (;;) {
                        double v1 = INFINITY, v2 = INFINITY;
                        int j1 = 0x7fffffff, i0 = -1;
                        if (act) {
                            const float ax = sx[i], ay = sy[i], az = sz[i];
#pragma unroll
                            for (int k = 0; k < CPC; ++k) {
                                lap_top2_push((double)sqrtf(reart_sqdist3(ax, ay, az, tcx[k], tcy[k], tcz[k])) + pr[k], tid + k * CTH, own[k],
                                              v1, j1, v2, i0);
                            }
                            lap_wave_top2_fast(v1, j1, v2, i0);
                            if (lane == 0) { s_cv1[par][wv] = v1; s_cv2[par][wv] = v2; s_cj1[par][wv] = j1; s_ci0[par][wv] = i0; }
                        }
                        if (race && tid == 0) s_abp[par] = ((st_bids & 31) == 0) ? lost() : s_abp[par ^ 1];
                        __syncthreads();
                        if (race && s_abp[par]) return;
                        v1 = lane < LAP_CW ? s_cv1[par][lane] : INFINITY; v2 = lane < LAP_CW ? s_cv2[par][lane] : INFINITY;
                        j1 = lane < LAP_CW ? s_cj1[par][lane] : 0x7fffffff; i0 = lane < LAP_CW ? s_ci0[par][lane] : -1;
                        static_assert(LAP_CW == 8 || LAP_CW == 16, "8 (measured best) or all 16 waves compute");
                        lap_lanes_top2<(LAP_CW == 8 ? 3 : 4)>(v1, j1, v2, i0);
                        par ^= 1;
                        if (!(v2 < INFINITY)) v2 = v1;
                        if (act && (j1 & (CTH - 1)) == tid) {                // the arg-min's thread
                            const int kk = j1 / CTH;
#pragma unroll
                            for (int k = 0; k < CPC; ++k)
                                if (k == kk) { pr[k] = pr[k] + (v2 - v1) + eps; own[k] = i; }
                            assigned[i] = j1;
                            if (i0 >= 0) assigned[i0] = -1;
                        }
                        ++st_bids;
                        if (i0 < 0) break;
                        i = i0;
                    }
                    __syncthreads();                                         // the last reads of the staged points
#pragma unroll
                    for (int k = 0; k < CPC; ++k) {
                        const int j = tid + k * CTH;
                        if (act && j < n) { price[j] = pr[k]; owner[j] = own[k]; }
                    }
                    for (int j = tid; j < n; j += LAP_BS) bidder[j] = 0x7fffffff;        // its resting value between rounds
                    __syncthreads();
                    JPH(4);
                    JPH_COUNT(8, st_bids - st_bids0);
                    tr_search += st_bids - st_bids0;
                    continue;
                }
                if (wmax == 1) {
                    if (wv == 0) {                       // no workgroup barrier inside the chain
                        for (;;) {
                            double v1, v2;
                            int j1;
                            lap_row_top2(C + (size_t)i * n, price, n, lane, v1, j1, v2);
                            if (!(v2 < INFINITY)) v2 = v1;
                            const int prev = owner[j1];
                            if (lane == 0) {
                                price[j1] = price[j1] + (v2 - v1) + eps;
                                owner[j1] = i; assigned[i] = j1;
                                if (prev >= 0) assigned[prev] = -1;
                            }
                            ++st_bids;
                            if (prev < 0) break;
                            i = prev;
                            if (race && (st_bids & 63) == 0 && lost()) { if (lane == 0) s_abort = 1; break; }   // one wave: uniform
                        }
                    }
                    __syncthreads();
                    if (s_abort) return;
                    JPH(4);
                    JPH_COUNT(8, st_bids - st_bids0);
                    tr_search += st_bids - st_bids0;
                    continue;
                }
                const int len = (((n + wmax - 1) / wmax) + 63) & ~63;
                for (;;) {
                    double v1, v2;
                    int j1;
                    if (wv < wmax) {
                        lap_row_top2_range(C + (size_t)i * n, price, min(n, wv * len), min(n, (wv + 1) * len), lane, v1, j1, v2);
                        if (lane == 0) { s_pv1[wv] = v1; s_pv2[wv] = v2; s_pj1[wv] = j1; }
                    }
                    __syncthreads();
                    if (wv == 0) {
                        v1 = INFINITY; v2 = INFINITY; j1 = 0x7fffffff;
                        if (lane < wmax) { v1 = s_pv1[lane]; v2 = s_pv2[lane]; j1 = s_pj1[lane]; }
#pragma unroll
                        for (int o = NW / 2; o >= 1; o >>= 1)
                            lap_merge_top2(__shfl_xor(v1, o, 64), __shfl_xor(j1, o, 64), __shfl_xor(v2, o, 64), v1, j1, v2);
                        if (lane == 0) {
                            if (!(v2 < INFINITY)) v2 = v1;
                            const int prev = owner[j1];
                            price[j1] = price[j1] + (v2 - v1) + eps;
                            owner[j1] = i; assigned[i] = j1;
                            if (prev >= 0) assigned[prev] = -1;
                            s_next = prev;
                        }
                    }
                    if (race && tid == 0 && (st_bids & 31) == 0) s_abort = lost();
                    __syncthreads();
                    if (s_abort) return;
                    ++st_bids;
                    i = s_next;
                    if (i < 0) break;
                }
                JPH(4);
                JPH_COUNT(8, st_bids - st_bids0);
                tr_search += st_bids - st_bids0;
                continue;
            }
            if (nu * 2 <= NW && wmax > 1) {
                // few bidders on long rows: several waves per row (a power of two), merged by one thread per row
                int wpr = 2;
                while (wpr * 2 * nu <= NW && wpr * 2 <= wmax) wpr *= 2;
                const int g = wv / wpr, seg = wv % wpr;
                const int len = (((n + wpr - 1) / wpr) + 63) & ~63;
                if (g < nu) {
                    double v1, v2;
                    int j1;
                    lap_row_top2_range(C + (size_t)ulist[g] * n, price, min(n, seg * len), min(n, (seg + 1) * len), lane, v1, j1, v2);
                    if (lane == 0) { s_pv1[wv] = v1; s_pv2[wv] = v2; s_pj1[wv] = j1; }
                }
                __syncthreads();
                if (tid < nu) {
                    const int i = ulist[tid];
                    double v1 = INFINITY, v2 = INFINITY;
                    int j1 = 0x7fffffff;
                    for (int w = tid * wpr; w < (tid + 1) * wpr; ++w) lap_merge_top2(s_pv1[w], s_pj1[w], s_pv2[w], v1, j1, v2);
                    if (!(v2 < INFINITY)) v2 = v1;
                    const double bid = price[j1] + (v2 - v1) + eps;
                    pbobj[i] = j1; pbval[i] = bid;
                    atomicMax(&bidval[j1], lap_key(bid));
                }
            } else
            // bids: one wave per unassigned row
            for (int u = wv; u < nu; u += NW) {
                const int i = ulist[u];
                double v1, v2;
                int j1;
                lap_row_top2(C + (size_t)i * n, price, n, lane, v1, j1, v2);
                if (lane == 0) {
                    if (!(v2 < INFINITY)) v2 = v1;                  // n == 1
                    const double bid = price[j1] + (v2 - v1) + eps;
                    pbobj[i] = j1; pbval[i] = bid;
                    atomicMax(&bidval[j1], lap_key(bid));
                }
            }
            }
            __syncthreads();
            JPH(3);
            for (int u = tid; u < nu; u += LAP_BS) {
                const int i = ulist[u], j = pbobj[i];
                if (lap_key(pbval[i]) == bidval[j]) atomicMin(&bidder[j], i);
            }
            __syncthreads();
            for (int j = tid; j < n; j += LAP_BS) {
                const int w = bidder[j];
                if (w != 0x7fffffff) {
                    const int prev = owner[j];
                    if (prev >= 0) assigned[prev] = -1;
                    owner[j] = w; assigned[w] = j;
                    price[j] = __longlong_as_double((long long)bidval[j]);
                    bidder[j] = 0x7fffffff;
                }
                bidval[j] = 0ull;
            }
            __syncthreads();
            JPH(5);
        }
        APH_TRACE(st_phases - 1, 1, st_rounds - tr_rounds0);
        APH_TRACE(st_phases - 1, 2, st_bids - tr_bids0);
        APH_TRACE(st_phases - 1, 3, tr_search);
        if (eps <= eps_final) break;
        __syncthreads();
    }

    // ---- certificate: potentials d (start: the prices) such that every assigned column is an exact arg-min
    double *d = price;
    const double tol = mx * 1e-13;
    int certified = 0;
    for (int round = 0; round < a.max_rounds_cert; ++round) {
        if (tid == 0) { s_flag = 0; if (race) s_abort = lost(); }
        ++st_cert;
        __syncthreads();
        if (s_abort) return;
        // Jacobi round: m_i = min_k (c_ik + d_k) with the old d; new d_sigma(i) = m_i - c_i,sigma(i)
        for (int i = wv; i < n; i += NW) {
            const double v1 = lap_row_min(C + (size_t)i * n, d, n, lane);
            if (lane == 0) {
                const int j = assigned[i];
                const double cur = (double)C[(size_t)i * n + j] + d[j];
                pbval[i] = (cur - v1 > tol) ? v1 - (double)C[(size_t)i * n + j] : d[j];
                if (cur - v1 > tol) s_flag = 1;
            }
        }
        __syncthreads();
        const int changed = s_flag;
        for (int i = tid; i < n; i += LAP_BS) d[assigned[i]] = pbval[i];
        __syncthreads();
        if (!changed) { certified = 1; break; }
    }
    if (race) {
        // the first CERTIFIED racer publishes (certified[b] was cleared before the launch: nobody certified = host solve)
        if (tid == 0) s_flag = certified && atomicCAS(a.done + b, 0, racer + 1) == 0;
        __syncthreads();
        if (!s_flag) return;
    }
    for (int i = tid; i < n; i += LAP_BS) a.col4row[(size_t)b * n + i] = assigned[i];
    if (a.price_out)
        for (int j = tid; j < n; j += LAP_BS) a.price_out[(size_t)b * n + j] = d[j];
    if (tid == 0) a.certified[b] = certified;
    JPH(6);
    APH_FLUSH();
    if (tid == 0 && a.stats) { int *o = a.stats + 4 * b; o[0] = st_phases + (race ? racer << 16 : 0); o[1] = st_rounds; o[2] = st_bids; o[3] = st_cert; }
}

extern "C" size_t reart_lap_workspace_bytes(int B, int n);
// workspace of reart_lap_auction_race: the plain layout, one row-bid array per racer, the `done` flags
extern "C" size_t reart_lap_race_workspace_bytes(int B, int n, int racers) {
    if (B < 0 || n < 1 || n > LAP_NMAX || racers < 1 || racers > (LAP_RACE_MAX > JV_RACE_MAX ? LAP_RACE_MAX : JV_RACE_MAX)) return 0;
    return reart_lap_workspace_bytes(B, n) + reart_align_up(sizeof(double) * (size_t)B * n, 256) * (size_t)racers +
           reart_align_up(sizeof(int) * (size_t)B, 256);
}

extern "C" size_t reart_lap_workspace_bytes(int B, int n) {
    if (B < 0 || n < 1 || n > LAP_NMAX) return 0;
    // potentials | diagnostics [B][4] | row bids (n > LAP_NLDS; always reserved) | the re-solve's pass results: v1, cur [B][n] f64,
    // j1 [B][n] i32, scale [B] f64, cert_bad [B] i32
    return reart_align_up(sizeof(double) * (size_t)B * n, 256) * 4 + reart_align_up(sizeof(int) * 4 * (size_t)B, 256) +
           reart_align_up(sizeof(int) * (size_t)B * n, 256) + reart_align_up(sizeof(double) * (size_t)B, 256) +
           reart_align_up(sizeof(int) * (size_t)B, 256);
}

// cost [B,n,n] fp32 (row-major: rows = sources), n <= 4096.  col4row [B,n] i32: column assigned to each row
// (minimum total cost); certified [B] i32: 1 when the dual certificate closed (the assignment is optimal),
// 0 when the caller must solve that matrix on the host.  price_in (nullable, [B,n] f64): potentials returned for
// an earlier, similar batch (the loop re-solves slowly moving matrices) -- the auction then starts from them
// with a small epsilon; price_out (nullable, [B,n] f64) receives this batch's potentials (may alias price_in).
static int lap_launch(const float *cost, int B, int n, int32_t *col4row, int32_t *certified, const double *price_in,
                      double *price_out, int warm_assign, void *workspace, size_t workspace_bytes, void *stream,
                      const float *src = nullptr, const float *tgt = nullptr, int racers = 1, const int32_t *col_in = nullptr) {
    if (B < 0 || n < 1 || n > LAP_NMAX) return REART_ERR_INVALID_ARG;
    if (B == 0) return REART_OK;
    if (!cost || !col4row || !certified) return REART_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < reart_lap_workspace_bytes(B, n)) return REART_ERR_INVALID_ARG;
    LapArgs a = {};
    a.cost = cost; a.B = B; a.n = n; a.col4row = col4row; a.certified = certified;
    a.price_out = price_out ? price_out : (double *)workspace; a.price_in = price_in;
    a.warm_assign = warm_assign;
    a.src = src; a.tgt = tgt;
    a.max_rounds_cert = 4 * n;
    {   // tuning knobs (defaults measured on the loop's matrices)
        a.eps0 = price_in ? (warm_assign ? 1e-2 : 1e-3) : LAP_EPS0;
        a.theta_inv = 1.0 / LAP_THETA;
        a.eps_final = 1e-11;
    }
    a.stats = (int *)((char *)workspace + reart_align_up(sizeof(double) * (size_t)B * n, 256));   // diagnostics, after the potentials
    a.pbval_ws = (double *)((char *)a.stats + reart_align_up(sizeof(int) * 4 * (size_t)B, 256));
    const size_t lds = (size_t)n * ((n <= LAP_NLDS ? 3 : 2) * 8 + 5 * 4);
    if (lds > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute((const void *)lap_auction_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
        return REART_ERR_LAUNCH;
    if (racers > 1) {
        if (racers > LAP_RACE_MAX) return REART_ERR_INVALID_ARG;
        // with the potentials (and assignment) of an earlier batch the last racers start warm; price_out / col4row must not be
        // those inputs (the winner writes them while slower racers may still be reading)
        a.warm_racers = price_in ? (racers - 1 < LAP_RACE_WARM ? racers - 1 : LAP_RACE_WARM) : 0;
        if (racers - a.warm_racers > LAP_RACE_COLD) return REART_ERR_INVALID_ARG;
        if (a.warm_racers && (price_in == a.price_out || col_in == col4row || !col_in)) return REART_ERR_INVALID_ARG;
        a.col_in = col_in;
        if (workspace_bytes < reart_lap_race_workspace_bytes(B, n, racers)) return REART_ERR_INVALID_ARG;
        char *w = (char *)workspace + reart_lap_workspace_bytes(B, n);
        a.pbval_ws = (double *)w;                                             // [racers][B][n]
        a.done = (int *)(w + reart_align_up(sizeof(double) * (size_t)B * n, 256) * (size_t)racers);
        if (hipMemsetAsync(a.done, 0, sizeof(int) * (size_t)B, (hipStream_t)stream) != hipSuccess ||
            hipMemsetAsync(certified, 0, sizeof(int32_t) * (size_t)B, (hipStream_t)stream) != hipSuccess)
            return REART_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(lap_auction_kernel, dim3(B, racers), dim3(LAP_BS), lds, (hipStream_t)stream, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// A cold solve as a RACE: `racers` (2..5) workgroups per matrix run the auction with different epsilon schedules on compute
// units that would idle (T-1 = 19 matrices on 256), the first to finish with its certificate closed publishes its result,
// the others stop at their next check of the flag.  Which schedule is fastest depends on the matrix (their solve times
// correlate 0.2-0.7): over five schedules the slowest matrix of a batch finishes 20 % earlier than under the best single
// schedule.  The ASSIGNMENT is the optimum whichever racer wins; the potentials returned are the winner's (any racer's are
// valid duals) and may differ from run to run.  src / tgt as in reart_lap_auction_points, or both NULL.
extern "C" int reart_lap_auction_race(const float *cost, const float *src, const float *tgt, int B, int n, int racers,
                                      int32_t *col4row, int32_t *certified, double *price_out, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    if ((src == nullptr) != (tgt == nullptr) || racers < 1) return REART_ERR_INVALID_ARG;
    return lap_launch(cost, B, n, col4row, certified, nullptr, price_out, 0, workspace, workspace_bytes, stream, src, tgt, racers);
}

// The race with WARM racers among the field: price_in / col4row_in are the potentials and the assignment of an earlier solve
// of a similar batch (reart_lap_auction* outputs); up to three of the racers start from them (potentials + assignment at
// two first epsilons, potentials alone), the others cold.  A loop that re-solves every few iterations does not have to know
// whether its matrices moved little (a warm racer is done in a fraction of a cold solve) or jumped (a cold one wins).
// price_out != price_in and col4row != col4row_in.
extern "C" int reart_lap_auction_race_warm(const float *cost, const float *src, const float *tgt, int B, int n, int racers,
                                           const int32_t *col4row_in, const double *price_in, int32_t *col4row,
                                           int32_t *certified, double *price_out, void *workspace, size_t workspace_bytes,
                                           void *stream) {
    if ((src == nullptr) != (tgt == nullptr) || racers < 2 || !price_in || !col4row_in || !price_out) return REART_ERR_INVALID_ARG;
    return lap_launch(cost, B, n, col4row, certified, price_in, price_out, 1, workspace, workspace_bytes, stream, src, tgt, racers,
                      col4row_in);
}

extern "C" int reart_lap_auction(const float *cost, int B, int n, int32_t *col4row, int32_t *certified,
                                 const double *price_in, double *price_out, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    return lap_launch(cost, B, n, col4row, certified, price_in, price_out, 0, workspace, workspace_bytes, stream);
}

// The same solve when the costs are Euclidean distances between two point sets: cost == reart_cdist(src, tgt) (src, tgt
// [B,n,3]).  The matrix still serves the bulk of the bidding (a row scan of a matrix is one load and eight instructions
// per element, recomputing it is twenty-five); the long single-bidder chains at the end of every phase recompute their
// rows from the points and read no memory.  Result and potentials identical to reart_lap_auction on the same matrix.
extern "C" int reart_lap_auction_points(const float *cost, const float *src, const float *tgt, int B, int n, int32_t *col4row,
                                        int32_t *certified, const double *price_in, double *price_out, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    if (B > 0 && (!src || !tgt)) return REART_ERR_INVALID_ARG;
    return lap_launch(cost, B, n, col4row, certified, price_in, price_out, 0, workspace, workspace_bytes, stream, src, tgt);
}

// Warm start from an earlier solve of a SIMILAR batch: col4row holds that solve's assignment on entry, price_in its
// potentials.  Pairs that still satisfy epsilon-complementary slackness under the new costs are kept, the other rows
// bid again; the result is certified exactly like a cold solve.  Pays when the costs move smoothly between solves
// (the kinematic projection: -35 % solver time); with the base model's resampled part labels a cold solve is faster.
extern "C" int reart_lap_auction_warm(const float *cost, int B, int n, int32_t *col4row, int32_t *certified,
                                      const double *price_in, double *price_out, void *workspace, size_t workspace_bytes,
                                      void *stream) {
    if (!price_in) return REART_ERR_INVALID_ARG;
    return lap_launch(cost, B, n, col4row, certified, price_in, price_out, 1, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------------
// Re-solve of a SLOWLY MOVING problem: shortest augmenting paths from the previous solve's assignment and potentials
// (the kinematic projection re-solves its (T-1) matrices every assign_gap iterations, run_robot.py:165-178, and one Adam
// step moves the costs by ~1e-3: README.md:125 runs it with --assign_gap=1).
//
// With prices p (column potentials of the previous solve) the row potentials are u_i = min_k (c_ik + p_k): dual feasible
// by construction.  A previous pair (i, s(i)) is kept when s(i) still attains that minimum (within `keep_tol`); a freed
// row takes its arg-min column when that is unowned (lowest row wins); every row still free then gets ONE Dijkstra
// search over reduced costs r_ik = c_ik + p_k - u_i (Jonker-Volgenant augmentation): columns are labelled with their
// shortest distance, the closest unlabelled column is fixed, its owner's row is relaxed, until an unowned column is
// reached; potentials move by (mu - d_j) on the fixed columns and the path is flipped.  The whole workgroup runs one
// search: thread t owns the columns t, t + 1024, ... (labels in registers), the arg-min is one wave reduction + 16
// values in LDS, the relaxation one coalesced row read.  A search costs a few dependent row reads when the old
// potentials are nearly right -- against the tens of thousands of bids the auction needs from the same start.
// The result is certified exactly like the auction's (same Jacobi rounds on the potentials).
// ------------------------------------------------------------------------------------------------------------
// BS threads per workgroup (one matrix each); JV_CPT = 4096 / BS columns per thread.  Fewer waves make a Dijkstra step
// cheaper (the arg-min meets in fewer LDS slots, the barrier joins fewer waves) but the row-scan passes slower.
// Racing re-solve (reart_lap_resolve_points_race): gridDim.y workgroups per matrix run the SAME exact algorithm from the same
// start and differ only in the order in which they take the free rows -- the number of sequential steps of a re-solve varies by +-18 % with that
// order, the first racer to finish publishes and the others leave at their next step.  Racer 0 is the plain solver
// (ascending rows), 1 takes them descending, the others in a fixed pseudo-random order k -> k * prime mod count.

// smallest (value, column) and second smallest value of c_ij + p_j over the columns of row i, costs from the points
__device__ __forceinline__ void lap_row_top2_pts(float ax, float ay, float az, const float *__restrict__ tx,
                                                 const float *__restrict__ ty, const float *__restrict__ tz,
                                                 const double *__restrict__ p, int jb, int je, int lane, double &v1, int &j1,
                                                 double &v2) {
    v1 = INFINITY; v2 = INFINITY; j1 = 0x7fffffff;
    for (int j0 = jb + lane; j0 < je; j0 += 64 * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 64 * u;
            if (j < je) {
                lap_top2_push((double)sqrtf(reart_sqdist3(ax, ay, az, tx[j], ty[j], tz[j])) + p[j], j, v1, j1, v2);
            }
        }
    }
    int pay = 0;
    lap_wave_top2_fast(v1, j1, v2, pay);
}


// The row's smallest c_ij + p_j and its lowest column ALONE (no second minimum), from a seed (v1, j1) every lane holds -- the
// row's own column when the caller has one: for most rows of a re-solve that IS the minimum.  A column can only matter if
// c <= v1 - p_j, i.e. if its SQUARED distance <= (v1 - p_j)^2: the correctly rounded square root (a dozen instructions), the
// conversion and the sum are skipped for every column that fails this -- nearly all of them once v1 is close to the minimum.
// The test errs on the side of passing (slack for the rounding of the difference, 1 + 2^-20 for the root's and the square's),
// and what passes goes through the same expression and the same tie rule (the lowest column) as lap_row_top2_pts:
// the same (v1, j1), bit for bit.
// EMIT (the certificate pass of a --deterministic solve, ties.hip): the columns that pass the test against cur + tol are exactly
// the candidates for TIGHT pairs of the row -- c_ij + p_j - cur <= tol, the cycle check's edges -- so the pass lists them on
// the way: their columns into the row's slots edges[0 .. cap), their number into *n_edges; the row's own column left out.
template <bool EMIT = false>
__device__ __forceinline__ void lap_row_min_pts_seeded(float ax, float ay, float az, const float *__restrict__ tx,
                                                       const float *__restrict__ ty, const float *__restrict__ tz,
                                                       const double *__restrict__ p, int n, int lane, double &v1, int &j1,
                                                       int row = 0, int own = -1, double cur = 0.0, double tol = 0.0, int *edges = nullptr,
                                                       int *n_edges = nullptr, int cap = 0) {
    unsigned long long hits = 0ull;                       // EMIT: bit q = this lane's q-th column (64 q + lane) is a tight pair of the row
    int q = 0;
    for (int j0 = lane; j0 < n; j0 += 64 * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u, ++q) {
            const int j = j0 + 64 * u;
            if (j < n) {
                const double pj = p[j];
                const float sq = reart_sqdist3(ax, ay, az, tx[j], ty[j], tz[j]);
                const double T = ((EMIT ? cur + tol : v1) - pj) + (fabs(v1) + fabs(pj)) * 8.9e-16;      // (EMIT: v1 <= cur, the seed)
                if (T >= 0.0 && (double)sq <= (T * T) * 1.00000095367431640625) {
                    const double v = (double)sqrtf(sq) + pj;
                    if (EMIT && j != own && v - cur <= tol) hits |= 1ull << q;
                    if (v < v1 || (v == v1 && j < j1)) { v1 = v; j1 = j; }
                }
            }
        }
    }
    if (EMIT) {       // the row's pairs, lowest lane first, into the row's OWN slots (a list shared by all rows behind one atomic
                      // counter serialised a thousand reservations per problem on one address: +35 us per pass at 9 x 1024^2)
        int cnt = 0;
        for (unsigned long long m = __ballot(hits != 0ull); m; m = __ballot(hits != 0ull)) {
            const int l = __ffsll((long long)m) - 1;
            int col = 0;
            if (lane == l) { const int s_ = __ffsll((long long)hits) - 1; hits &= hits - 1ull; col = 64 * s_ + lane; }
            col = __builtin_amdgcn_readlane(col, l);
            if (lane == 0 && cnt < cap) edges[cnt] = col;
            ++cnt;
        }
        if (lane == 0) *n_edges = cnt;
    }
    lap_wave_argmin_fast(v1, j1);
}

// MODE 0: the whole re-solve in one launch.  MODE 1: the sequential part only -- row potentials come from lap_jv_pass_kernel
// (pre_*), the certificate is left to the next two launches (certified[b] = 2: pending).  MODE 2: certificate of a pending
// matrix whose first round (lap_jv_pass_kernel, pass_mode 1) found a violation: the Jacobi rounds from the solve's prices.
template <int BS, bool PTS, int MODE>
__global__ __launch_bounds__(BS) void lap_jv_kernel(JvArgs a) {
    constexpr int JV_CPT = (PTS ? JV_PTS_NMAX : LAP_NMAX) / BS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    const int n = a.n, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = BS / 64;
    double *price = (double *)lsm;            // [n]
    double *u = price + n;                    // [n] row potentials
    int *owner = (int *)(u + n);              // [n] column -> row
    int *assigned = owner + n;                // [n] row -> column
    int *pred = assigned + n;                 // [n] column -> row it was reached from
    int *flist = pred + n;                    // [n] free rows / arg-min columns
    float *psx = (float *)(flist + n);        // PTS: src x|y|z [3][n], tgt x|y|z [3][n]
    float *psy = psx + n, *psz = psy + n, *ptx = psz + n, *pty = ptx + n, *ptz = pty + n;
    __shared__ double s_rv[2][NW];
    __shared__ int s_rj[2][NW];
    __shared__ double s_red[NW];
    __shared__ int s_cnt, s_flag;
    const float *C = PTS ? nullptr : a.cost + (size_t)b * n * n;
    const bool race = MODE == 1 && a.done != nullptr;
    const int racer = race ? (int)blockIdx.y : 0;
    __shared__ int s_lost[2];                  // race: "another racer has published", double-buffered like the steps' slots
    auto lost = [&]() -> int { return __hip_atomic_load(a.done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // c_ij for any (i, j), and the row scan, through one interface
    auto cost_at = [&](int i, int j) -> float {
        return PTS ? sqrtf(reart_sqdist3(psx[i], psy[i], psz[i], ptx[j], pty[j], ptz[j])) : C[(size_t)i * n + j];
    };
    auto row_top2 = [&](int i, const double *pr, double &v1, int &j1, double &v2) {
        if (PTS) lap_row_top2_pts(psx[i], psy[i], psz[i], ptx, pty, ptz, pr, 0, n, lane, v1, j1, v2);
        else lap_row_top2(C + (size_t)i * n, pr, n, lane, v1, j1, v2);
    };
    __shared__ double s_av1[2][NW], s_av2[2][NW];
    __shared__ int s_aj1[2][NW], s_ai0[2][NW];

    JPH_DECL;
    double mx;
    if (PTS) {
        // both point sets into LDS; the tolerances only need the scale of the costs: the diagonal of the clouds' box
        float lo = INFINITY, hi = -INFINITY;
        const float *S_ = a.src + (size_t)b * n * 3, *T_ = a.tgt + (size_t)b * n * 3;
        for (int e = tid; e < 3 * n; e += BS) {
            const float sv = S_[e], tv = T_[e];
            (e % 3 == 0 ? psx : (e % 3 == 1 ? psy : psz))[e / 3] = sv;
            (e % 3 == 0 ? ptx : (e % 3 == 1 ? pty : ptz))[e / 3] = tv;
            lo = fminf(lo, fminf(sv, tv)); hi = fmaxf(hi, fmaxf(sv, tv));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
        mx = 1.7320508 * (double)(hi - lo);
    } else if (MODE == 2) {
        mx = a.scale[b];
    } else {
        mx = (double)lap_matrix_max<BS>(C, (size_t)n * n, tid);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    }
    if (MODE == 2 && (a.certified[b] != 2 || !a.cert_bad[b])) {        // uniform: nothing pending, or the first round was clean
        if (tid == 0 && a.certified[b] == 2) a.certified[b] = 1;
        return;
    }
    if (lane == 0) s_red[wv] = mx;
    for (int j = tid; j < n; j += BS) {
        price[j] = race ? a.price_start[(size_t)b * n + j] : (a.price_in ? a.price_in[(size_t)b * n + j] : 0.0);
        owner[j] = -1;
        const int c = (race ? a.col_start : a.col4row)[(size_t)b * n + j];
        assigned[j] = (c >= 0 && c < n) ? c : -1;
    }
    if (tid == 0) { s_cnt = 0; s_lost[0] = 0; s_lost[1] = 0; }
    __syncthreads();
    mx = 0.0;
    for (int w = 0; w < NW; ++w) mx = fmax(mx, s_red[w]);
    if (!(mx > 0.0)) mx = 1.0;
    const double keep_tol = mx * a.keep_tol;
    int st_freed = 0, st_left = 0, st_steps = 0, st_cert = 0, st_arr = 0;
    bool solved = true;
    if (MODE != 2) {
    // previous pairs: a repeated column keeps its lowest row
    for (int i = tid; i < n; i += BS)
        if (assigned[i] >= 0) atomicMin((unsigned int *)&owner[assigned[i]], (unsigned int)i);   // -1 = 0xffffffff: empty
    __syncthreads();
    for (int i = tid; i < n; i += BS)
        if (assigned[i] >= 0 && owner[assigned[i]] != i) assigned[i] = -1;
    __syncthreads();
    // row potentials under the old prices; pairs that lost their arg-min are released
    if (MODE == 1) {
        for (int i = tid; i < n; i += BS) {
            const double v1 = a.pre_v1[(size_t)b * n + i];
            const int j = assigned[i];
            u[i] = v1; flist[i] = a.pre_j1[(size_t)b * n + i];
            if (j >= 0) {
                const double cur = a.pre_cur[(size_t)b * n + i];          // of the column col4row named: still the row's
                if (cur - v1 > keep_tol) { assigned[i] = -1; owner[j] = -1; }
                else u[i] = cur;
            }
        }
    } else
    for (int i = wv; i < n; i += NW) {
        double v1, v2;
        int j1;
        row_top2(i, price, v1, j1, v2);
        if (lane == 0) {
            const int j = assigned[i];
            u[i] = v1; flist[i] = j1;
            if (j >= 0) {
                const double cur = (double)cost_at(i, j) + price[j];
                if (cur - v1 > keep_tol) { assigned[i] = -1; owner[j] = -1; }
                else u[i] = cur;                      // the kept pair is tight by definition
            }
        }
    }
    __syncthreads();
    // greedy: a free row takes its arg-min column when nobody owns it (lowest row wins)
    for (int i = tid; i < n; i += BS)
        if (assigned[i] < 0) atomicAdd(&s_cnt, 1);
    for (int j = tid; j < n; j += BS) pred[j] = 0x7fffffff;
    __syncthreads();
    st_freed = s_cnt;
    for (int i = tid; i < n; i += BS)
        if (assigned[i] < 0 && owner[flist[i]] < 0) atomicMin(&pred[flist[i]], i);
    __syncthreads();
    for (int i = tid; i < n; i += BS)
        if (assigned[i] < 0 && owner[flist[i]] < 0 && pred[flist[i]] == i) assigned[i] = flist[i];
    __syncthreads();
    for (int i = tid; i < n; i += BS)
        if (assigned[i] >= 0) owner[assigned[i]] = i;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // the rows still free, in ascending order (deterministic)
    for (int i0 = 0; i0 < n; i0 += BS) {
        const int i = i0 + tid;
        const bool fr = i < n && assigned[i] < 0;
        const unsigned long long m = __ballot(fr);
        if (lane == 0) s_rj[0][wv] = __builtin_popcountll(m);
        __syncthreads();
        int off = s_cnt;
        for (int w = 0; w < wv; ++w) off += s_rj[0][w];
        if (fr) flist[off + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < NW; ++w) t += s_rj[0][w]; s_cnt += t; }
        __syncthreads();
    }
    int nfree = s_cnt;
    JPH(6);
    // ---- augmenting row reduction (Jonker-Volgenant): a free row takes its cheapest column at once and pays for it --
    // the column's price rises by the gap to the row's second-cheapest column, which keeps every dual constraint and
    // makes the new pair tight -- and the row it displaces is handled next.  One row scan per step, no search: most of
    // the rows a small change of the costs has released settle here.  Exact ties and the rows left when the step budget
    // runs out go to the path search below.  The chain is sequential.
    // PTS: this thread's columns (target points) in registers, two per packed-fp32 operand: a step is bound by the
    // instructions its waves issue (two waves per SIMD, ~50 per column), and reart_sqdist3 on pairs halves its share
    static_assert(JV_CPT % 2 == 0, "columns per thread come in pairs");
    jv_f2 tcx[JV_CPT / 2], tcy[JV_CPT / 2], tcz[JV_CPT / 2];
#pragma unroll
    for (int k = 0; k < JV_CPT; ++k) {
        const int j = tid + k * BS < n ? tid + k * BS : 0;
        tcx[k >> 1][k & 1] = PTS ? ptx[j] : 0.f; tcy[k >> 1][k & 1] = PTS ? pty[j] : 0.f; tcz[k >> 1][k & 1] = PTS ? ptz[j] : 0.f;
    }
    auto row_costs = [&](int i, float (&rc)[JV_CPT]) {
        if (PTS) {
            const float ax = psx[i], ay = psy[i], az = psz[i];
            const jv_f2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
#pragma unroll
            for (int k = 0; k < JV_CPT / 2; ++k) {
                const jv_f2 dx = ax2 - tcx[k], dy = ay2 - tcy[k], dz = az2 - tcz[k];
                const jv_f2 sq = (dx * dx + dy * dy) + dz * dz;               // reart_sqdist3, two columns at a time
                rc[2 * k] = sqrtf(sq.x); rc[2 * k + 1] = sqrtf(sq.y);
            }
        } else {
            const float *row = C + (size_t)i * n;
#pragma unroll
            for (int k = 0; k < JV_CPT; ++k) rc[k] = row[tid + k * BS < n ? tid + k * BS : 0];
        }
    };
    {
        // thread t holds the columns t, t + BS, ... : coordinates (or the row read), prices and owners in registers.  A
        // step is the row's costs in every thread, a (min, arg-min, second min) reduction that carries the arg-min's owner
        // along, ONE barrier (the waves' triples meet in double-buffered LDS slots), and the update by the arg-min's thread.
        double pr[JV_CPT];
        int own[JV_CPT];
#pragma unroll
        for (int k = 0; k < JV_CPT; ++k) {
            const int j = tid + k * BS;
            pr[k] = j < n ? price[j] : INFINITY;
            own[k] = j < n ? owner[j] : -1;
        }
        int ncur = nfree, budget = JV_ARR_BUDGET * nfree + 64;          // uniform over the workgroup: every thread follows the chain
        int *next = pred;                                   // not needed before the path search
        int par = 0;
        for (int pass = 0; pass < 2 && ncur > 0; ++pass) {
            int nnext = 0;
            for (int k0 = 0; k0 < ncur; ++k0) {
                int i = flist[jv_order(k0, ncur, racer)];
                for (;;) {
                    float rc[JV_CPT];
                    row_costs(i, rc);
                    double v1 = INFINITY, v2 = INFINITY;
                    int j1 = 0x7fffffff, i0 = -1;
#pragma unroll
                    for (int k = 0; k < JV_CPT; ++k) {
                        lap_top2_push((double)rc[k] + pr[k], tid + k * BS, own[k], v1, j1, v2, i0);   // +inf beyond n
                    }
                    lap_wave_top2_fast(v1, j1, v2, i0);
                    if (lane == 0) { s_av1[par][wv] = v1; s_av2[par][wv] = v2; s_aj1[par][wv] = j1; s_ai0[par][wv] = i0; }
                    if (race && tid == 0) s_lost[par] = (budget & 15) == 0 ? lost() : s_lost[par ^ 1];
                    __syncthreads();
                    if (race && s_lost[par]) return;                        // uniform: everybody reads the step's slot
                    v1 = lane < NW ? s_av1[par][lane] : INFINITY; v2 = lane < NW ? s_av2[par][lane] : INFINITY;
                    j1 = lane < NW ? s_aj1[par][lane] : 0x7fffffff; i0 = lane < NW ? s_ai0[par][lane] : -1;
                    lap_lanes_top2<(NW <= 2 ? 1 : (NW <= 4 ? 2 : (NW <= 8 ? 3 : 4)))>(v1, j1, v2, i0);
                    par ^= 1;
                    const bool tie = !(v1 < v2);
                    const bool stop = budget-- <= 0 || (tie && i0 >= 0);
                    if (stop) {
                        if (tid == 0) { next[nnext] = i; u[i] = v1; }
                        ++nnext;
                        break;
                    }
                    if ((j1 & (BS - 1)) == tid) {                            // the arg-min's thread
                        const int kk = j1 / BS;
#pragma unroll
                        for (int k = 0; k < JV_CPT; ++k)
                            if (k == kk) { if (!tie) pr[k] += v2 - v1; own[k] = i; }
                        u[i] = tie ? v1 : v2;
                        assigned[i] = j1;
                        if (i0 >= 0) assigned[i0] = -1;
                    }
                    ++st_steps;
                    if (i0 < 0) break;
                    i = i0;
                }
            }
            __syncthreads();
            for (int k = tid; k < nnext; k += BS) flist[k] = next[k];
            __syncthreads();
            ncur = nnext;
        }
#pragma unroll
        for (int k = 0; k < JV_CPT; ++k) {
            const int j = tid + k * BS;
            if (j < n) { price[j] = pr[k]; owner[j] = own[k]; }
        }
        if (tid == 0) s_cnt = ncur;
    }
    __syncthreads();
    nfree = s_cnt;
    st_left = nfree;
    st_arr = st_steps;
    st_steps = 0;

    JPH(7);
    // ---- one shortest augmenting path per free row
    // The prices of this thread's columns stay in registers through all searches (LDS keeps a copy for the certificate):
    // read from LDS inside the step each one was a latency the step waited out.
    double pj[JV_CPT];
#pragma unroll
    for (int k = 0; k < JV_CPT; ++k) pj[k] = tid + k * BS < n ? price[tid + k * BS] : 0.0;
    for (int f = 0; f < nfree; ++f) {
        const int i0 = flist[jv_order(f, nfree, racer)];
        double d[JV_CPT];
        unsigned scanned = 0u;
        // this thread's unowned columns: among columns at the SAME smallest distance an unowned one ends the search at once
        // (any column at the minimum is a valid Dijkstra pick), so the arg-min's tie key puts them first
        unsigned freecol = 0u;
#pragma unroll
        for (int k = 0; k < JV_CPT; ++k)
            if (tid + k * BS < n && owner[tid + k * BS] < 0) freecol |= 1u << k;
        {
            const double ui = u[i0];
            float rc0[JV_CPT];
            row_costs(i0, rc0);
#pragma unroll
            for (int k = 0; k < JV_CPT; ++k) {
                const int j = tid + k * BS;
                d[k] = j < n ? ((double)rc0[k] + pj[k]) - ui : INFINITY;
                if (j < n) pred[j] = i0;
                if (j >= n) scanned |= 1u << k;
            }
        }
        double mu = 0.0;
        int sink = -1;
        [[maybe_unused]] const int hist0_ = st_steps;
        JPH(5);
        for (int it = 0; ; ++it) {
            // closest unlabelled column: (distance, index) minimum over the workgroup
            double bv = INFINITY;
            int bj = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < JV_CPT; ++k)
                if (!((scanned >> k) & 1u)) {
                    const int key = (tid + k * BS) | (((freecol >> k) & 1u) ? 0 : JV_OWNED);
                    if (d[k] < bv || (d[k] == bv && key < bj)) { bv = d[k]; bj = key; }
                }
            lap_wave_argmin_fast(bv, bj);
            const int par = it & 1;
            if (lane == 0) { s_rv[par][wv] = bv; s_rj[par][wv] = bj; }
            if (race && tid == 0) s_lost[par] = (it & 15) == 0 ? lost() : s_lost[par ^ 1];
            JPH(0);
            __syncthreads();
            JPH(1);
            if (race && s_lost[par]) return;
            // the waves' minima meet in the first NW <= 16 lanes of every wave: four butterfly steps, then a broadcast
            bv = lane < NW ? s_rv[par][lane] : INFINITY; bj = lane < NW ? s_rj[par][lane] : 0x7fffffff;
            lap_lanes_argmin<(NW <= 2 ? 1 : (NW <= 4 ? 2 : (NW <= 8 ? 3 : 4)))>(bv, bj);
            ++st_steps;
            mu = bv;
            const int jstar = bj == 0x7fffffff ? bj : (bj & ~JV_OWNED);
            JPH(2);
            if (jstar == 0x7fffffff || !(bv < INFINITY)) break;   // only with non-finite costs: the matrix is reported uncertified
            if ((jstar & (BS - 1)) == tid) scanned |= 1u << (jstar / BS);
            const int i = owner[jstar];
            if (i < 0) { sink = jstar; break; }
            // the step's costs: every column of this thread in row i (matrix form: the ONE dependent global read of
            // the step, all loads in flight together; points form: no memory beyond LDS at all).  All columns are
            // evaluated, labelled ones included (a search labels ~ 7 % of them), and the relaxation is a select: the
            // thread's columns advance together instead of one conditional block after the other
            float rc[JV_CPT];
            row_costs(i, rc);
            const double ui = u[i];
#pragma unroll
            for (int k = 0; k < JV_CPT; ++k) {
                const double nd = mu + (((double)rc[k] + pj[k]) - ui);
                const bool better = !((scanned >> k) & 1u) && nd < d[k];
                d[k] = better ? nd : d[k];
                if (better) pred[tid + k * BS] = i;
            }
            JPH(3);
        }
        if (sink < 0) { solved = false; break; }
        JPH_HIST(st_steps - hist0_);
        // potentials: fixed columns (all labelled ones except the sink) and their rows
#pragma unroll
        for (int k = 0; k < JV_CPT; ++k) {
            const int j = tid + k * BS;
            if (j < n && ((scanned >> k) & 1u) && j != sink) {
                const double delta = mu - d[k];
                pj[k] += delta;
                price[j] = pj[k];
                u[owner[j]] += delta;
            }
        }
        if (tid == 0) u[i0] += mu;
        __syncthreads();
        if (tid == 0 && sink >= 0) {                           // flip the path
            int j = sink;
            for (;;) {
                const int i = pred[j];
                const int jn = assigned[i];
                assigned[i] = j; owner[j] = i;
                if (i == i0) break;
                j = jn;
            }
        }
        __syncthreads();
    }

    JPH(4);
    if (race) {                                // first to finish publishes; everybody else has nothing to add
        if (tid == 0) s_flag = atomicCAS(a.done + b, 0, racer + 1) == 0;
        __syncthreads();
        if (!s_flag) return;
    }
    }   // MODE != 2
    // ---- certificate (the auction's): Jacobi rounds on the potentials until every assigned column is an exact arg-min
    double *dd = price;
    double *pb = u;                            // scratch: the row potentials are not needed any more
    const double tol = mx * 1e-13;
    int certified = 0;
    for (int round = 0; MODE != 1 && solved && round < a.max_rounds_cert; ++round) {
        if (tid == 0) s_flag = 0;
        ++st_cert;
        __syncthreads();
        for (int i = wv; i < n; i += NW) {
            double v1, v2;
            int j1;
            row_top2(i, dd, v1, j1, v2);
            if (lane == 0) {
                const int j = assigned[i];
                const double cur = (double)cost_at(i, j) + dd[j];
                pb[i] = (cur - v1 > tol) ? v1 - (double)cost_at(i, j) : dd[j];
                if (cur - v1 > tol) s_flag = 1;
            }
        }
        __syncthreads();
        const int changed = s_flag;
        for (int i = tid; i < n; i += BS) dd[assigned[i]] = pb[i];
        __syncthreads();
        if (!changed) { certified = 1; break; }
    }
    for (int i = tid; i < n; i += BS) a.col4row[(size_t)b * n + i] = assigned[i];
    if (a.price_out)
        for (int j = tid; j < n; j += BS) a.price_out[(size_t)b * n + j] = dd[j];
    if (tid == 0) a.certified[b] = MODE == 1 ? (solved ? 2 : 0) : certified;
    if (MODE == 1 && tid == 0) { a.scale[b] = mx; a.cert_bad[b] = 0; }
    JPH(8);
    JPH_FLUSH();
    if (tid == 0 && a.stats) {
        int *o = a.stats + 4 * b;
        if (MODE == 2) o[3] += st_cert;          // on top of the first round the pass kernel ran
        else { o[0] = st_freed + (racer << 16); o[1] = st_left; o[2] = st_steps; o[3] = (MODE == 1 ? 1 : st_cert) + (st_arr << 8); }
    }
}

// The two full passes over the costs of a re-solve -- every row's (min, arg-min) of c_ik + p_k, before (row potentials:
// which previous pairs are still tight) and after (first certificate round) the sequential part -- on the whole chip:
// gridDim.y workgroups per matrix, one wave per row.  The sequential part runs one workgroup per matrix (T-1 = 19 of 256
// compute units), where each pass cost 8 % of the kernel.  Same scan functions, so the same bits as the one-launch form.
#define JV_PASS_BS 256
template <bool PTS>
__global__ __launch_bounds__(JV_PASS_BS) void lap_jv_pass_kernel(JvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    const int n = a.n, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (a.pass_mode == 1 && a.certified[b] != 2) return;
    double *price = (double *)lsm;
    float *ptx = (float *)(price + n), *pty = ptx + n, *ptz = pty + n;
    const double *pin = a.pass_mode == 0 ? a.price_in : a.price_out;
    const float *S_ = PTS ? a.src + (size_t)b * n * 3 : nullptr, *T_ = PTS ? a.tgt + (size_t)b * n * 3 : nullptr;
    const float *C = PTS ? nullptr : a.cost + (size_t)b * n * n;
    for (int j = tid; j < n; j += JV_PASS_BS) {
        price[j] = pin ? pin[(size_t)b * n + j] : 0.0;
        if (PTS) { ptx[j] = T_[3 * j]; pty[j] = T_[3 * j + 1]; ptz[j] = T_[3 * j + 2]; }
    }
    __syncthreads();
    const double tol = a.pass_mode == 1 ? a.scale[b] * 1e-13 : 0.0;
    for (int i = blockIdx.y * (JV_PASS_BS / 64) + wv; i < n; i += gridDim.y * (JV_PASS_BS / 64)) {
        double v1, v2;
        int j1;
        float ax = 0.f, ay = 0.f, az = 0.f;
        const int c = a.col4row[(size_t)b * n + i];
        const bool has = c >= 0 && c < n;
        double cur = 0.0;
        if (PTS) {
            // only the minimum and its column are kept of a pass (pre_v1 / pre_j1; the certificate compares v1 with cur): the
            // scan starts from the row's own column and skips what cannot beat it (lap_row_min_pts_seeded: same v1, j1)
            ax = S_[3 * i]; ay = S_[3 * i + 1]; az = S_[3 * i + 2];
            if (has) cur = (double)sqrtf(reart_sqdist3(ax, ay, az, ptx[c], pty[c], ptz[c])) + price[c];
            v1 = has ? cur : INFINITY; j1 = has ? c : 0x7fffffff;
            lap_row_min_pts_seeded(ax, ay, az, ptx, pty, ptz, price, n, lane, v1, j1);
        } else {
            lap_row_top2(C + (size_t)i * n, price, n, lane, v1, j1, v2);
            if (has) cur = (double)C[(size_t)i * n + c] + price[c];
        }
        if (lane == 0) {
            if (a.pass_mode == 0) {
                a.pre_v1[(size_t)b * n + i] = v1; a.pre_j1[(size_t)b * n + i] = j1; a.pre_cur[(size_t)b * n + i] = cur;
            } else if (!has || cur - v1 > tol) atomicOr(&a.cert_bad[b], 1);
        }
    }
}

// The points form of the passes, as it is launched since round 6.  The generic kernel above spent its 61 us per pass (9 x 2048^2)
// WAITING, not computing: eight serial round trips of its staging loop, then one more per row for the row's point and column,
// with eight waves on a compute unit to hide them -- taking the square roots out of its scan changed nothing.  Here: one
// workgroup of 16 waves per compute unit (one staged copy of the targets and potentials serves 16 waves); the staging loop's
// loads are all in flight before the first LDS store; a wave fetches the points and columns of ALL its rows at once (lane k holds
// row k) and reads them back with v_readlane.  Same scan (lap_row_min_pts_seeded), same results.
#define JV_PASSP_BS 1024
__global__ __launch_bounds__(JV_PASSP_BS) void lap_jv_pass_pts_kernel(JvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    constexpr int NW = JV_PASSP_BS / 64, NJ = (JV_PTS_NMAX + JV_PASSP_BS - 1) / JV_PASSP_BS;
    const int n = a.n, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (a.pass_mode == 1 && a.certified[b] != 2) return;
    double *price = (double *)lsm;
    float *ptx = (float *)(price + n), *pty = ptx + n, *ptz = pty + n;
    const double *pin = a.pass_mode == 0 ? a.price_in : a.price_out;
    const float *S_ = a.src + (size_t)b * n * 3, *T_ = a.tgt + (size_t)b * n * 3;
    {
        double pv[NJ];
        float vx[NJ], vy[NJ], vz[NJ];
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const int j = tid + k * JV_PASSP_BS, jj = j < n ? j : 0;
            pv[k] = pin ? pin[(size_t)b * n + jj] : 0.0;
            vx[k] = T_[3 * jj]; vy[k] = T_[3 * jj + 1]; vz[k] = T_[3 * jj + 2];
        }
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const int j = tid + k * JV_PASSP_BS;
            if (j < n) { price[j] = pv[k]; ptx[j] = vx[k]; pty[j] = vy[k]; ptz[j] = vz[k]; }
        }
    }
    // this wave's rows: first, first + stride, ...; lane k fetches row k of every block of 64 of them
    const int first = blockIdx.y * NW + wv, stride = gridDim.y * NW;
    const int nrows = first < n ? (n - first + stride - 1) / stride : 0;
    const double tol = a.pass_mode == 1 ? a.scale[b] * 1e-13 : 0.0;
    for (int k0 = 0; k0 < nrows; k0 += 64) {
        const int il = first + (k0 + lane) * stride, ic = il < n ? il : 0;
        const float lx = S_[3 * ic], ly = S_[3 * ic + 1], lz = S_[3 * ic + 2];
        const int lc = a.col4row[(size_t)b * n + ic];
        if (k0 == 0) __syncthreads();                    // (the staged copies; the rows' loads are in flight across it)
        const int kn = nrows - k0 < 64 ? nrows - k0 : 64;
        for (int k = 0; k < kn; ++k) {
            const int i = first + (k0 + k) * stride;
            const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lx), k));
            const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ly), k));
            const float az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lz), k));
            const int c = __builtin_amdgcn_readlane(lc, k);
            const bool has = c >= 0 && c < n;
            double cur = 0.0;
            if (has) cur = (double)sqrtf(reart_sqdist3(ax, ay, az, ptx[c], pty[c], ptz[c])) + price[c];
            double v1 = has ? cur : INFINITY;
            int j1 = has ? c : 0x7fffffff;
            if (a.pass_mode == 1 && a.tie_edges && has)
                lap_row_min_pts_seeded<true>(ax, ay, az, ptx, pty, ptz, price, n, lane, v1, j1, i, c, cur, tol,
                                             a.tie_edges + ((size_t)b * n + i) * a.tie_cap, a.tie_n + (size_t)b * n + i, a.tie_cap);
            else {
                lap_row_min_pts_seeded(ax, ay, az, ptx, pty, ptz, price, n, lane, v1, j1);
                if (a.pass_mode == 1 && a.tie_edges && lane == 0) a.tie_n[(size_t)b * n + i] = 0;
            }
            if (lane == 0) {
                if (a.pass_mode == 0) {
                    a.pre_v1[(size_t)b * n + i] = v1; a.pre_j1[(size_t)b * n + i] = j1; a.pre_cur[(size_t)b * n + i] = cur;
                } else if (!has || cur - v1 > tol) atomicOr(&a.cert_bad[b], 1);
            }
        }
    }
    if (nrows == 0) __syncthreads();                     // (every wave of the workgroup meets the barrier above exactly once)
}

// Re-solve from the assignment in col4row and the potentials in price_in (both from an earlier solve of a similar batch,
// reart_lap_auction* or these functions); same outputs and the same certificate as reart_lap_auction.
// the state arrays of the many-compute-unit row reduction, behind the race layout
static size_t jv_mc_extra_bytes(int B, int n) {
    return reart_align_up(sizeof(double) * (size_t)B * n, 256) + 6 * reart_align_up(sizeof(int) * (size_t)B * n, 256) +
           reart_align_up(sizeof(int) * 8 * ((size_t)B + 1), 256);      // (+ one block of launch-wide counters behind the problems')
}
extern "C" size_t reart_lap_mc_workspace_bytes(int B, int n, int racers) {
    const size_t r = reart_lap_race_workspace_bytes(B, n, racers);
    return r ? r + jv_mc_extra_bytes(B, n) : 0;
}

// per_wave: 0 = one row at a time (lap_jv_kernel), 1 = row reduction one chain per wave (lap_mw.hip), 2 = row reduction on
// arr_wgs workgroups per problem (lap_mw.hip, three launches)
template <bool PTS>
static int jv_launch(JvArgs a, void *workspace, size_t workspace_bytes, void *stream, int racers = 1, int per_wave = 0, int arr_wgs = 0,
                     int *tie_out = nullptr) {
    if (a.B < 0 || a.n < 1 || a.n > (PTS ? JV_PTS_NMAX : LAP_NMAX)) return REART_ERR_INVALID_ARG;
    if (racers < 1 || racers > JV_RACE_MAX) return REART_ERR_INVALID_ARG;
    if (a.B == 0) return REART_OK;
    if (!a.col4row || !a.certified || !a.price_in) return REART_ERR_INVALID_ARG;
    if (per_wave && (!PTS || a.n < JV_SPLIT_NMIN || a.n > reart_internal_jvmw_nmax())) return REART_ERR_UNSUPPORTED;
    if (a.n < JV_SPLIT_NMIN) racers = 1;       // one launch does everything there: nothing worth racing
    if (!workspace || workspace_bytes < (racers > 1 ? reart_lap_race_workspace_bytes(a.B, a.n, racers) : reart_lap_workspace_bytes(a.B, a.n)))
        return REART_ERR_INVALID_ARG;
    if (!a.price_out) a.price_out = (double *)workspace;
    a.max_rounds_cert = 4 * a.n; a.keep_tol = 1e-12;
    a.stats = (int *)((char *)workspace + reart_align_up(sizeof(double) * (size_t)a.B * a.n, 256));
    const size_t lds = (size_t)a.n * (2 * 8 + 4 * 4 + (PTS ? 6 * 4 : 0));
    constexpr int JVBS = PTS ? JV_PTS_BS : 256;
    if (lds > REART_LDS_DEFAULT_CAP &&
        (hipFuncSetAttribute((const void *)lap_jv_kernel<JVBS, PTS, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess ||
         hipFuncSetAttribute((const void *)lap_jv_kernel<JVBS, PTS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess ||
         hipFuncSetAttribute((const void *)lap_jv_kernel<JVBS, PTS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess))
        return REART_ERR_LAUNCH;
    if (a.n < JV_SPLIT_NMIN) {      // small problems: one launch, the passes are a few microseconds
        hipLaunchKernelGGL((lap_jv_kernel<JVBS, PTS, 0>), dim3(a.B), dim3(JVBS), lds, (hipStream_t)stream, a);
        REART_CHECK_LAUNCH();
        return REART_OK;
    }
    {
        const size_t bn = reart_align_up(sizeof(double) * (size_t)a.B * a.n, 256);
        char *w = (char *)a.stats + reart_align_up(sizeof(int) * 4 * (size_t)a.B, 256) + bn;      // past the auction's row bids
        a.pre_v1 = (double *)w; w += bn;
        a.pre_cur = (double *)w; w += bn;
        a.pre_j1 = (int *)w; w += reart_align_up(sizeof(int) * (size_t)a.B * a.n, 256);
        a.scale = (double *)w; w += reart_align_up(sizeof(double) * (size_t)a.B, 256);
        a.cert_bad = (int *)w;
    }
    int per = PTS ? (256 + a.B - 1) / a.B : (2 * 256 + a.B - 1) / a.B;   // workgroups per matrix: one of 16 waves (points) / two of 4 per compute unit over the batch
    const int pass_waves = (PTS ? JV_PASSP_BS : JV_PASS_BS) / 64;
    const int per_max = (a.n + pass_waves - 1) / pass_waves;
    per = per < 1 ? 1 : (per > per_max ? per_max : per);
    const size_t lds_pass = (size_t)a.n * (8 + (PTS ? 12 : 0));
    auto launch_pass = [&]() {
        if (PTS) hipLaunchKernelGGL(lap_jv_pass_pts_kernel, dim3(a.B, per), dim3(JV_PASSP_BS), lds_pass, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((lap_jv_pass_kernel<false>), dim3(a.B, per), dim3(JV_PASS_BS), lds_pass, (hipStream_t)stream, a);
    };
    if (lds_pass > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute(PTS ? (const void *)lap_jv_pass_pts_kernel : (const void *)lap_jv_pass_kernel<false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pass) != hipSuccess)
        return REART_ERR_LAUNCH;
    a.pass_mode = 0;
    launch_pass();
    REART_CHECK_LAUNCH();
    if (racers > 1) {
        // the racers' common start: copies of the assignment and the potentials (the winner overwrites the originals while
        // a late racer may still be loading), and the flags they meet in -- in the race layout's per-racer arrays
        const size_t bn = reart_align_up(sizeof(double) * (size_t)a.B * a.n, 256);
        char *w = (char *)workspace + reart_lap_workspace_bytes(a.B, a.n);
        double *pc = (double *)w;
        int *cc = (int *)(w + bn);
        a.done = (int *)(w + bn * (size_t)racers);
        // (the form of lap_mw.hip with the row reduction on many compute units hands its racers the state of that launch: they
        // read neither copy, and its set-up launch clears the flags -- three launches less per refresh)
        if (per_wave != 2 &&
            (hipMemcpyAsync(pc, a.price_in, sizeof(double) * (size_t)a.B * a.n, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess ||
             hipMemcpyAsync(cc, a.col4row, sizeof(int) * (size_t)a.B * a.n, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess ||
             hipMemsetAsync(a.done, 0, sizeof(int) * (size_t)a.B, (hipStream_t)stream) != hipSuccess))
            return REART_ERR_LAUNCH;
        a.price_start = pc; a.col_start = cc;
    }
    if (per_wave == 2) {                        // lap_mw.hip: set-up | row reduction on arr_wgs workgroups per problem | searches
        if (workspace_bytes < reart_lap_mc_workspace_bytes(a.B, a.n, racers)) return REART_ERR_INVALID_ARG;
        char *w = (char *)workspace + reart_lap_race_workspace_bytes(a.B, a.n, racers);
        const size_t bi = reart_align_up(sizeof(int) * (size_t)a.B * a.n, 256);
        a.mc_price = (double *)w; w += reart_align_up(sizeof(double) * (size_t)a.B * a.n, 256);
        a.mc_owner = (int *)w; w += bi; a.mc_assigned = (int *)w; w += bi; a.mc_list = (int *)w; w += bi; a.mc_next = (int *)w; w += bi;
        a.mc_tree = (int *)w; w += bi; a.mc_tpar = (int *)w; w += bi;
        a.mc_cnt = (int *)w;
        const int rc = reart_internal_jvmc_launch(a, racers, arr_wgs, (hipStream_t)stream);
        if (rc != REART_OK) return rc;
    } else if (per_wave) {                      // lap_mw.hip: the row reduction one chain per wave
        const int rc = reart_internal_jvmw_launch(a, racers, (hipStream_t)stream);
        if (rc != REART_OK) return rc;
    } else {
        hipLaunchKernelGGL((lap_jv_kernel<JVBS, PTS, 1>), dim3(a.B, racers), dim3(JVBS), lds, (hipStream_t)stream, a);
        REART_CHECK_LAUNCH();
    }
    a.done = nullptr;
    a.pass_mode = 1;
    launch_pass();
    REART_CHECK_LAUNCH();
    a.price_in = a.price_out;                                  // the certificate continues from the solve's potentials
    hipLaunchKernelGGL((lap_jv_kernel<JVBS, PTS, 2>), dim3(a.B), dim3(JVBS), lds, (hipStream_t)stream, a);
    REART_CHECK_LAUNCH();
    if (PTS && a.tie_edges && tie_out)                         // the cycle check over the pairs the pass listed (cert_bad: the rounds above moved the potentials)
        return reart_internal_tie_cycles(a.B, a.n, a.col4row, tie_out, a.tie_edges, a.tie_n, a.tie_cap, a.cert_bad, (hipStream_t)stream);
    return REART_OK;
}

extern "C" int reart_lap_resolve(const float *cost, int B, int n, int32_t *col4row, int32_t *certified,
                                 const double *price_in, double *price_out, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    if (!cost && B > 0) return REART_ERR_INVALID_ARG;
    JvArgs a = {};
    a.cost = cost; a.B = B; a.n = n; a.col4row = col4row; a.certified = certified; a.price_in = price_in; a.price_out = price_out;
    return jv_launch<false>(a, workspace, workspace_bytes, stream);
}

// The same re-solve for Euclidean costs between two point sets WITHOUT a cost matrix: c_ij = the value reart_cdist(src,
// tgt) would hold, bit for bit, recomputed from copies of both sets in LDS wherever a cost is needed -- a Dijkstra step
// then touches no memory beyond LDS (with a matrix it is one dependent 4n-byte row read from HBM: the matrices of a batch
// do not fit any cache).  n <= 2048 (both sets + the solver state in 160 KB of LDS).
extern "C" int reart_lap_resolve_points(const float *src, const float *tgt, int B, int n, int32_t *col4row,
                                        int32_t *certified, const double *price_in, double *price_out, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    if ((!src || !tgt) && B > 0) return REART_ERR_INVALID_ARG;
    JvArgs a = {};
    a.src = src; a.tgt = tgt; a.B = B; a.n = n; a.col4row = col4row; a.certified = certified; a.price_in = price_in;
    a.price_out = price_out;
    return jv_launch<true>(a, workspace, workspace_bytes, stream);
}

// reart_lap_resolve_points with `racers` (2..13) workgroups per problem on otherwise idle compute units: the same exact
// algorithm taking the free rows in different orders, first to finish publishes (see jv_order).  The assignment is the
// optimum either way; the potentials (and, among optima of exactly equal cost, the assignment) are the winner's, so they
// are not reproducible run to run.  workspace: reart_lap_race_workspace_bytes(B, n, racers).  n < 512: the plain re-solve.
extern "C" int reart_lap_resolve_points_race(const float *src, const float *tgt, int B, int n, int racers, int32_t *col4row,
                                             int32_t *certified, const double *price_in, double *price_out, void *workspace,
                                             size_t workspace_bytes, void *stream) {
    if ((!src || !tgt) && B > 0) return REART_ERR_INVALID_ARG;
    if (racers < 2) return REART_ERR_INVALID_ARG;
    JvArgs a = {};
    a.src = src; a.tgt = tgt; a.B = B; a.n = n; a.col4row = col4row; a.certified = certified; a.price_in = price_in;
    a.price_out = price_out;
    return jv_launch<true>(a, workspace, workspace_bytes, stream, racers);
}

// reart_lap_resolve_points[_race] with the sequential part run by lap_mw.hip: every WAVE of a problem's workgroup follows its
// own free row (row-reduction chain, then shortest augmenting path) on the columns it holds in registers, and commits under
// a workgroup lock after checking that the columns it is about to write are as it saw them.  512 <= n <= 2048
// (REART_ERR_UNSUPPORTED otherwise: use reart_lap_resolve_points_race).  racers >= 1 workgroups per problem (1: no race;
// workspace reart_lap_race_workspace_bytes(B, n, racers) either way).  Same outputs and certificate; like the raced
// solves the potentials (and, among optima of exactly equal cost, the assignment) depend on timing.  stats [B][4] in the
// workspace: released rows (+ winner << 16), rows left for the path search | conflicts << 16, path-search steps,
// 1 + 256 * row-reduction steps.
extern "C" int reart_lap_resolve_points_mw(const float *src, const float *tgt, int B, int n, int racers, int32_t *col4row,
                                           int32_t *certified, const double *price_in, double *price_out, void *workspace,
                                           size_t workspace_bytes, void *stream) {
    if ((!src || !tgt) && B > 0) return REART_ERR_INVALID_ARG;
    if (racers < 1) return REART_ERR_INVALID_ARG;
    if (workspace_bytes < reart_lap_race_workspace_bytes(B, n, racers)) return REART_ERR_INVALID_ARG;
    JvArgs a = {};
    a.src = src; a.tgt = tgt; a.B = B; a.n = n; a.col4row = col4row; a.certified = certified; a.price_in = price_in;
    a.price_out = price_out;
    return jv_launch<true>(a, workspace, workspace_bytes, stream, racers, 1);
}

// reart_lap_resolve_points_mw with the row reduction of every problem spread over `arr_wgs` workgroups (lap_mc_arr_kernel:
// lock-free commits on state in memory), the path searches then one workgroup per problem and racer as before.  workspace:
// reart_lap_mc_workspace_bytes(B, n, racers).  512 <= n <= 2048.
extern "C" int reart_lap_resolve_points_mc(const float *src, const float *tgt, int B, int n, int racers, int arr_wgs, int32_t *col4row,
                                           int32_t *certified, const double *price_in, double *price_out, void *workspace,
                                           size_t workspace_bytes, void *stream) {
    if ((!src || !tgt) && B > 0) return REART_ERR_INVALID_ARG;
    if (racers < 1 || arr_wgs < 1 || arr_wgs > 256) return REART_ERR_INVALID_ARG;
    if (workspace_bytes < reart_lap_mc_workspace_bytes(B, n, racers)) return REART_ERR_INVALID_ARG;
    JvArgs a = {};
    a.src = src; a.tgt = tgt; a.B = B; a.n = n; a.col4row = col4row; a.certified = certified; a.price_in = price_in;
    a.price_out = price_out;
    return jv_launch<true>(a, workspace, workspace_bytes, stream, racers, 2, arr_wgs);
}

extern "C" int reart_lap_resolve_points_mc_ties(const float *src, const float *tgt, int B, int n, int racers, int arr_wgs, int32_t *col4row,
                                                int32_t *certified, const double *price_in, double *price_out, int32_t *tie, int32_t *edges,
                                                int32_t *n_edges, int cap, void *workspace, size_t workspace_bytes, void *stream) {
    if ((!src || !tgt) && B > 0) return REART_ERR_INVALID_ARG;
    if (racers < 1 || arr_wgs < 1 || arr_wgs > 256 || cap < 1 || cap > 32 || ((!tie || !edges || !n_edges) && B > 0)) return REART_ERR_INVALID_ARG;
    if (workspace_bytes < reart_lap_mc_workspace_bytes(B, n, racers)) return REART_ERR_INVALID_ARG;
    if (n < JV_SPLIT_NMIN) return REART_ERR_UNSUPPORTED;       // (below the three-launch form there is no separate certificate pass)
    JvArgs a = {};
    a.src = src; a.tgt = tgt; a.B = B; a.n = n; a.col4row = col4row; a.certified = certified; a.price_in = price_in;
    a.price_out = price_out;
    a.tie_edges = edges; a.tie_n = n_edges; a.tie_cap = cap;
    return jv_launch<true>(a, workspace, workspace_bytes, stream, racers, 2, arr_wgs, tie);
}

// ------------------------------------------------------------------------------------------------------------
// Measurement aid (bench.py's latency roofline of the re-solve, tools/lap_floor.py): the FLOOR of one path-search step of
// lap_jv_kernel<512, points, .> -- what is left of a Dijkstra step when its costs are free: every thread's minimum over its
// JV_PTS_NMAX / 512 = 4 labels in registers, the wave arg-min (lap_wave_argmin_fast), one LDS slot per wave, ONE barrier,
// the meeting of the waves' results in the first lanes (lap_lanes_argmin), the winner's thread marking its column.  The
// same primitives in the same order as the loop above, nothing else: no row costs, no square roots, no relaxation.  A
// re-solve is a sequential chain of such steps per problem, so steps x this floor bounds its duration from below.
template <int BS>
__global__ __launch_bounds__(BS) void lap_step_floor_kernel(int n, int steps, unsigned long long *__restrict__ ticks,
                                                            double *__restrict__ sink) {
    constexpr int NW = BS / 64, CPT = JV_PTS_NMAX / BS;
    __shared__ double s_rv[2][NW];
    __shared__ int s_rj[2][NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double d[CPT];
    unsigned scanned = 0u;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const unsigned j = (unsigned)(tid + k * BS);
        unsigned h = (j + 1u + 7919u * blockIdx.x) * 2654435761u;           // distinct pseudo-random positive labels
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        d[k] = (int)j < n ? 1.0 + (double)h * (1.0 / 4294967296.0) : INFINITY;
        if ((int)j >= n) scanned |= 1u << k;
    }
    const unsigned dead = scanned;
    double acc = 0.0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < steps; ++it) {
        double bv = INFINITY;
        int bj = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < CPT; ++k)
            if (!((scanned >> k) & 1u)) {
                const int key = tid + k * BS;
                if (d[k] < bv || (d[k] == bv && key < bj)) { bv = d[k]; bj = key; }
            }
        lap_wave_argmin_fast(bv, bj);
        const int par = it & 1;
        if (lane == 0) { s_rv[par][wv] = bv; s_rj[par][wv] = bj; }
        __syncthreads();
        bv = lane < NW ? s_rv[par][lane] : INFINITY; bj = lane < NW ? s_rj[par][lane] : 0x7fffffff;
        lap_lanes_argmin<(NW <= 2 ? 1 : (NW <= 4 ? 2 : (NW <= 8 ? 3 : 4)))>(bv, bj);
        if (bj == 0x7fffffff) { scanned = dead; continue; }                 // every column labelled: start over
        acc += bv;
        if ((bj & (BS - 1)) == tid) scanned |= 1u << (bj / BS);
    }
    const unsigned long long t1 = wall_clock64();
    if (tid == 0) { ticks[blockIdx.x] = t1 - t0; sink[blockIdx.x] = acc; }
}

extern "C" int reart_lap_step_floor(int B, int n, int steps, void *workspace, size_t workspace_bytes, double *h_us_per_step,
                                    void *stream) {
    if (B < 1 || B > 4096 || n < 1 || n > JV_PTS_NMAX || steps < 1 || !h_us_per_step) return REART_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < 16 * (size_t)B) return REART_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *ticks = (unsigned long long *)workspace;
    hipLaunchKernelGGL(lap_step_floor_kernel<512>, dim3(B), dim3(512), 0, st, n, steps, ticks, (double *)(ticks + B));
    REART_CHECK_LAUNCH();
    unsigned long long h[4096];
    if (hipMemcpyAsync(h, ticks, sizeof(unsigned long long) * B, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return REART_ERR_LAUNCH;
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        return REART_ERR_LAUNCH;
    unsigned long long worst = 0;
    for (int b = 0; b < B; ++b) worst = h[b] > worst ? h[b] : worst;
    *h_us_per_step = (double)worst / (1e3 * (double)khz) * 1e6 / (double)steps;    // slowest workgroup, like the solve
    return REART_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Cost matrices of the assignment loss / assignment error: Euclidean distances between two batches of points
// (`torch.cdist(pc_src, pc_tgt)` at run_robot.py:171 and utils/model_utils.py:93).  Direct differences in fp32,
// ((dx*dx)+(dy*dy))+(dz*dz) then sqrt -- the library-wide distance contract -- and one pass over the output, which is
// all the traffic there is (B*n*m*4 bytes; torch's matmul-based cdist took 17 ms for 19 x 1024 x 1024 here).
__global__ __launch_bounds__(256) void cdist_kernel(const float *__restrict__ a, const float *__restrict__ b, int n, int m,
                                                    float *__restrict__ out) {
    const int bi = blockIdx.z, i = blockIdx.y;
    const float *pa = a + ((size_t)bi * n + i) * 3;
    const float ax = pa[0], ay = pa[1], az = pa[2];
    const float *pb = b + (size_t)bi * m * 3;
    float *o = out + ((size_t)bi * n + i) * m;
    for (int j = (blockIdx.x * 256 + threadIdx.x) * 4; j < m; j += gridDim.x * 256 * 4) {
        float r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int jj = j + c < m ? j + c : m - 1;
            r[c] = sqrtf(reart_sqdist3(ax, ay, az, pb[3 * jj], pb[3 * jj + 1], pb[3 * jj + 2]));
        }
        if (j + 3 < m && (((size_t)(o + j)) & 15) == 0) *(float4 *)(o + j) = make_float4(r[0], r[1], r[2], r[3]);
        else
            for (int c = 0; c < 4 && j + c < m; ++c) o[j + c] = r[c];
    }
}

extern "C" int reart_cdist(const float *a, const float *b, int B, int n, int m, float *out, void *stream) {
    if (B < 0 || n < 0 || m < 0 || n > 65535 || B > 65535) return REART_ERR_INVALID_ARG;
    if (B == 0 || n == 0 || m == 0) return REART_OK;
    if (!a || !b || !out) return REART_ERR_INVALID_ARG;
    const int gx = (m + 1023) / 1024 < 4 ? (m + 1023) / 1024 : 4;
    hipLaunchKernelGGL(cdist_kernel, dim3(gx, n, B), dim3(256), 0, (hipStream_t)stream, a, b, n, m, out);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
