// reart_amd/csrc/ties.hip -- is the optimal assignment a re-solve returned the ONLY optimal one?
//
// The reference's refresh is `scipy.optimize.linear_sum_assignment` on the cost matrix (run_robot.py:172-176,
// utils/model_utils.py:85-103): a pure function of the matrix, so two runs under one --manual_seed (run_robot.py:37-49) are
// the same run.  The solvers of lap.hip / lap_mw.hip race (orders of free rows, lock-free chains): all of them end in an
// optimal assignment, but when several assignments are optimal -- two rows whose costs to each other's columns tie to the
// last bit of the fp32 costs happen about once in a few hundred re-solves of the kinematic projection -- which one depends on
// who won.  This file finds those cases so that the host can make the choice canonical (reart_amd/utils/lap.py:
// canonical_among_ties).
//
// With optimal potentials p (columns) and u_i = c_i,s(i) + p_s(i) (rows), EVERY optimal assignment uses only tight pairs,
// r_ij = c_ij + p_j - u_i = 0 (complementary slackness), and every perfect matching of the tight pairs is optimal: the set of
// optima is the set of perfect matchings of the tight graph whatever optimal potentials it was drawn with.  The returned
// optimum s is the only one iff the directed graph on rows, i -> owner(j) for every tight pair (i, j) with j != s(i), has no
// cycle (an alternating cycle IS another perfect matching of the tight pairs).
//   lap_tie_edges_kernel  whole chip, one wave per four rows: every row's tight pairs off the assignment (into the row's own K
//                         slots), with the costs' own expression and the certificate's tolerance (lap.hip: `cur - v1 > tol`,
//                         tol = 1e-13 of the cost scale).  The in-place re-solve of the loops lists them in its own
//                         certificate pass instead (lap.hip: lap_row_min_pts_seeded<true>, reart_lap_resolve_points_mc_ties)
//   lap_tie_cycle_kernel  one workgroup per problem: does that graph hold a cycle?  Chains of rows with one tight pair out
//                         of them are contracted by pointer jumping, the rows with several are peeled layer by layer
//                         (see there).  No cycle (the normal case): tie[b] = 0.
#include "common.h"
#include "lap_dev.h"

#define TIE_PASS_BS 1024
#define TIE_CYC_BS 1024

struct TieArgs {
    const float *src, *tgt;        // [B][n][3]
    int B, n;
    const int *col4row;            // [B][n] the optimum
    const double *price;           // [B][n] its column potentials (the solvers' sign convention: a row minimises c + p)
    int *tie;                      // [B] out
    int *cols;                     // [B][n][K] out: the columns of every row's tight pairs off the assignment (the first K of them)
    int *cnt;                      // [B][n] out: how many tight pairs the row has (may exceed K)
    int K;
    const int *stale;              // nullable [B]: the potentials moved after the pairs were listed (tie = 2: the host looks itself)
    int fresh;                     // the cycle kernel is the only writer of tie (no pair-listing launch in front of it): it writes every
                                   // problem's flag itself, zero included -- no memset launch
};

#define TIE_ROWS 4                 // rows a wave tests per pass over the columns (one set of LDS reads serves all of them)
// One workgroup of 16 waves per compute unit; the staging loads are all in flight before the first LDS store, and a wave fetches
// the points and columns of all its rows at once (lane k holds row k): as four waves per workgroup with a load per row the
// kernel spent 100 us waiting for round trips whatever its scan cost (the same finding as lap_jv_pass_pts_kernel, lap.hip).
// A row's pairs go to the row's OWN K slots (and its count to the row's own counter): a list shared by the problem's rows
// behind one atomic counter serialised a thousand reservations per problem on one address (+35 us per pass at 9 x 1024^2).
__global__ __launch_bounds__(TIE_PASS_BS) void lap_tie_edges_kernel(TieArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    constexpr int NW = TIE_PASS_BS / 64, NJ = (LAP_NMAX + TIE_PASS_BS - 1) / TIE_PASS_BS;
    const int n = a.n, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *price = (double *)lsm;
    float *ptx = (float *)(price + n), *pty = ptx + n, *ptz = pty + n, *pf = ptz + n;
    __shared__ float s_lo[NW], s_hi[NW], s_pm[NW];
    const float *S_ = a.src + (size_t)b * n * 3, *T_ = a.tgt + (size_t)b * n * 3;
    // (the fp32 copies of the potentials are taken relative to the problem's first: only differences of potentials enter the
    // test, and a run's potentials drift -- thousands of re-solves each lower some -- until their magnitude, not their spread,
    // would set the margin)
    const double pref = a.price[(size_t)b * n];
    // the cost scale of the solvers' tolerances (lap.hip, points form): the diagonal of the clouds' common box
    float lo = INFINITY, hi = -INFINITY, pm = 0.f;
    {
        double pv[NJ];
        float vx[NJ], vy[NJ], vz[NJ], wx[NJ], wy[NJ], wz[NJ];
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const int j = tid + k * TIE_PASS_BS, jj = j < n ? j : 0;
            pv[k] = a.price[(size_t)b * n + jj];
            vx[k] = T_[3 * jj]; vy[k] = T_[3 * jj + 1]; vz[k] = T_[3 * jj + 2];
            wx[k] = S_[3 * jj]; wy[k] = S_[3 * jj + 1]; wz[k] = S_[3 * jj + 2];
        }
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            const int j = tid + k * TIE_PASS_BS;
            if (j < n) {
                price[j] = pv[k]; pf[j] = (float)(pv[k] - pref);
                ptx[j] = vx[k]; pty[j] = vy[k]; ptz[j] = vz[k];
                pm = fmaxf(pm, fabsf((float)(pv[k] - pref)));
                lo = fminf(lo, fminf(fminf(fminf(vx[k], vy[k]), vz[k]), fminf(fminf(wx[k], wy[k]), wz[k])));
                hi = fmaxf(hi, fmaxf(fmaxf(fmaxf(vx[k], vy[k]), vz[k]), fmaxf(fmaxf(wx[k], wy[k]), wz[k])));
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); pm = fmaxf(pm, __shfl_xor(pm, o, 64));
    }
    if (lane == 0) { s_lo[wv] = lo; s_hi[wv] = hi; s_pm[wv] = pm; }
    // this wave's rows, TIE_ROWS consecutive ones per step: block t of the wave starts at row (first + t * stride) * TIE_ROWS;
    // lane k fetches row k of the wave's sequence (their loads are in flight across the barrier)
    const int first = blockIdx.y * NW + wv, stride = gridDim.y * NW;
    const int nblk_all = (n + TIE_ROWS - 1) / TIE_ROWS;
    const int nblk = first < nblk_all ? (nblk_all - first + stride - 1) / stride : 0;
    auto row_of = [&](int q) { return (first + (q / TIE_ROWS) * stride) * TIE_ROWS + q % TIE_ROWS; };      // q-th row of the wave
    int q0 = 0;
    int il = row_of(lane);
    int ic = il < n ? il : n - 1;
    float lx = S_[3 * ic], ly = S_[3 * ic + 1], lz = S_[3 * ic + 2];
    int lc = a.col4row[(size_t)b * n + ic];
    __syncthreads();
    for (int w = 0; w < NW; ++w) { lo = fminf(lo, s_lo[w]); hi = fmaxf(hi, s_hi[w]); pm = fmaxf(pm, s_pm[w]); }
    double mx = 1.7320508 * (double)(hi - lo);
    if (!(mx > 0.0)) mx = 1.0;
    const double tol = mx * 1e-13;
    // Nearly no pair is tight, and the exact test costs a correctly rounded square root (a dozen instructions), a conversion
    // and two double-precision operations per pair.  In front of it, in fp32 and without the root: c_ij <= t := (u_i + margin) - p_j,
    // i.e. t >= 0 and the SQUARED distance <= t^2 (1 + 2^-20) -- the margin is eight times the worst rounding of the two rounded
    // potentials and of the subtraction (2^-24 each of |p_j|, |u_i| and the difference), the factor covers the root's and the
    // square's roundings (2^-23 in all): a superset of the tight pairs passes, the exact test decides.
    for (int t = 0; t < nblk; ++t) {
        if (TIE_ROWS * t - q0 >= 64) {                                  // the next 64 rows of the wave's sequence
            q0 = TIE_ROWS * t;
            il = row_of(q0 + lane); ic = il < n ? il : n - 1;
            lx = S_[3 * ic]; ly = S_[3 * ic + 1]; lz = S_[3 * ic + 2];
            lc = a.col4row[(size_t)b * n + ic];
        }
        const int i0 = (first + t * stride) * TIE_ROWS;
        float ax[TIE_ROWS], ay[TIE_ROWS], az[TIE_ROWS], curf[TIE_ROWS];      // curf: u_i + margin, rounded, relative to pref
        double cur[TIE_ROWS];
        int c[TIE_ROWS];
        unsigned long long hits[TIE_ROWS];                              // bit s: the lane's column 64 s + lane is a tight pair of row r
#pragma unroll
        for (int r = 0; r < TIE_ROWS; ++r) {
            const int k = TIE_ROWS * t - q0 + r;                        // (rows beyond n - 1 repeat row n - 1: masked below)
            ax[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lx), k));
            ay[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ly), k));
            az[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lz), k));
            c[r] = __builtin_amdgcn_readlane(lc, k);
            if (c[r] < 0 || c[r] >= n) {                                // not an assignment: nothing to say about it
                if (lane == 0) atomicMax(&a.tie[b], 3);
                c[r] = 0;
            }
            cur[r] = (double)sqrtf(reart_sqdist3(ax[r], ay[r], az[r], ptx[c[r]], pty[c[r]], ptz[c[r]])) + price[c[r]];
            curf[r] = (float)(cur[r] - pref);
            curf[r] += 4.76837158203125e-7f * (2.f * pm + 2.f * (float)mx + fabsf(curf[r]));      // + margin: 2^-21 x ... (covers this rounding too)
            hits[r] = 0ull;
        }
        for (int j0 = 0, st = 0; j0 < n; j0 += 64, ++st) {
            const int j = j0 + lane;
            const bool in = j < n;
            const int jj = in ? j : n - 1;
            const float tx = ptx[jj], ty = pty[jj], tz = ptz[jj], pj = pf[jj];
#pragma unroll
            for (int r = 0; r < TIE_ROWS; ++r) {
                const float tt = curf[r] - pj;
                if (in && tt >= 0.f && reart_sqdist3(ax[r], ay[r], az[r], tx, ty, tz) <= (tt * tt) * 1.00000095367431640625f) {
                    if (j != c[r] && ((double)sqrtf(reart_sqdist3(ax[r], ay[r], az[r], tx, ty, tz)) + price[j]) - cur[r] <= tol)
                        hits[r] |= 1ull << st;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < TIE_ROWS; ++r) {                            // the row's pairs, lowest lane first, to the row's own slots
            const int i = i0 + r;
            if (i >= n) break;
            int *out = a.cols + ((size_t)b * n + i) * a.K;
            int cnt = 0;
            for (unsigned long long m = __ballot(hits[r] != 0ull); m; m = __ballot(hits[r] != 0ull)) {
                const int l = __ffsll((long long)m) - 1;
                int col = 0;
                if (lane == l) { const int s_ = __ffsll((long long)hits[r]) - 1; hits[r] &= hits[r] - 1ull; col = 64 * s_ + lane; }
                col = __builtin_amdgcn_readlane(col, l);
                if (lane == 0 && cnt < a.K) out[cnt] = col;
                ++cnt;
            }
            if (lane == 0) a.cnt[(size_t)b * n + i] = cnt;
        }
    }
}

// "did any thread see it?" with ONE barrier: three flags in rotation -- call k sets flag k % 3 and, behind its barrier, clears
// flag (k + 2) % 3 (last read before this barrier, next set behind the following one).  __syncthreads_or is three barriers'
// worth, and this kernel is a chain of such questions.
struct TieOr {
    int *flag;
    int k;
    __device__ __forceinline__ bool operator()(int v) {
        const int cur = k % 3;
        if (v) flag[cur] = 1;
        __syncthreads();
        const bool r = flag[cur] != 0;
        if (threadIdx.x == 0) flag[(k + 2) % 3] = 0;
        ++k;
        return r;
    }
};

__global__ __launch_bounds__(TIE_CYC_BS) void lap_tie_cycle_kernel(TieArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    const int n = a.n, b = blockIdx.x, tid = threadIdx.x, K = a.K;
    __shared__ int s_or[3];
    if (!a.fresh && a.tie[b] == 3) return;
    if (a.stale && a.stale[b]) { if (tid == 0) a.tie[b] = 2; return; }
    // Does the row graph hold a cycle?  Peeling rows without a live pair out of them one layer per round takes as many rounds
    // as the longest chain of tight pairs -- hundreds: the tight pairs are mostly what the searches' and the backward growth's
    // trees left behind, chains by construction (measured: 0.4 ms per solve of 9 x 2048^2).  So the chains are CONTRACTED first:
    // a row with exactly one tight pair out of it is a link (pointer jumping takes every link to the end of its chain in
    // log2(longest chain) rounds: a row with no pair out of it = dead, a row with several = a junction; a link that never
    // arrives runs into a cycle of links), and only the junctions are peeled layer by layer -- a junction dies when every pair
    // out of it leads to a dead end.  Junctions that survive the fixed point all lead to surviving junctions: a cycle.
    unsigned short *owner = (unsigned short *)lsm;                     // column -> row
    unsigned short *ptr = owner + n;                                   // a link: where its chain has got to; an end: itself
    unsigned short *nbr = ptr + n;                                     // [n][K] the rows a row's pairs lead to
    unsigned char *deg = (unsigned char *)(nbr + (size_t)n * K);       // pairs out of the row (<= K)
    unsigned char *dead = deg + n;
    TieOr wg_or{s_or, 0};
    if (tid < 3) s_or[tid] = 0;
    const int *cnt = a.cnt + (size_t)b * n, *cols = a.cols + (size_t)b * n * K;
    int over = 0, any = 0;
    for (int i = tid; i < n; i += TIE_CYC_BS) {
        const int c = a.col4row[(size_t)b * n + i];
        int d = cnt[i];
        d = d < 0 ? 0 : d;                                             // (rows of a problem the solve did not certify were never written)
        if (c >= 0 && c < n) owner[c] = (unsigned short)i;             // (a permutation whenever the solve was certified)
        over |= d > K; any |= d > 0;
        deg[i] = (unsigned char)(d < K ? d : K);
        dead[i] = d == 0;
    }
    if (wg_or(over)) { if (tid == 0) a.tie[b] = 2; return; }            // a row with more tight pairs than it has slots: the host looks itself
    if (!wg_or(any)) { if (a.fresh && tid == 0) a.tie[b] = 0; return; }   // (otherwise tie[b] was cleared by the launch's memset)
    for (int i = tid; i < n; i += TIE_CYC_BS) {
        const int d = deg[i];
        for (int k = 0; k < d; ++k) nbr[(size_t)i * K + k] = owner[(unsigned)cols[(size_t)i * K + k] < (unsigned)n ? cols[(size_t)i * K + k] : 0];
        ptr[i] = d == 1 ? nbr[(size_t)i * K] : (unsigned short)i;      // ends of chains (dead ends, junctions) point at themselves
    }
    __syncthreads();
    // ptr[i] <- ptr[ptr[i]] in place (a stale read is still a row further down the same chain) until nothing moves:
    // ceil(log2(longest chain)) rounds of one barrier; a cycle of links comes to rest on links
    int steps = 1;
    while ((1 << steps) < n) ++steps;
    for (int r = 0; r <= steps; ++r) {
        int moved = 0;
        for (int i = tid; i < n; i += TIE_CYC_BS) {
            const int p1 = ptr[i], p2 = ptr[p1];
            if (p2 != p1) { ptr[i] = (unsigned short)p2; moved = 1; }
        }
        if (!wg_or(moved)) break;
    }
    // a link whose pointer has not arrived at an end of a chain (it rests on a link: itself or another) runs into a cycle of links
    int cyc = 0, junctions = 0;
    for (int i = tid; i < n; i += TIE_CYC_BS) { cyc |= (deg[i] == 1 && deg[ptr[i]] == 1); junctions |= deg[i] > 1; }
    if (wg_or(cyc)) { if (tid == 0) a.tie[b] = 1; return; }
    if (!wg_or(junctions)) { if (a.fresh && tid == 0) a.tie[b] = 0; return; }   // chains only, all of them ending: no cycle
    for (int round = 0; round <= n; ++round) {                         // a junction dies when every pair out of it leads to a dead end
        int died = 0;
        for (int i = tid; i < n; i += TIE_CYC_BS) {
            const int d = deg[i];
            if (d > 1 && !dead[i]) {
                bool live = false;
                for (int k = 0; k < d; ++k) live = live || !dead[ptr[nbr[(size_t)i * K + k]]];
                if (!live) { dead[i] = 1; died = 1; }
            }
        }
        if (!wg_or(died)) break;
    }
    int left = 0;
    for (int i = tid; i < n; i += TIE_CYC_BS) left |= (deg[i] > 1 && !dead[i]);
    const int cyc2 = wg_or(left);
    if (tid == 0 && (cyc2 || a.fresh)) a.tie[b] = cyc2 ? 1 : 0;
}

static int tie_cycle_launch(const TieArgs &a, hipStream_t stream) {
    const size_t lds_cyc = (size_t)a.n * (2 + 2 + 2 * (size_t)a.K + 1 + 1) + 16;
    if (lds_cyc > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute((const void *)lap_tie_cycle_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cyc) != hipSuccess)
        return REART_ERR_LAUNCH;
    hipLaunchKernelGGL(lap_tie_cycle_kernel, dim3(a.B), dim3(TIE_CYC_BS), lds_cyc, stream, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// the cycle check alone, over pairs a solve's certificate pass listed (lap.hip: reart_lap_resolve_points_mc_ties)
int reart_internal_tie_cycles(int B, int n, const int *col4row, int *tie, const int *cols, const int *cnt, int K, const int *stale,
                              hipStream_t stream) {
    if (n > LAP_NMAX || K < 1 || K > 32) return REART_ERR_INVALID_ARG;
    TieArgs a{nullptr, nullptr, B, n, col4row, nullptr, tie, (int *)cols, (int *)cnt, K, stale, 1};
    return tie_cycle_launch(a, stream);
}

extern "C" int reart_lap_ties(const float *src, const float *tgt, int B, int n, const int32_t *col4row, const double *price,
                              int32_t *tie, int32_t *cols, int32_t *cnt, int K, void *stream) {
    if (B < 0 || n < 1 || n > LAP_NMAX || K < 1 || K > 32) return REART_ERR_INVALID_ARG;
    if (B == 0) return REART_OK;
    if (!src || !tgt || !col4row || !price || !tie || !cols || !cnt) return REART_ERR_INVALID_ARG;
    TieArgs a{src, tgt, B, n, col4row, price, tie, cols, cnt, K, nullptr, 0};
    if (hipMemsetAsync(tie, 0, sizeof(int) * (size_t)B, (hipStream_t)stream) != hipSuccess) return REART_ERR_LAUNCH;
    int per = (256 + B - 1) / B;                                       // workgroups (16 waves) per problem: one per compute unit over the batch
    const int per_max = ((n + TIE_ROWS - 1) / TIE_ROWS + TIE_PASS_BS / 64 - 1) / (TIE_PASS_BS / 64);
    per = per < 1 ? 1 : (per > per_max ? per_max : per);
    const size_t lds_pass = (size_t)n * (8 + 16);
    if (lds_pass > REART_LDS_DEFAULT_CAP &&
        hipFuncSetAttribute((const void *)lap_tie_edges_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pass) != hipSuccess)
        return REART_ERR_LAUNCH;
    hipLaunchKernelGGL(lap_tie_edges_kernel, dim3(B, per), dim3(TIE_PASS_BS), lds_pass, (hipStream_t)stream, a);
    REART_CHECK_LAUNCH();
    return tie_cycle_launch(a, (hipStream_t)stream);
}
