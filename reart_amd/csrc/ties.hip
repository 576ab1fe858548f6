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
//   lap_tie_edges_kernel  whole chip, one wave per row: the tight pairs off the assignment, with the costs' own expression
//                         and the certificate's tolerance (lap.hip: `cur - v1 > tol`, tol = 1e-13 of the cost scale)
//   lap_tie_cycle_kernel  one workgroup per problem: does that graph hold a cycle?  Chains of rows with one tight pair out
//                         of them are contracted by pointer jumping, the rows with several are peeled layer by layer
//                         (see there).  No cycle (the normal case): tie[b] = 0.
#include "common.h"
#include "lap_dev.h"

#define TIE_PASS_BS 1024
#define TIE_CYC_BS 1024
#define TIE_LDS_EDGES 12288        // tight pairs one workgroup keeps in LDS (48 KB); more: tie[b] = 2, the host looks itself

struct TieArgs {
    const float *src, *tgt;        // [B][n][3]
    int B, n;
    const int *col4row;            // [B][n] the optimum
    const double *price;           // [B][n] its column potentials (the solvers' sign convention: a row minimises c + p)
    int *tie;                      // [B] out
    int *edges;                    // [B][cap][2] out: (row, column) of every tight pair off the assignment
    int *n_edges;                  // [B] out (may exceed cap: the pairs beyond it are not stored)
    int cap;
};

#define TIE_ROWS 4                 // rows a wave tests per pass over the columns (one set of LDS reads serves all of them)
// One workgroup of 16 waves per compute unit; the staging loads are all in flight before the first LDS store, and a wave fetches
// the points and columns of all its rows at once (lane k holds row k): as four waves per workgroup with a load per row the
// kernel spent 100 us waiting for round trips whatever its scan cost (the same finding as lap_jv_pass_pts_kernel, lap.hip).
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
    int *edges = a.edges + (size_t)b * a.cap * 2;
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
        }
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const bool in = j < n;
            const int jj = in ? j : n - 1;
            const float tx = ptx[jj], ty = pty[jj], tz = ptz[jj], pj = pf[jj];
            unsigned near = 0;
#pragma unroll
            for (int r = 0; r < TIE_ROWS; ++r) {
                const float tt = curf[r] - pj;
                near |= ((tt >= 0.f && reart_sqdist3(ax[r], ay[r], az[r], tx, ty, tz) <= (tt * tt) * 1.00000095367431640625f) ? 1u : 0u) << r;
            }
            if (!in) near = 0;
            if (__ballot(near != 0)) {
#pragma unroll
                for (int r = 0; r < TIE_ROWS; ++r) {
                    bool hit = false;
                    if (((near >> r) & 1u) && i0 + r < n && j != c[r])
                        hit = ((double)sqrtf(reart_sqdist3(ax[r], ay[r], az[r], tx, ty, tz)) + price[j]) - cur[r] <= tol;
                    const unsigned long long m = __ballot(hit);
                    if (m) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&a.n_edges[b], __builtin_popcountll(m));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (hit) {
                            const int at = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                            if (at < a.cap) { edges[2 * at] = i0 + r; edges[2 * at + 1] = j; }
                        }
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(TIE_CYC_BS) void lap_tie_cycle_kernel(TieArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
    const int n = a.n, b = blockIdx.x, tid = threadIdx.x;
    if (a.tie[b] == 3) return;
    const int E = a.n_edges[b];
    if (E == 0) return;                                                // (tie[b] was cleared by the launch's memset)
    if (E > a.cap || E > TIE_LDS_EDGES) { if (tid == 0) a.tie[b] = 2; return; }
    // Does the row graph hold a cycle?  Peeling rows without a live pair out of them one layer per round takes as many rounds
    // as the longest chain of tight pairs -- hundreds: the tight pairs are mostly what the searches' and the backward growth's
    // trees left behind, chains by construction (measured: 0.4 ms per solve of 9 x 2048^2).  So the chains are CONTRACTED first:
    // a row with exactly one tight pair out of it is a link (12 rounds of pointer jumping take every link to the end of its
    // chain: a row with no pair out of it = dead, a row with several = a junction; a link that never arrives runs into a cycle
    // of links), and only the junctions are peeled layer by layer -- a junction dies when every pair out of it leads to a dead
    // end.  Junctions that survive the fixed point all lead to surviving junctions: a cycle.
    int *owner = (int *)lsm;                                           // column -> row
    int *deg = owner + n;                                              // tight pairs out of the row
    int *ptr = deg + n;                                                // a link: where its chain has got to; an end: itself
    int *mark = ptr + n;                                               // the last round in which a junction saw a live way out
    unsigned *pair = (unsigned *)(mark + n);                           // (row << 16) | row the pair leads to
    unsigned char *dead = (unsigned char *)(pair + TIE_LDS_EDGES);
    for (int i = tid; i < n; i += TIE_CYC_BS) { owner[i] = 0xffff; deg[i] = 0; ptr[i] = i; mark[i] = -1; dead[i] = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += TIE_CYC_BS) owner[a.col4row[(size_t)b * n + i]] = i;      // (a permutation: tie[b] != 3)
    __syncthreads();
    const int *edges = a.edges + (size_t)b * a.cap * 2;
    for (int e = tid; e < E; e += TIE_CYC_BS) {
        const int i = edges[2 * e], k = owner[edges[2 * e + 1]];
        pair[e] = ((unsigned)i << 16) | (unsigned)k;
        if (k != 0xffff) { atomicAdd(&deg[i], 1); ptr[i] = k; }        // (deg 1: the one way out; otherwise overwritten below)
    }
    __syncthreads();
    for (int i = tid; i < n; i += TIE_CYC_BS) {
        if (deg[i] != 1) ptr[i] = i;                                   // ends of chains: dead ends and junctions
        dead[i] = deg[i] == 0;
    }
    __syncthreads();
    int steps = 1;
    while ((1 << steps) < n) ++steps;
    for (int r = 0; r < steps; ++r) {                                  // ptr[i] <- ptr[ptr[i]]: ends point at themselves
        int nx[(LAP_NMAX + TIE_CYC_BS - 1) / TIE_CYC_BS];
        int c = 0;
        for (int i = tid; i < n; i += TIE_CYC_BS) nx[c++] = ptr[ptr[i]];
        __syncthreads();
        c = 0;
        for (int i = tid; i < n; i += TIE_CYC_BS) ptr[i] = nx[c++];
        __syncthreads();
    }
    int cyc = 0;                                                       // a link whose chain has no end: a cycle of links
    for (int i = tid; i < n; i += TIE_CYC_BS) cyc |= (deg[i] == 1 && deg[ptr[i]] == 1);
    if (__syncthreads_or(cyc)) { if (tid == 0) a.tie[b] = 1; return; }
    for (int round = 0; round <= n; ++round) {
        for (int e = tid; e < E; e += TIE_CYC_BS) {
            const int i = (int)(pair[e] >> 16), k = (int)(pair[e] & 0xffffu);
            if (k != 0xffff && deg[i] > 1 && !dead[i] && !dead[ptr[k]]) mark[i] = round;   // (every writer writes the same value)
        }
        __syncthreads();
        int died = 0;
        for (int i = tid; i < n; i += TIE_CYC_BS)
            if (deg[i] > 1 && !dead[i] && mark[i] != round) { dead[i] = 1; died = 1; }
        if (!__syncthreads_or(died)) break;
    }
    int left = 0;
    for (int i = tid; i < n; i += TIE_CYC_BS) left |= (deg[i] > 1 && !dead[i]);
    if (__syncthreads_or(left) && tid == 0) a.tie[b] = 1;
}

extern "C" int reart_lap_ties(const float *src, const float *tgt, int B, int n, const int32_t *col4row, const double *price,
                              int32_t *tie, int32_t *edges, int32_t *n_edges, int cap, void *stream) {
    if (B < 0 || n < 1 || n > LAP_NMAX || cap < 1) return REART_ERR_INVALID_ARG;
    if (B == 0) return REART_OK;
    if (!src || !tgt || !col4row || !price || !tie || !edges || !n_edges) return REART_ERR_INVALID_ARG;
    TieArgs a{src, tgt, B, n, col4row, price, tie, edges, n_edges, cap};
    if (n_edges == tie + B) {                                          // one buffer of 2 B ints (reart_amd/utils/lap.py): one fill
        if (hipMemsetAsync(tie, 0, sizeof(int) * 2 * (size_t)B, (hipStream_t)stream) != hipSuccess) return REART_ERR_LAUNCH;
    } else if (hipMemsetAsync(tie, 0, sizeof(int) * (size_t)B, (hipStream_t)stream) != hipSuccess ||
               hipMemsetAsync(n_edges, 0, sizeof(int) * (size_t)B, (hipStream_t)stream) != hipSuccess)
        return REART_ERR_LAUNCH;
    int per = (256 + B - 1) / B;                                       // workgroups (16 waves) per problem: one per compute unit over the batch
    const int per_max = ((n + TIE_ROWS - 1) / TIE_ROWS + TIE_PASS_BS / 64 - 1) / (TIE_PASS_BS / 64);
    per = per < 1 ? 1 : (per > per_max ? per_max : per);
    const size_t lds_pass = (size_t)n * (8 + 16);
    const size_t lds_cyc = (size_t)n * (4 * 4 + 1) + 4 * (size_t)TIE_LDS_EDGES + 16;
    if ((lds_pass > REART_LDS_DEFAULT_CAP &&
         hipFuncSetAttribute((const void *)lap_tie_edges_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pass) != hipSuccess) ||
        (lds_cyc > REART_LDS_DEFAULT_CAP &&
         hipFuncSetAttribute((const void *)lap_tie_cycle_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cyc) != hipSuccess))
        return REART_ERR_LAUNCH;
    hipLaunchKernelGGL(lap_tie_edges_kernel, dim3(B, per), dim3(TIE_PASS_BS), lds_pass, (hipStream_t)stream, a);
    REART_CHECK_LAUNCH();
    hipLaunchKernelGGL(lap_tie_cycle_kernel, dim3(B), dim3(TIE_CYC_BS), lds_cyc, (hipStream_t)stream, a);
    REART_CHECK_LAUNCH();
    return REART_OK;
}
