// reart_amd/csrc/grid.hip -- EXACT K-nearest-neighbour search against STATIC target sets through a
// uniform grid (K = 1 and K = 3).  Same answer, bit for bit, as the brute-force kernels of knn.hip
// (same distance expression, ties -> lowest original index), but ~30-50x fewer distance
// evaluations: in the relaxation loop the observed frames `pc_list` (Chamfer direction
// pc_trans -> pc_list, reference utils/chamfer.py:78-84) and the flow reference sets
// (utils/flow_utils.py:158) never change, so their grids are built once per instance.
//
// Exactness argument.  Every candidate that is evaluated uses reart_sqdist3 and the winner is the
// minimum (d, original index) key, exactly the brute-force rule.  A point is skipped only when it
// lies in a cell outside the visited cube of cells around the query, and the search stops only
// when  best_K < (gap - eps)^2 (1 - eps)  where gap is the distance from the query to the nearest
// face of the visited cube that still has unvisited cells behind it.  Points were assigned to cells
// with fp32 arithmetic, so a point of an unvisited cell can sit at most ~3e-7 * extent inside the
// nominal face; eps = 4e-6 (relative to the grid extent) covers that and the rounding of the
// distance itself with an order of magnitude to spare.  Conservative => never wrong, only slower.
#include "common.h"
#include "internal.h"
#include "blocksort.h"
#include <math.h>

#define GR_G 16                       // cells per axis
#define GR_CELLS (GR_G * GR_G * GR_G)
#define GR_META 8                     // floats per set: ox, oy, oz, h, invh, extent, n, pad

// ---------------------------------------------------------------------------------------
// build: one 1024-thread workgroup per target set
// ---------------------------------------------------------------------------------------

__global__ __launch_bounds__(RS_BS) void grid_build_kernel(GridBuildArgs a) {
    __shared__ int s_cnt[RS_DIG * RS_BS];
    __shared__ int s_wave[RS_BS / 64];
    __shared__ float s_mn[3][RS_BS / 64], s_mx[3][RS_BS / 64];
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int o = a.offsets ? a.offsets[e] : e * a.N;
    const int n = a.offsets ? a.offsets[e + 1] - o : a.N;
    const float *p = a.pts + 3 * (size_t)o;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += RS_BS)
#pragma unroll
        for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], p[3 * i + c]); mx[c] = fmaxf(mx[c], p[3 * i + c]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], s, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], s, 64));
        }
        if (lane == 0) { s_mn[c][wv] = mn[c]; s_mx[c][wv] = mx[c]; }
    }
    __syncthreads();
    float org[3], ext = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float lo = INFINITY, hi = -INFINITY;
        for (int w = 0; w < RS_BS / 64; ++w) { lo = fminf(lo, s_mn[c][w]); hi = fmaxf(hi, s_mx[c][w]); }
        org[c] = lo;
        ext = fmaxf(ext, hi - lo);
    }
    if (!(ext > 0.f)) ext = 1.0f;                       // single point / empty set
    const float h = ext * (1.0f + 1e-5f) / (float)GR_G;  // the maximum coordinate maps below GR_G
    const float invh = 1.0f / h;
    if (tid == 0) {
        float *m = a.meta + (size_t)e * GR_META;
        m[0] = org[0]; m[1] = org[1]; m[2] = org[2]; m[3] = h; m[4] = invh; m[5] = ext; m[6] = (float)n; m[7] = 0.f;
    }
    int *cid = a.scratch + (size_t)e * 3 * a.stride, *bufA = cid + a.stride, *bufB = bufA + a.stride;
    int *cs = a.cell_start + (size_t)e * (GR_CELLS + 1);
    for (int c = tid; c <= GR_CELLS; c += RS_BS) cs[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += RS_BS) {
        int cc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v = (int)floorf((p[3 * i + c] - org[c]) * invh);
            cc[c] = v < 0 ? 0 : (v > GR_G - 1 ? GR_G - 1 : v);
        }
        const int id = (cc[2] * GR_G + cc[1]) * GR_G + cc[0];
        cid[i] = id;
        atomicAdd(&cs[id + 1], 1);  // integer atomics: order-independent histogram
    }
    __syncthreads();
    // inclusive scan of the histogram in place: cs[c] = number of points in cells < c
    const int chunk = (GR_CELLS + RS_BS - 1) / RS_BS;
    const int c0 = 1 + tid * chunk, c1 = (c0 + chunk <= GR_CELLS + 1) ? c0 + chunk : GR_CELLS + 1;
    int tot = 0;
    for (int c = c0; c < c1; ++c) tot += cs[c];
    int run = block_excl_scan(tot, s_wave, nullptr);
    for (int c = c0; c < c1; ++c) { run += cs[c]; cs[c] = run; }
    const int *sorted = block_stable_sort_ids(n, 12, bufA, bufB, s_cnt, s_wave, [&](int i) { return cid[i]; });
    float *gx = a.gx + (size_t)e * a.stride, *gy = a.gy + (size_t)e * a.stride, *gz = a.gz + (size_t)e * a.stride;
    int *go = a.gorig + (size_t)e * a.stride;
    for (int q = tid; q < a.stride; q += RS_BS) {
        if (q < n) {
            const int i = sorted[q];
            gx[q] = p[3 * i]; gy[q] = p[3 * i + 1]; gz[q] = p[3 * i + 2]; go[q] = i;
        } else {
            gx[q] = INFINITY; gy[q] = INFINITY; gz[q] = INFINITY; go[q] = 0x7fffffff;
        }
    }
}

// ---------------------------------------------------------------------------------------
// query: one wave per query point
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ bool key_lt(float d, int i, float d2, int i2) { return d < d2 || (d == d2 && i < i2); }

template <int KK>
__device__ __forceinline__ void key_insert(float (&kd)[KK], int (&ki)[KK], float d, int j) {
    if (!key_lt(d, j, kd[KK - 1], ki[KK - 1])) return;
#pragma unroll
    for (int s = KK - 1; s >= 0; --s) {
        const int sp = s > 0 ? s - 1 : 0;
        const bool lp = (s > 0) && key_lt(d, j, kd[sp], ki[sp]);
        const bool lc = key_lt(d, j, kd[s], ki[s]);
        kd[s] = lp ? kd[sp] : (lc ? d : kd[s]);
        ki[s] = lp ? ki[sp] : (lc ? j : ki[s]);
    }
}

// smallest key among the GL lanes of a query group; every lane of the group gets it
#define GQ_GL 16          // lanes cooperating on one query (4 queries per wave keep 4x more
                          // independent load chains in flight: the search is latency bound)
#define GQ_WAVES 4        // waves per workgroup
#define GQ_QPB (GQ_WAVES * 64 / GQ_GL)

__device__ __forceinline__ void group_min_key(float d, int i, float *od, int *oi) {
#pragma unroll
    for (int s = GQ_GL / 2; s >= 1; s >>= 1) {
        const float d2 = __shfl_xor(d, s, 64);
        const int i2 = __shfl_xor(i, s, 64);
        if (key_lt(d2, i2, d, i)) { d = d2; i = i2; }
    }
    *od = d; *oi = i;
}

template <int KK>
__global__ __launch_bounds__(64 * GQ_WAVES) void grid_knn_kernel(GridQueryArgs a) {
    __shared__ int s_start[GQ_QPB][GQ_GL];
    __shared__ int s_off[GQ_QPB][GQ_GL + 1];
    const int gl = threadIdx.x & (GQ_GL - 1), grp = threadIdx.x / GQ_GL;   // lane in group, group in block
    const int qi = blockIdx.x * GQ_QPB + grp, e = blockIdx.y;
    const bool valid = qi < a.nq;
    const int qic = valid ? qi : a.nq - 1;
    const int qb = a.qmap ? a.qmap[e] : e;
    const float *qp = (qb < 0 ? a.q_alt : a.q + (size_t)qb * a.nq * 3) + 3 * (size_t)qic;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float *m = a.meta + (size_t)e * GR_META;
    const float ox = m[0], oy = m[1], oz = m[2], h = m[3], invh = m[4], ext = m[5];
    const int *cs = a.cell_start + (size_t)e * (GR_CELLS + 1);
    const float *gx = a.gx + (size_t)e * a.stride, *gy = a.gy + (size_t)e * a.stride, *gz = a.gz + (size_t)e * a.stride;
    const int *go = a.gorig + (size_t)e * a.stride;
    int cq[3];
    {
        const float qq[3] = {qx, qy, qz}, oo[3] = {ox, oy, oz};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float f = floorf((qq[c] - oo[c]) * invh);
            cq[c] = f < 0.f ? 0 : (f > (float)(GR_G - 1) ? GR_G - 1 : (int)f);
        }
    }
    float kd[KK];
    int ki[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) { kd[k] = INFINITY; ki[k] = 0x7fffffff; }
    float rd[KK];   // group-merged result (valid after a merge)
    int ri[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) { rd[k] = INFINITY; ri[k] = 0x7fffffff; }

    // All control flow below is per query group (the 16 lanes of a group always agree), so the
    // shuffles of width 16 only ever exchange data between lanes that are active together.
    bool done = false;
    for (int r = 1; r <= GR_G && !done; ++r) {
        // --- segments of this pass: r == 1 -> the full 3x3x3 cube as 9 x-runs; r >= 2 -> the shell
        const int side = 2 * r + 1;
        const int inner = side - 2;
        const int nfull = (r == 1) ? 9 : side * side - inner * inner;      // rows with a full x-run
        const int nseg = (r == 1) ? 9 : nfull + 2 * inner * inner;         // + two end cells per inner row
        for (int s0 = 0; s0 < nseg; s0 += GQ_GL) {
            const int sg = s0 + gl;
            int start = 0, len = 0;
            if (sg < nseg) {
                int dy, dz, x0, x1;
                if (sg < nfull) {
                    const int row = sg;
                    if (r == 1) { dy = row % 3 - 1; dz = row / 3 - 1; }
                    else if (row < 2 * side) { dz = (row < side) ? -r : r; dy = (row % side) - r; }
                    else { const int t = row - 2 * side; dy = (t & 1) ? r : -r; dz = (t >> 1) - (r - 1); }
                    x0 = cq[0] - r; x1 = cq[0] + r;
                } else {
                    const int t = sg - nfull, row = t >> 1;
                    dy = row % inner - (r - 1); dz = row / inner - (r - 1);
                    x0 = x1 = (t & 1) ? cq[0] + r : cq[0] - r;
                }
                const int cy = cq[1] + dy, cz = cq[2] + dz;
                if (cy >= 0 && cy < GR_G && cz >= 0 && cz < GR_G) {
                    x0 = x0 < 0 ? 0 : x0;
                    x1 = x1 > GR_G - 1 ? GR_G - 1 : x1;
                    if (x0 <= x1) {   // an end cell outside the grid gives x0 > x1 after clipping
                        const int base = (cz * GR_G + cy) * GR_G;
                        start = cs[base + x0];
                        len = cs[base + x1 + 1] - start;
                    }
                }
            }
            // inclusive scan of len over the group -> candidate offsets
            int inc = len;
#pragma unroll
            for (int o = 1; o < GQ_GL; o <<= 1) {
                const int u = __shfl_up(inc, o, GQ_GL);
                if (gl >= o) inc += u;
            }
            const int total = __shfl(inc, GQ_GL - 1, GQ_GL);
            s_start[grp][gl] = start;
            s_off[grp][gl] = inc - len;
            if (gl == GQ_GL - 1) s_off[grp][GQ_GL] = total;
            __builtin_amdgcn_wave_barrier();   // same wave: LDS accesses complete in program order
            for (int t0 = 0; t0 < total; t0 += GQ_GL) {
                const int cnd = t0 + gl;
                if (cnd < total) {
                    int sidx = 0;   // largest s with off[s] <= cnd: independent LDS reads, no chain
#pragma unroll
                    for (int u = 1; u < GQ_GL; ++u) sidx += (s_off[grp][u] <= cnd) ? 1 : 0;
                    const int pidx = s_start[grp][sidx] + (cnd - s_off[grp][sidx]);
                    const float d = reart_sqdist3(qx, qy, qz, gx[pidx], gy[pidx], gz[pidx]);
                    key_insert<KK>(kd, ki, d, go[pidx]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // --- group merge of the lane-local lists -> K best keys so far
        {
            float td[KK];
            int ti[KK];
#pragma unroll
            for (int k = 0; k < KK; ++k) { td[k] = kd[k]; ti[k] = ki[k]; }
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                float bd; int bi;
                group_min_key(td[0], ti[0], &bd, &bi);
                rd[k] = bd; ri[k] = bi;
                if (td[0] == bd && ti[0] == bi) {   // the owner pops its head (keys are unique)
#pragma unroll
                    for (int u = 0; u + 1 < KK; ++u) { td[u] = td[u + 1]; ti[u] = ti[u + 1]; }
                    td[KK - 1] = INFINITY; ti[KK - 1] = 0x7fffffff;
                }
            }
        }
        // --- termination: distance from the query to the nearest face with unvisited cells behind it
        float gap = INFINITY;
        bool any_face = false;
        {
            const float qq[3] = {qx, qy, qz}, oo[3] = {ox, oy, oz};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (cq[c] - r > 0) { any_face = true; gap = fminf(gap, qq[c] - (oo[c] + (float)(cq[c] - r) * h)); }
                if (cq[c] + r < GR_G - 1) { any_face = true; gap = fminf(gap, (oo[c] + (float)(cq[c] + r + 1) * h) - qq[c]); }
            }
        }
        const float g = gap - 4e-6f * (ext + fabsf(qx - ox) + fabsf(qy - oy) + fabsf(qz - oz));
        done = !any_face || (g > 0.f && rd[KK - 1] < (g * g) * (1.0f - 4e-6f));
    }
    if (gl == 0 && valid) {
        const size_t o = ((size_t)e * a.nq + qi) * KK;
#pragma unroll
        for (int k = 0; k < KK; ++k) { a.od[o + k] = rd[k]; a.oi[o + k] = ri[k] == 0x7fffffff ? 0 : ri[k]; }
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
size_t reart_grid_bytes(int E, int stride) {
    size_t off = 0;
    off += 3 * reart_align_up(sizeof(float) * (size_t)E * stride, 256);      // gx gy gz
    off += reart_align_up(sizeof(int) * (size_t)E * stride, 256);            // gorig
    off += reart_align_up(sizeof(int) * (size_t)E * (GR_CELLS + 1), 256);    // cell_start
    off += reart_align_up(sizeof(float) * (size_t)E * GR_META, 256);         // meta
    off += reart_align_up(sizeof(int) * (size_t)E * 3 * stride, 256);        // build scratch
    return off;
}

void reart_grid_layout(void *mem, int E, int stride, GridBuildArgs *b) {
    char *p = (char *)mem;
    const size_t f = reart_align_up(sizeof(float) * (size_t)E * stride, 256);
    b->gx = (float *)p; p += f;
    b->gy = (float *)p; p += f;
    b->gz = (float *)p; p += f;
    b->gorig = (int *)p; p += reart_align_up(sizeof(int) * (size_t)E * stride, 256);
    b->cell_start = (int *)p; p += reart_align_up(sizeof(int) * (size_t)E * (GR_CELLS + 1), 256);
    b->meta = (float *)p; p += reart_align_up(sizeof(float) * (size_t)E * GR_META, 256);
    b->scratch = (int *)p;
    b->stride = stride;
}

int reart_grid_build_launch(const GridBuildArgs &b, int E, hipStream_t st) {
    hipLaunchKernelGGL(grid_build_kernel, dim3(E), dim3(RS_BS), 0, st, b);
    REART_CHECK_LAUNCH();
    return REART_OK;
}

int reart_grid_query_launch(const GridQueryArgs &q, int K, hipStream_t st) {
    const dim3 grid(reart_div_up(q.nq, GQ_QPB), q.E);
    if (K == 1) hipLaunchKernelGGL(grid_knn_kernel<1>, grid, dim3(64 * GQ_WAVES), 0, st, q);
    else if (K == 3) hipLaunchKernelGGL(grid_knn_kernel<3>, grid, dim3(64 * GQ_WAVES), 0, st, q);
    else return REART_ERR_UNSUPPORTED;
    REART_CHECK_LAUNCH();
    return REART_OK;
}

// Stand-alone entry (tests, and callers whose targets are static): build + query in one call.
//   targets [E,Nt,3] (offsets == NULL) or ragged (offsets [E+1] into a concatenated array),
//   queries [E,nq,3]; dists [E,nq,K] squared ascending, idx [E,nq,K] i32.  K in {1, 3}; every set
//   must hold at least K points.
extern "C" size_t reart_grid_knn_workspace_bytes(int E, int Nt_max) {
    if (E <= 0 || Nt_max <= 0) return 0;
    return reart_grid_bytes(E, (int)reart_align_up((size_t)Nt_max, 64));
}

extern "C" int reart_grid_knn(const float *targets, const int32_t *offsets, int E, int Nt_max,
                              const float *queries, int nq, int K, float *dists, int32_t *idx,
                              void *workspace, size_t workspace_bytes, void *stream) {
    if (E < 0 || Nt_max < 1 || nq < 0) return REART_ERR_INVALID_ARG;
    if (K != 1 && K != 3) return REART_ERR_UNSUPPORTED;
    if (E == 0 || nq == 0) return REART_OK;
    if (!targets || !queries || !dists || !idx || !workspace) return REART_ERR_INVALID_ARG;
    if (workspace_bytes < reart_grid_knn_workspace_bytes(E, Nt_max)) return REART_ERR_INVALID_ARG;
    const int stride = (int)reart_align_up((size_t)Nt_max, 64);
    GridBuildArgs b = {};
    reart_grid_layout(workspace, E, stride, &b);
    b.pts = targets; b.offsets = offsets; b.N = Nt_max;
    int rc = reart_grid_build_launch(b, E, (hipStream_t)stream);
    if (rc != REART_OK) return rc;
    GridQueryArgs q = {};
    q.q = queries; q.nq = nq; q.E = E; q.stride = stride; q.gx = b.gx; q.gy = b.gy; q.gz = b.gz; q.gorig = b.gorig;
    q.cell_start = b.cell_start; q.meta = b.meta; q.od = dists; q.oi = idx;
    return reart_grid_query_launch(q, K, (hipStream_t)stream);
}
