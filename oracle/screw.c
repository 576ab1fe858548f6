/*
 * oracle/screw.c -- TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of
 *   se3_exp_map / _so3_exp_map / _se3_V_matrix     screw_se3/geo_utils.py:90-222
 *   screw_param_to_exponential_coordinates          screw_se3/screw_utils.py:6-23
 *   transform_from_exponential_coordinates          screw_se3/screw_utils.py:27-30
 *   fk                                              utils/kinematic_utils.py:151-198
 * PINNED by tests/golden/se3.npz and kinematic.npz (reference Python).
 *
 * Reproduced on purpose: the clamp is on the SQUARED rotation norm at 1e-4, the
 * no-rotation test |theta| < 1e-6 is strict and in fp32, and revolute joints carry d = 1e-6.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

static void mat3_mul(const float *A, const float *B, float *C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = fmaf(A[3 * i + 2], B[6 + j], fmaf(A[3 * i + 1], B[3 + j], A[3 * i] * B[j]));
}

/* rotation R and translation tr of exp([v | w]); geo_utils.py:90-144, 202-222 */
static void se3_exp_core(const float *v, const float *w, float *R, float *tr) {
    const float n2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    const float ph = sqrtf(n2 < 1e-4f ? 1e-4f : n2);
    const float inv = 1.0f / ph;
    const float s = sinf(ph), c = cosf(ph);
    const float fac1 = inv * s, fac2 = inv * inv * (1.0f - c);
    const float K[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
    float K2[9];
    mat3_mul(K, K, K2);
    const float bV = (1.0f - c) / (ph * ph), cV = (ph - s) / (ph * ph * ph);
    float V[9];
    for (int i = 0; i < 9; ++i) {
        const float id = (i % 4 == 0) ? 1.0f : 0.0f;
        R[i] = (fac1 * K[i] + fac2 * K2[i]) + id;
        V[i] = (id + K[i] * bV) + K2[i] * cV;
    }
    for (int i = 0; i < 3; ++i)
        tr[i] = fmaf(V[3 * i + 2], v[2], fmaf(V[3 * i + 1], v[1], V[3 * i] * v[0]));
}

/* screw_se3/geo_utils.py:147-222: input rows [log_translation | log_rotation]; output is the
 * pytorch3d row-vector form [[R^T, 0], [T, 1]] (the function returns transform.permute(0,2,1)) */
void oracle_se3_exp_map(const float *log_transform, int n, float *T44) {
    for (int e = 0; e < n; ++e) {
        float R[9], tr[3];
        se3_exp_core(log_transform + 6 * e, log_transform + 6 * e + 3, R, tr);
        float *T = T44 + 16 * (size_t)e;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * j + i];
            T[4 * i + 3] = 0.f;
            T[12 + i] = tr[i];
        }
        T[15] = 1.f;
    }
}

/* screw_utils.py:6-30 composed -> column-vector [[R, t], [0, 1]] */
static void screw_one(const float *l, const float *m, float theta, float d, float *T) {
    const int no_rot = (fabsf(theta) < 1e-6f) || (fabsf(theta - 3.14159265358979323846f) < 1e-6f);
    float w[3], v[3];
    if (!no_rot) {
        const float q[3] = {l[1] * m[2] - l[2] * m[1], l[2] * m[0] - l[0] * m[2], l[0] * m[1] - l[1] * m[0]};
        const float h = d / theta;
        const float ql[3] = {q[1] * l[2] - q[2] * l[1], q[2] * l[0] - q[0] * l[2], q[0] * l[1] - q[1] * l[0]};
        for (int c = 0; c < 3; ++c) { w[c] = l[c]; v[c] = ql[c] + h * l[c]; }
    } else {
        for (int c = 0; c < 3; ++c) { w[c] = 0.f; v[c] = l[c]; }
    }
    float om[3], u[3], R[9], tr[3];
    for (int c = 0; c < 3; ++c) { om[c] = w[c] * theta; u[c] = v[c] * theta; }
    se3_exp_core(u, om, R, tr);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = tr[i];
    }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
}

void oracle_screw_to_transform(const float *l, const float *m, const float *theta,
                               const float *d, int n, float *T44) {
    for (int e = 0; e < n; ++e) screw_one(l + 3 * e, m + 3 * e, theta[e], d[e], T44 + 16 * (size_t)e);
}

/* utils/kinematic_utils.py:151-198.  The joint tree is given as arrays: parent[c] (-1 for the
 * root), edge_of_part[c] = index of the edge "c_parent" in axis/moment/theta, order = parts from
 * root to leaf (reverse_topo).  Because parents precede children in `order`, the reference's
 * path walk stops at the first edge (:188-191): FK[c] = FK[parent] * T_rel(edge c).
 * theta [B,E]; distance [B,E] or NULL (= 1e-6, :176); trans [B,P,4,4]. */
void oracle_fk(const int32_t *parent, const int32_t *edge_of_part,
               const int32_t *order, int P,
               const float *axis, const float *moment, const float *theta,
               const float *distance, int B, int E, float *trans) {
    for (int t = 0; t < B; ++t)
        for (int oi = 0; oi < P; ++oi) {
            const int c = order[oi];
            float *F = trans + 16 * ((size_t)t * P + c);
            if (parent[c] < 0) {
                memset(F, 0, 64);
                F[0] = F[5] = F[10] = F[15] = 1.f;
                continue;
            }
            const int e = edge_of_part[c];
            float Tr[16];
            screw_one(axis + 3 * e, moment + 3 * e, theta[(size_t)t * E + e],
                      distance ? distance[(size_t)t * E + e] : 1e-6f, Tr);
            const float *Fp = trans + 16 * ((size_t)t * P + parent[c]);
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    float acc = Fp[4 * i] * Tr[j];
                    for (int k = 1; k < 4; ++k) acc = fmaf(Fp[4 * i + k], Tr[4 * k + j], acc);
                    F[4 * i + j] = acc;
                }
        }
}
