// reart_amd/csrc/screw_dev.h -- device helpers shared by kinematic.hip and structure.hip:
// one screw joint (l, m, theta, d) -> rigid transform, i.e. the reference's
//   screw_param_to_exponential_coordinates  screw_se3/screw_utils.py:6-23
//   transform_from_exponential_coordinates  screw_se3/screw_utils.py:27-30
//   se3_exp_map (+ _so3_exp_map, _se3_V_matrix) screw_se3/geo_utils.py:90-222
// composed (clamp on the SQUARED rotation norm at 1e-4, strict fp32 no-rotation test).
#pragma once
#include <math.h>

#define PI_F 3.14159265358979323846f

__device__ __forceinline__ void mat3_mul(const float *A, const float *B, float *C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = fmaf(A[3 * i + 2], B[6 + j], fmaf(A[3 * i + 1], B[3 + j], A[3 * i] * B[j]));
}

// one joint: (l, m, theta, d) -> T = [R | tr] (3x4 row-major, T[4*i+j])
__device__ __forceinline__ void screw_fwd(const float *l, const float *m, float theta, float d, float *T) {
    const bool no_rot = (fabsf(theta) < 1e-6f) || (fabsf(theta - PI_F) < 1e-6f);
    const float q[3] = {l[1] * m[2] - l[2] * m[1], l[2] * m[0] - l[0] * m[2], l[0] * m[1] - l[1] * m[0]};
    const float h = d / theta;
    const float ql[3] = {q[1] * l[2] - q[2] * l[1], q[2] * l[0] - q[0] * l[2], q[0] * l[1] - q[1] * l[0]};
    float om[3], u[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float w = no_rot ? 0.f : l[c];
        const float v = no_rot ? l[c] : ql[c] + h * l[c];
        om[c] = w * theta;
        u[c] = v * theta;
    }
    const float n2 = (om[0] * om[0] + om[1] * om[1]) + om[2] * om[2];
    const float ph = sqrtf(n2 < 1e-4f ? 1e-4f : n2);
    const float inv = 1.0f / ph;
    const float s = sinf(ph), c = cosf(ph);
    const float fac1 = inv * s, fac2 = inv * inv * (1.0f - c);
    const float K[9] = {0.f, -om[2], om[1], om[2], 0.f, -om[0], -om[1], om[0], 0.f};
    float K2[9];
    mat3_mul(K, K, K2);
    const float bV = (1.0f - c) / (ph * ph), cV = (ph - s) / (ph * ph * ph);
    float V[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float id = (i % 4 == 0) ? 1.0f : 0.0f;
        T[4 * (i / 3) + i % 3] = (fac1 * K[i] + fac2 * K2[i]) + id;
        V[i] = (id + K[i] * bV) + K2[i] * cV;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        T[4 * i + 3] = fmaf(V[3 * i + 2], u[2], fmaf(V[3 * i + 1], u[1], V[3 * i] * u[0]));
}

