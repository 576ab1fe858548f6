// tools/ubench_mfma.hip -- issue rate of v_mfma_f32_32x32x2_f32 from ONE wave per SIMD (the situation of the extractor's fused
// chains, whose LDS tiles leave room for one workgroup per compute unit): cycles per MFMA with NB independent accumulators,
// with and without an LDS read + wait per group.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/um tools/ubench_mfma.hip && /tmp/um
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NB, int WAVES, bool LDS>
__global__ __launch_bounds__(64 * WAVES) void mfma_kernel(float *out, long long *cyc, int iters, float seed) {
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) sm[i] = seed + i;
    __syncthreads();
    f16v acc[NB];
    for (int n = 0; n < NB; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = seed + threadIdx.x, b[NB];
    for (int n = 0; n < NB; ++n) b[n] = seed + n;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            __builtin_amdgcn_sched_barrier(0);
            a = sm[(threadIdx.x + it) & 4095];
#pragma unroll
            for (int n = 0; n < NB; ++n) b[n] = sm[(threadIdx.x + 64 * n + it) & 4095];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[n], acc[n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int n = 0; n < NB; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NB, int WAVES, bool LDS>
static int run(const char *name) {
    float *out; long long *cyc, h = 0;
    CK(hipMalloc(&out, sizeof(float) * 256 * 64 * WAVES)); CK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    hipLaunchKernelGGL((mfma_kernel<NB, WAVES, LDS>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters, 1.0f);
    hipLaunchKernelGGL((mfma_kernel<NB, WAVES, LDS>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s %7.1f clock64 ticks per MFMA (%d accumulators, %d waves per workgroup = %d per SIMD)\n", name, (double)h / iters / NB, NB, WAVES, WAVES / 4);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    run<4, 4, false>("registers only, 4 acc, 1 wave/SIMD");
    run<8, 4, false>("registers only, 8 acc, 1 wave/SIMD");
    run<4, 8, false>("registers only, 4 acc, 2 waves/SIMD");
    run<4, 4, true>("LDS fragments + wait per group, 4 acc, 1/SIMD");
    run<8, 4, true>("LDS fragments + wait per group, 8 acc, 1/SIMD");
    run<4, 8, true>("LDS fragments + wait per group, 4 acc, 2/SIMD");
    run<8, 8, true>("LDS fragments + wait per group, 8 acc, 2/SIMD");
    return 0;
}
