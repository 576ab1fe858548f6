// Exhaustive check of lap_mw.hip's mw_sqrt against sqrtf over every non-negative finite float (2^31 - 2^23 values).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/check_mw_sqrt.hip -o /tmp/check_mw_sqrt && /tmp/check_mw_sqrt
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float mw_sqrt(float x) {
    if (__builtin_expect(x < 0x1p-96f && x > 0.f, 0)) return sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
    float r = rm <= 0.f ? sm : s;
    r = rp > 0.f ? sp : r;
    return r;
}
__global__ void k(unsigned long long *bad) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long n = 0;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < 0x7f800000ull; u += stride) {
        const float x = __uint_as_float((unsigned)u);
        if (__float_as_uint(mw_sqrt(x)) != __float_as_uint(sqrtf(x))) ++n;
    }
    if (n) atomicAdd(bad, n);
}
int main() {
    unsigned long long *d, h = 0;
    hipMalloc(&d, 8); hipMemcpy(d, &h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mw_sqrt != sqrtf on %llu of %llu inputs\n", h, 0x7f800000ull);
    return h != 0;
}
