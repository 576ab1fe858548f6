// How long does a barrier among the G workgroups of one GROUP take when it is an atomic counter in memory (no cooperative-groups
// grid sync: only the group's workgroups meet)?  Groups either sit on ONE XCD each (the dispatcher deals workgroup L to XCD
// L % 8: group = L % 8 + 8 * (L / (8 G)), member = (L / 8) % G) or are spread over all of them (group = L / G).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_group_barrier.hip -o /tmp/ubg && /tmp/ubg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(int G, int rounds, int pinned, unsigned *cnt, unsigned long long *ticks, int *err) {
    const int L = blockIdx.x;
    int group, member;
    if (pinned) { group = (L % 8) + 8 * (L / (8 * G)); member = (L / 8) % G; }
    else { group = L / G; member = L % G; }
    unsigned *c = cnt + 64 * group;                       // one counter per group, its own cache line
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        // ... a round's work would be here ...
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)G * (unsigned)(r + 1);
            int spins = 0;
            while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > (1 << 22)) { s_bad = 1; break; }           // never hang the GPU: give up loudly
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (s_bad) { if (threadIdx.x == 0) *err = 1; return; }
    }
    if (threadIdx.x == 0 && member == 0) ticks[group] = wall_clock64() - t0;
}
int main() {
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    unsigned *cnt; unsigned long long *ticks; int *err;
    hipMalloc(&cnt, 64 * 64 * 4); hipMalloc(&ticks, 64 * 8); hipMalloc(&err, 4);
    const int rounds = 2000;
    for (int pinned = 0; pinned < 2; ++pinned)
        for (int G : {2, 4, 8, 16, 28})
            for (int groups : {8, 9, 16}) {
                if (groups * G > 256) continue;
                const int ngrp = pinned ? ((groups + 7) / 8) * 8 : groups;       // pinned layout needs whole layers of 8 groups
                hipMemset(cnt, 0, 64 * 64 * 4); hipMemset(err, 0, 4);
                hipLaunchKernelGGL(k, dim3(ngrp * G), dim3(256), 0, 0, G, rounds, pinned, cnt, ticks, err);
                std::vector<unsigned long long> h(64);
                int e = 0;
                hipMemcpy(h.data(), ticks, 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
                unsigned long long mx = 0; for (int g = 0; g < ngrp; ++g) mx = h[g] > mx ? h[g] : mx;
                printf("%s G %2d groups %2d: %.2f us per barrier (slowest group)%s\n", pinned ? "one XCD per group " : "spread over XCDs  ", G, ngrp,
                       (double)mx / khz * 1e3 / rounds, e ? "  [TIMED OUT]" : "");
            }
    return 0;
}
