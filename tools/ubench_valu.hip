// tools/ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (design input
// for the Chamfer/K-NN kernel: are packed-fp32 ops worth shaping the data for?).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_valu.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
template<int MODE>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters, float seed) {
  float a0=seed+threadIdx.x, a1=a0+1, a2=a0+2, a3=a0+3, a4=a0+4, a5=a0+5, a6=a0+6, a7=a0+7;
  f2 p0={a0,a1},p1={a2,a3},p2={a4,a5},p3={a6,a7},p4={a1,a0},p5={a3,a2},p6={a5,a4},p7={a7,a6};
  float c = 1.0000001f; f2 c2={c,c};
  for (int it=0; it<iters; ++it) {
    if (MODE==0) { // v_mul_f32 x8 independent chains, 8 reps
      REP8(asm volatile("v_mul_f32 %0,%0,%8\n v_mul_f32 %1,%1,%8\n v_mul_f32 %2,%2,%8\n v_mul_f32 %3,%3,%8\n v_mul_f32 %4,%4,%8\n v_mul_f32 %5,%5,%8\n v_mul_f32 %6,%6,%8\n v_mul_f32 %7,%7,%8\n"
        : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));)
    } else if (MODE==1) { // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0,%0,%8\n v_pk_mul_f32 %1,%1,%8\n v_pk_mul_f32 %2,%2,%8\n v_pk_mul_f32 %3,%3,%8\n v_pk_mul_f32 %4,%4,%8\n v_pk_mul_f32 %5,%5,%8\n v_pk_mul_f32 %6,%6,%8\n v_pk_mul_f32 %7,%7,%8\n"
        : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3),"+v"(p4),"+v"(p5),"+v"(p6),"+v"(p7) : "v"(c2));)
    } else if (MODE==2) { // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0,%0,%8\n v_pk_add_f32 %1,%1,%8\n v_pk_add_f32 %2,%2,%8\n v_pk_add_f32 %3,%3,%8\n v_pk_add_f32 %4,%4,%8\n v_pk_add_f32 %5,%5,%8\n v_pk_add_f32 %6,%6,%8\n v_pk_add_f32 %7,%7,%8\n"
        : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3),"+v"(p4),"+v"(p5),"+v"(p6),"+v"(p7) : "v"(c2));)
    } else if (MODE==3) { // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0,%0,%8,%8\n v_fma_f32 %1,%1,%8,%8\n v_fma_f32 %2,%2,%8,%8\n v_fma_f32 %3,%3,%8,%8\n v_fma_f32 %4,%4,%8,%8\n v_fma_f32 %5,%5,%8,%8\n v_fma_f32 %6,%6,%8,%8\n v_fma_f32 %7,%7,%8,%8\n"
        : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));)
    } else if (MODE==4) { // v_min3_f32
      REP8(asm volatile("v_min3_f32 %0,%0,%8,%1\n v_min3_f32 %1,%1,%8,%2\n v_min3_f32 %2,%2,%8,%3\n v_min3_f32 %3,%3,%8,%4\n v_min3_f32 %4,%4,%8,%5\n v_min3_f32 %5,%5,%8,%6\n v_min3_f32 %6,%6,%8,%7\n v_min3_f32 %7,%7,%8,%0\n"
        : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c));)
    } else if (MODE==5) { // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0,%0,%8,%8\n v_pk_fma_f32 %1,%1,%8,%8\n v_pk_fma_f32 %2,%2,%8,%8\n v_pk_fma_f32 %3,%3,%8,%8\n v_pk_fma_f32 %4,%4,%8,%8\n v_pk_fma_f32 %5,%5,%8,%8\n v_pk_fma_f32 %6,%6,%8,%8\n v_pk_fma_f32 %7,%7,%8,%8\n"
        : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3),"+v"(p4),"+v"(p5),"+v"(p6),"+v"(p7) : "v"(c2));)
    } else if (MODE==6) { // v_cndmask_b32 (vcc)
      REP8(asm volatile("v_cndmask_b32 %0,%0,%8,vcc\n v_cndmask_b32 %1,%1,%8,vcc\n v_cndmask_b32 %2,%2,%8,vcc\n v_cndmask_b32 %3,%3,%8,vcc\n v_cndmask_b32 %4,%4,%8,vcc\n v_cndmask_b32 %5,%5,%8,vcc\n v_cndmask_b32 %6,%6,%8,vcc\n v_cndmask_b32 %7,%7,%8,vcc\n"
        : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(c) : "vcc");)
    } else if (MODE==7) { // v_pk_add with SGPR-pair operand (as the NN kernel uses)
      REP8(asm volatile("v_pk_add_f32 %0,%0,%8\n v_pk_add_f32 %1,%1,%8\n v_pk_add_f32 %2,%2,%8\n v_pk_add_f32 %3,%3,%8\n v_pk_add_f32 %4,%4,%8\n v_pk_add_f32 %5,%5,%8\n v_pk_add_f32 %6,%6,%8\n v_pk_add_f32 %7,%7,%8\n"
        : "+v"(p0),"+v"(p1),"+v"(p2),"+v"(p3),"+v"(p4),"+v"(p5),"+v"(p6),"+v"(p7) : "s"(c2));)
    }
  }
  float r = a0+a1+a2+a3+a4+a5+a6+a7 + p0.x+p0.y+p1.x+p1.y+p2.x+p2.y+p3.x+p3.y+p4.x+p5.x+p6.x+p7.x;
  if (r == 12345.678f) out[0] = r;
}

template<int MODE> int run(const char* name, int lanes_ops_per_instr, int wpb_blocks) {
  float* d; CK(hipMalloc(&d, 4));
  int iters = 2000; int blocks = wpb_blocks;
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  valu_kernel<MODE><<<blocks,256>>>(d, 10, 1.0f); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); valu_kernel<MODE><<<blocks,256>>>(d, iters, 1.0f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  double instr = (double)blocks*4 /*waves*/ * iters * 64.0; // wave-instructions
  double winst_per_s = instr/(ms*1e-3);
  // per SIMD per cycle at 2.4GHz: 1024 SIMDs
  double cyc_per_winstr = (1024.0*2.4e9)/winst_per_s;
  printf("%-22s blocks=%5d  %8.3f ms  %.3f Twave-instr/s  => %.2f cycles/wave-instr/SIMD @2.4GHz  (%.1f Tlane-elem-ops/s)\n",
         name, blocks, ms, winst_per_s*1e-12, cyc_per_winstr, winst_per_s*64*lanes_ops_per_instr*1e-12);
  CK(hipFree(d)); return 0;
}

int main() {
  int dev=0; hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  for (int blocks : {256*2, 256*8}) {
    run<0>("v_mul_f32",1,blocks); run<1>("v_pk_mul_f32",2,blocks); run<2>("v_pk_add_f32",2,blocks);
    run<7>("v_pk_add_f32(sgpr)",2,blocks);
    run<3>("v_fma_f32",1,blocks); run<5>("v_pk_fma_f32",2,blocks); run<4>("v_min3_f32",1,blocks); run<6>("v_cndmask_b32",1,blocks);
  }
  return 0;
}
