// tools/proto_nn.hip -- design prototype: brute-force 1-NN, AoS scalar-load vs SoA packed-fp32,
// B=38 (19 frames x 2 directions) x N=4096.  Timing only; the product kernels live in reart_amd/csrc.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sqd(float ax,float ay,float az,float bx,float by,float bz){
  float dx=ax-bx, dy=ay-by, dz=az-bz; return (dx*dx+dy*dy)+dz*dz; }

template<int UB, int BS>
__global__ __launch_bounds__(BS) void nn1_aos(const float* __restrict__ qa, const float* __restrict__ ta, int P, int S, float* __restrict__ od, int* __restrict__ oi) {
  int b = blockIdx.y, s = blockIdx.z; int len = P/S; int j0 = s*len, j1 = j0+len;
  const float* q = qa + (size_t)b*P*3; const float* t = ta + (size_t)b*P*3;
  int i = blockIdx.x*BS + threadIdx.x;
  float qx=q[3*i],qy=q[3*i+1],qz=q[3*i+2];
  float best = __builtin_inff(); int blk=j0;
  for (int j=j0;j+UB<=j1;j+=UB){
    float m = __builtin_inff();
    #pragma unroll
    for(int u=0;u<UB;++u) m = fminf(m, sqd(qx,qy,qz,t[3*(j+u)],t[3*(j+u)+1],t[3*(j+u)+2]));
    if (m<best){best=m;blk=j;}
  }
  int bi=blk;
  for(int u=UB-1;u>=0;--u){ float d = sqd(qx,qy,qz,t[3*(blk+u)],t[3*(blk+u)+1],t[3*(blk+u)+2]); if (d==best) bi=blk+u; }
  od[((size_t)s*gridDim.y+b)*P+i]=best; oi[((size_t)s*gridDim.y+b)*P+i]=bi;
}

template<int UB, int BS>
__global__ __launch_bounds__(BS) void nn1_soa(const float* __restrict__ qa, const float* __restrict__ tsoa, int P, int S, float* __restrict__ od, int* __restrict__ oi) {
  int b = blockIdx.y, s = blockIdx.z; int len = P/S; int j0 = s*len, j1 = j0+len;
  const float* tx = tsoa + (size_t)b*P*3; const float* ty = tx + P; const float* tz = ty + P;
  const float* q = qa + (size_t)b*P*3;
  int i = blockIdx.x*BS + threadIdx.x;
  float qx=q[3*i],qy=q[3*i+1],qz=q[3*i+2];
  f2 qx2={qx,qx},qy2={qy,qy},qz2={qz,qz};
  float best = __builtin_inff(); int blk=j0;
  for (int j=j0;j+UB<=j1;j+=UB){
    float m = __builtin_inff();
    #pragma unroll
    for(int u=0;u<UB;u+=2){
      f2 dx = qx2 - *(const f2*)(tx+j+u); f2 dy = qy2 - *(const f2*)(ty+j+u); f2 dz = qz2 - *(const f2*)(tz+j+u);
      f2 d = (dx*dx+dy*dy)+dz*dz; m = fminf(fminf(m,d.x),d.y);
    }
    if (m<best){best=m;blk=j;}
  }
  int bi=blk;
  for(int u=UB-1;u>=0;--u){ float d = sqd(qx,qy,qz,tx[blk+u],ty[blk+u],tz[blk+u]); if (d==best) bi=blk+u; }
  od[((size_t)s*gridDim.y+b)*P+i]=best; oi[((size_t)s*gridDim.y+b)*P+i]=bi;
}

__global__ void to_soa(const float* __restrict__ a, float* __restrict__ s, int P) {
  int b = blockIdx.y; int i = blockIdx.x*256+threadIdx.x; if (i>=P) return;
  const float* p = a + ((size_t)b*P+i)*3; float* o = s + (size_t)b*P*3;
  o[i]=p[0]; o[P+i]=p[1]; o[2*P+i]=p[2];
}

int main(){
  const int B=38, P=4096; size_t n=(size_t)B*P*3;
  std::vector<float> hq(n), ht(n); srand(2);
  for(size_t k=0;k<n;++k){ hq[k]=rand()/(float)RAND_MAX*0.7f-0.35f; ht[k]=rand()/(float)RAND_MAX*0.7f-0.35f; }
  float *dq,*dt,*ds,*od; int* oi; const int SMAX=8;
  CK(hipMalloc(&dq,n*4)); CK(hipMalloc(&dt,n*4)); CK(hipMalloc(&ds,n*4)); CK(hipMalloc(&od,(size_t)SMAX*B*P*4)); CK(hipMalloc(&oi,(size_t)SMAX*B*P*4));
  CK(hipMemcpy(dq,hq.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(dt,ht.data(),n*4,hipMemcpyHostToDevice));
  to_soa<<<dim3(P/256,B),256>>>(dt,ds,P); CK(hipDeviceSynchronize());
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> r0((size_t)B*P), r1((size_t)B*P); std::vector<int> i0((size_t)B*P), i1((size_t)B*P);
  auto timeit=[&](const char* name, auto launch)->int{
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for(int r=0;r<20;++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); double us=ms*1e3/20; double pairs=(double)B*P*P;
    printf("%-28s %9.2f us  %.2f Gpair/s  (8 flop/pair: %.1f TFLOP/s = %.1f%% of 157.3)\n",name,us,pairs/us*1e-3,pairs*8/us*1e-6,pairs*8/us*1e-6/157.3*100); return 0; };
  for (int S : {1,2,4,8}) {
    char nm[64];
    snprintf(nm,64,"aos UB8 BS256 S=%d",S); timeit(nm,[&]{ nn1_aos<8,256><<<dim3(P/256,B,S),256>>>(dq,dt,P,S,od,oi); });
    snprintf(nm,64,"aos UB8 BS64  S=%d",S); timeit(nm,[&]{ nn1_aos<8,64><<<dim3(P/64,B,S),64>>>(dq,dt,P,S,od,oi); });
    snprintf(nm,64,"soa UB16 BS256 S=%d",S); timeit(nm,[&]{ nn1_soa<16,256><<<dim3(P/256,B,S),256>>>(dq,ds,P,S,od,oi); });
    snprintf(nm,64,"soa UB16 BS64  S=%d",S); timeit(nm,[&]{ nn1_soa<16,64><<<dim3(P/64,B,S),64>>>(dq,ds,P,S,od,oi); });
    snprintf(nm,64,"soa UB32 BS64  S=%d",S); timeit(nm,[&]{ nn1_soa<32,64><<<dim3(P/64,B,S),64>>>(dq,ds,P,S,od,oi); });
  }
  timeit("to_soa", [&]{ to_soa<<<dim3(P/256,B),256>>>(dt,ds,P); });
  // cross-check aos vs soa (S=1)
  nn1_aos<8,256><<<dim3(P/256,B,1),256>>>(dq,dt,P,1,od,oi); CK(hipMemcpy(r0.data(),od,r0.size()*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(i0.data(),oi,i0.size()*4,hipMemcpyDeviceToHost));
  nn1_soa<16,64><<<dim3(P/64,B,1),64>>>(dq,ds,P,1,od,oi); CK(hipMemcpy(r1.data(),od,r1.size()*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(i1.data(),oi,i1.size()*4,hipMemcpyDeviceToHost));
  size_t bad=0; for(size_t k=0;k<r0.size();++k) if(r0[k]!=r1[k]||i0[k]!=i1[k]) ++bad;
  // host check of a few queries
  size_t hbad=0; for(int k=0;k<200;++k){ size_t qi=(size_t)(rand()%(B*P)); int b=qi/P; float best=INFINITY; int bi=0;
    for(int j=0;j<P;++j){ float dx=hq[qi*3]-ht[((size_t)b*P+j)*3],dy=hq[qi*3+1]-ht[((size_t)b*P+j)*3+1],dz=hq[qi*3+2]-ht[((size_t)b*P+j)*3+2]; float d=(dx*dx+dy*dy)+dz*dz; if(d<best){best=d;bi=j;} }
    if(best!=r0[qi]||bi!=i0[qi]) ++hbad; }
  printf("aos-vs-soa mismatches: %zu ; host spot-check mismatches: %zu / 200\n", bad, hbad);
  return 0;
}
