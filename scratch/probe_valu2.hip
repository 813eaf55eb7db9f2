#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 as_h2(unsigned u){ return __builtin_bit_cast(h2,u);} 
__device__ __forceinline__ s2 as_s2(unsigned u){ return __builtin_bit_cast(s2,u);} 
template<int CH, int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* in, float* out, int iters){
  unsigned w = in[threadIdx.x]; unsigned xb = in[threadIdx.x+256];
  float acc[CH]; unsigned ww[CH]; h2 hacc[CH]; int iacc[CH];
#pragma unroll
  for(int c=0;c<CH;c++){acc[c]=0.f; ww[c]=w+c; hacc[c]=as_h2(0u); iacc[c]=0;}
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int u=0;u<8;u++){
#pragma unroll
      for(int c=0;c<CH;c++){
        if (MODE==0) acc[c] = __builtin_amdgcn_fdot2(as_h2(ww[c]), as_h2(xb), acc[c], false);
        else if (MODE==1) acc[c] = __builtin_amdgcn_fdot2_f32_bf16(as_s2(ww[c]), as_s2(xb), acc[c], false);
        else if (MODE==2) hacc[c] = as_h2(ww[c]) * as_h2(xb) + hacc[c];      // v_pk_fma_f16
        else if (MODE==3) iacc[c] = __builtin_amdgcn_sdot4(ww[c], xb, iacc[c], false);
        else if (MODE==4) iacc[c] = __builtin_amdgcn_sdot8(ww[c], xb, iacc[c], false);
        else if (MODE==5) ww[c] = (ww[c] & 0x00030003u) | (0x43004300u + u);   // v_and_or
        else if (MODE==6) acc[c] = fmaf(__uint_as_float(ww[c]), 1.0001f, acc[c]);
      }
    }
  }
  float s=0; 
#pragma unroll
  for(int c=0;c<CH;c++) s+=acc[c] + (float)ww[c] + (float)hacc[c].x + (float)iacc[c];
  out[blockIdx.x*256+threadIdx.x]=s;
}
template<int CH,int MODE> void run(const char* name, int bpc, const unsigned* in, float* out){
  int iters=2000; int grid=256*bpc;
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<CH,MODE><<<grid,256>>>(in,out,iters); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<CH,MODE><<<grid,256>>>(in,out,iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms,e0,e1);
  double ops = (double)grid*256*iters*8*CH;
  printf("%-12s CH=%d waves/SIMD=%d : %.1f T lane-ops/s\n", name, CH, bpc, ops/ms/1e9);
}
int main(){ unsigned* in; float* out; (void)hipMalloc(&in,4096); (void)hipMemset(in,0x11,4096); (void)hipMalloc(&out,256*8*256*4);
  for(int bpc : {2,4}){
    run<4,0>("dot2_f16",bpc,in,out); run<4,1>("dot2_bf16",bpc,in,out); run<4,2>("pk_fma_f16",bpc,in,out);
    run<4,3>("dot4_i8",bpc,in,out); run<4,4>("dot8_i4",bpc,in,out); run<4,5>("and_or",bpc,in,out); run<4,6>("fma_f32",bpc,in,out);
  }
  return 0; }
