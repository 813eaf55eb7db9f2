// VALU throughput probe: v_dot2c_f32_f16 / v_and / v_fma in dependent vs independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 as_h2(unsigned u){ return __builtin_bit_cast(h2,u);} 
template<int CH, int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* in, float* out, int iters){
  unsigned w = in[threadIdx.x]; h2 x = as_h2(in[threadIdx.x+256]);
  float acc[CH]; unsigned ww[CH];
#pragma unroll
  for(int c=0;c<CH;c++){acc[c]=0.f; ww[c]=w+c;}
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int u=0;u<8;u++){
#pragma unroll
      for(int c=0;c<CH;c++){
        if (MODE==0) acc[c] = __builtin_amdgcn_fdot2(as_h2(ww[c]), x, acc[c], false);           // dot2 only
        else if (MODE==1) acc[c] = __builtin_amdgcn_fdot2(as_h2(ww[c] & (0x00030003u<<(2*(u&3)))), x, acc[c], false); // and+dot2
        else if (MODE==2) acc[c] = fmaf(__uint_as_float(ww[c]), 1.0001f, acc[c]);                // fma
        else if (MODE==3) { ww[c] = (ww[c] & 0x7fffffffu) + 3u; }                                  // and+add int
      }
    }
  }
  float s=0; 
#pragma unroll
  for(int c=0;c<CH;c++) s+=acc[c] + (float)ww[c];
  out[blockIdx.x*256+threadIdx.x]=s;
}
template<int CH,int MODE> void run(const char* name, int blocks_per_cu, const unsigned* in, float* out){
  int iters=2000; int grid=256*blocks_per_cu;
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<CH,MODE><<<grid,256>>>(in,out,iters); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<CH,MODE><<<grid,256>>>(in,out,iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms,e0,e1);
  double ops = (double)grid*256*iters*8*CH;  // lane-ops of the main instruction
  double per_cu_clk = ops/(ms*1e-3)/256/2.4e9;
  printf("%-10s CH=%d waves/SIMD=%d : %.1f T lane-ops/s  = %.1f lanes/clk/CU (at 2.4GHz)\n", name, CH, blocks_per_cu, ops/ms/1e9, per_cu_clk);
}
int main(){ unsigned* in; float* out; (void)hipMalloc(&in,4096); (void)hipMemset(in,0x11,4096); (void)hipMalloc(&out,256*8*256*4);
  for(int bpc : {1,2,4}){
    run<1,0>("dot2",bpc,in,out); run<4,0>("dot2",bpc,in,out); run<8,0>("dot2",bpc,in,out);
    run<1,1>("and+dot2",bpc,in,out); run<4,1>("and+dot2",bpc,in,out); run<8,1>("and+dot2",bpc,in,out);
    run<1,2>("fma",bpc,in,out); run<4,2>("fma",bpc,in,out); run<8,2>("fma",bpc,in,out);
    run<4,3>("and+add",bpc,in,out);
  }
  return 0; }
