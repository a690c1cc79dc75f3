#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64*64];   // [row][col] 64x64
  for (int i = threadIdx.x; i < 64*64; i += 64) tile[i] = in[i];
  __syncthreads();
  int l = threadIdx.x; int i16 = l & 15, g = l >> 4; int q = i16 >> 2, p = i16 & 3;
  // block: rows 8g..8g+3, cols 0..15
  const unsigned short* addr = &tile[(8*g + q)*64 + 4*p];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
  for (int e = 0; e < 4; ++e) out[l*4+e] = (unsigned short)v[e];
}
int main(){ unsigned short h[4096], *d, *o, r[256]; for(int i=0;i<4096;i++) h[i]=i; hipMalloc(&d,8192); hipMalloc(&o,512); hipMemcpy(d,h,8192,hipMemcpyHostToDevice);
 k<<<1,64>>>(d,o); hipMemcpy(r,o,512,hipMemcpyDeviceToHost); for(int l=0;l<64;l+= 1){ if(l<20||l%16==0) printf("lane %d: %d %d %d %d\n",l,r[l*4],r[l*4+1],r[l*4+2],r[l*4+3]);} return 0; }
