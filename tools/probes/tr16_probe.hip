// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS[i] = i (u16); every lane supplies an
// address; print which LDS elements each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_ptr;
__global__ void k(unsigned short* out, int pitch, int mode) {
  __shared__ unsigned short lds[8192];
  int l = threadIdx.x;
  for (int i = l; i < 8192; i += 64) lds[i] = i;
  __syncthreads();
  int g = l >> 4, i = l & 15;
  const unsigned short* p;
  if (mode == 0) p = lds + (8 * g + (i >> 2)) * pitch + 4 * (i & 3);   // my assumed pattern
  else p = lds + l * 4;                                                  // linear
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    int pitch = 64;
    k<<<1, 64>>>(d, pitch, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d pitch %d (value = row*pitch+col => printed as row:col)\n", mode, pitch);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" %3d:%-2d", h[l * 4 + j] / pitch, h[l * 4 + j] % pitch);
      printf("\n");
    }
  }
  return 0;
}
