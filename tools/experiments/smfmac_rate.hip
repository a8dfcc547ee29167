// Issue rate of v_smfmac_f32_16x16x64_f16 against v_mfma_f32_16x16x32_f16 (gfx950): 8 waves per CU, independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int SPARSE, int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int l = threadIdx.x;
  h8 a; h16 b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(l + i);
  for (int i = 0; i < 16; ++i) b[i] = (_Float16)(l - i);
  h8 b8 = __builtin_shufflevector(b, b, 0, 1, 2, 3, 4, 5, 6, 7);
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  const int idx = 0x4444;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (SPARSE) acc[i % NACC] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, acc[i % NACC], idx, 0, 0);
      else acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b8, acc[i % NACC], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + l] = s;
}
int main() {
  float* out;
  if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
  const int iters = 20000;
  for (int sp = 0; sp < 2; ++sp)
    for (int na = 1; na <= 8; na *= 2) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, 0);
#define GO(S_, N_) hipLaunchKernelGGL((k<S_, N_>), dim3(256), dim3(512), 0, 0, out, iters)
      if (sp) { if (na == 1) GO(1, 1); else if (na == 2) GO(1, 2); else if (na == 4) GO(1, 4); else GO(1, 8); }
      else { if (na == 1) GO(0, 1); else if (na == 2) GO(0, 2); else if (na == 4) GO(0, 4); else GO(0, 8); }
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double n = 256.0 * 8 * 8 * iters;      // wave-level instructions
      printf("%s, %d independent accumulators per wave: %.3f ms, %.2f ns per instruction per SIMD-pair (2 waves/SIMD), dense-equivalent %.0f TFLOP/s\n", sp ? "smfmac 16x16x64" : "mfma   16x16x32", na,
             ms, ms * 1e6 / (8.0 * iters * 2), n * (sp ? 2 : 1) * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
