// What one MI355X SUSTAINS on the bf16 matrix pipe (tools/experiments: measurement aid, not part of the library).
// Bare loops of v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16, operands in registers (or the B operand re-read from LDS by
// ds_read_b128 as the x3 kernels do), one workgroup of 8 waves per CU on all 256 CUs, on ZERO and on RANDOM operands, each
// configuration run back to back for >= 1.5 s so that the chip settles on the clock it holds under that load (MI355X_MICROARCH.md,
// "DVFS give-back").  Prints executed TFLOP/s and the fraction of the nominal 2516.8 TFLOP/s: the figure the x3 kernels'
// `mfma_frac` should be read against.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_sustained.hip -o tools/_bin/mfma_sustained && tools/_bin/mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// SHAPE 0: 16x16x32, 1: 32x32x16.  LDSB 1: every B fragment comes from LDS (one ds_read_b128 per 6 MFMAs, the x3 kernels' ratio)
template <int SHAPE, int LDSB>
__global__ __launch_bounds__(512, 2) void k(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ uint4 tile[4096];       // 64 KB
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 512) tile[i] = src[(blockIdx.x * 4096 + i) & 0xffff];
  __syncthreads();
  uint4 a[6], b[3];
  for (int i = 0; i < 6; ++i) a[i] = src[(blockIdx.x * 512 + l + 977 * i) & 0xffff];
  for (int i = 0; i < 3; ++i) b[i] = src[(blockIdx.x * 512 + l + 131 * i + 7) & 0xffff];
  f4 acc4[8];
  f16v acc16[4];
  for (int i = 0; i < 8; ++i) acc4[i] = f4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int c = 0; c < 16; ++c) acc16[i][c] = 0.f;
  int off = l;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (LDSB) {
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = tile[(off + 512 * p + 64 * g) & 4095];
      }
      // six products of one (filter planes, pixel planes) pairing: a0 b0, a1 b0, a2 b0, a3 b1, a4 b1, a5 b2
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const uint4 bb = b[i < 3 ? 0 : (i < 5 ? 1 : 2)];
        if (SHAPE == 0) acc4[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a[i]), __builtin_bit_cast(b8, bb), acc4[g], 0, 0, 0);
        else if (i % 2 == 0) acc16[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a[i]), __builtin_bit_cast(b8, bb), acc16[g & 3], 0, 0, 0);
      }
    }
    off += 64;
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
  for (int i = 0; i < 4; ++i)
    for (int c = 0; c < 16; ++c) s += acc16[i][c];
  out[blockIdx.x * 512 + l] = s;
}

int main() {
  uint4* src;
  float* out;
  const size_t n = 65536;
  if (hipMalloc(&src, n * 16) != hipSuccess || hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
  std::vector<uint16_t> h(n * 8);
  const double peak = 2516.8;
  for (int data = 0; data < 2; ++data) {
    srand(1);
    for (auto& v : h) {
      // random bf16 in [-2, 2): sign, exponent 120..127, random 7-bit mantissa -- or all zeros
      v = data ? (uint16_t)(((rand() & 1) << 15) | ((120 + (rand() & 7)) << 7) | (rand() & 127)) : 0;
    }
    (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 4; ++cfg) {
      const int shape = cfg & 1, ldsb = cfg >> 1;
      const int iters = 4000;
      auto go = [&]() {
        if (cfg == 0) hipLaunchKernelGGL((k<0, 0>), dim3(256), dim3(512), 0, 0, src, out, iters);
        if (cfg == 1) hipLaunchKernelGGL((k<1, 0>), dim3(256), dim3(512), 0, 0, src, out, iters);
        if (cfg == 2) hipLaunchKernelGGL((k<0, 1>), dim3(256), dim3(512), 0, 0, src, out, iters);
        if (cfg == 3) hipLaunchKernelGGL((k<1, 1>), dim3(256), dim3(512), 0, 0, src, out, iters);
      };
      // flops of one launch: 256 CUs x 8 waves x iters x 8 groups x (6 x 16x16x32 or 3 x 32x32x16) x 2
      const double fl = 256.0 * 8 * iters * 8 * (shape ? 3.0 * 32 * 32 * 16 : 6.0 * 16 * 16 * 32) * 2.0;
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      go();
      (void)hipDeviceSynchronize();
      // settle: ~1.5 s of back-to-back launches, then time the last 20
      float ms1 = 0;
      (void)hipEventRecord(e0, 0);
      go();
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms1, e0, e1);
      const int settle = (int)(1500.0 / ms1) + 1;
      for (int i = 0; i < settle; ++i) go();
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) go();
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double tf = fl * 20 / (ms * 1e-3) / 1e12;
      printf("%-6s operands, %s, B %-14s: first launch %.3f ms, settled %.3f ms per launch = %7.1f TFLOP/s = %.3f of the nominal %.1f\n",
             data ? "random" : "zero", shape ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", ldsb ? "from LDS (b128)" : "in registers", ms1, ms / 20, tf,
             tf / peak, peak);
      fflush(stdout);
    }
  }
  return 0;
}
