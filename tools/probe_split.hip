// Probe (round 3): can the f16 matrix pipe reproduce an fp32 GEMM when every operand is split into hi + lo halves?
//   x = H + L,  H = f16(x),  L = f16(x - H)      (22 significant bits, exact residual in fp32)
//   a*b ~= aH*bH + aH*bL + aL*bH                  (3 v_mfma_f32_32x32x16_f16, fp32 accumulate; aL*bL ~ 2^-22 dropped)
// Compares against an fp64 host reference, beside the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) and a 3-way bf16 split
// (6 products).  Also runs on tiny-magnitude data (scale 1e-6) to see what f16 subnormals do to the lo halves, with and
// without a power-of-two prescale.   Build: hipcc --offload-arch=gfx950 -O3 -o probe_split probe_split.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

constexpr int M = 32, N = 32;

// A [M][K] row-major, B [K][N] row-major; one wave computes the 32x32 tile.
__global__ void k_f32(const float* A, const float* B, float* D, int K) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * N + r], acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * N + r] = acc[i];
}

__device__ inline void split_h(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// mode 0: 3 products (hh, hl, lh); mode 1: 4 products (+ ll); mode 2: hh only (plain f16)
__global__ void k_h2(const float* A, const float* B, float* D, int K, float sa, float sb, int mode) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 16) {
    h8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      _Float16 x, y;
      split_h(A[r * K + k + 8 * h + j] * sa, x, y);
      ah[j] = x; al[j] = y;
      split_h(B[(k + 8 * h + j) * N + r] * sb, x, y);
      bh[j] = x; bl[j] = y;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    if (mode != 2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    }
    if (mode == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
  }
  const float inv = 1.f / (sa * sb);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * N + r] = acc[i] * inv;
}

__global__ void k_b3(const float* A, const float* B, float* D, int K) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 16) {
    b8 a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      float x = A[r * K + k + 8 * h + j], y = B[(k + 8 * h + j) * N + r];
      for (int p = 0; p < 3; ++p) {
        a[p][j] = (__bf16)x; x -= (float)a[p][j];
        b[p][j] = (__bf16)y; y -= (float)b[p][j];
      }
    }
    const int pa[6] = {0, 0, 1, 0, 2, 1}, pb[6] = {0, 1, 0, 2, 0, 1};
    for (int q = 5; q >= 0; --q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa[q]], b[pb[q]], acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * N + r] = acc[i];
}

// subnormal handling of the conversions and of the MFMA inputs: D[0] = f16 MFMA of (2^-20 as f16 subnormal) * 1.0 summed
__global__ void k_sub(float* out) {
  const int l = threadIdx.x;
  h8 a = {0}, b = {0};
  const float tiny = 9.5367431640625e-07f;   // 2^-20: an f16 subnormal (min normal 2^-14)
  a[0] = (_Float16)tiny;
  b[0] = (_Float16)1.0f;
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (l == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}

static double frand() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

int main() {
  float *dA, *dB, *dD, *dS;
  const int KMAX = 4608;
  hipMalloc(&dA, M * KMAX * 4); hipMalloc(&dB, KMAX * N * 4); hipMalloc(&dD, M * N * 4); hipMalloc(&dS, 64);
  k_sub<<<1, 64>>>(dS);
  float s[2]; hipMemcpy(s, dS, 8, hipMemcpyDeviceToHost);
  printf("subnormal probe: mfma(2^-20 * 1) = %.9g (expect 9.5367e-07; 0 = inputs flushed), cvt round trip = %.9g\n", s[0], s[1]);
  struct Case { const char* name; int K; double sa, sb; int heavy; };
  const Case cases[] = {{"K=288 unit", 288, 1, 1, 0}, {"K=1152 unit", 1152, 1, 1, 0}, {"K=4608 unit", 4608, 1, 1, 0},
                        {"K=1152 a*1e-6", 1152, 1e-6, 1, 0}, {"K=1152 a*1e-6 heavy-tailed", 1152, 1e-6, 1, 1},
                        {"K=1152 weights 0.03", 1152, 1, 0.03, 0}};
  for (const Case& c : cases) {
    srand(1234);
    std::vector<float> A(M * c.K), B(c.K * N);
    for (auto& v : A) { double x = frand(); if (c.heavy) x = x * x * x * x * x * (rand() % 50 == 0 ? 300.0 : 1.0); v = (float)(x * c.sa); }
    for (auto& v : B) v = (float)(frand() * c.sb);
    std::vector<double> R(M * N, 0.0), S(M * N, 0.0);
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) {
        double acc = 0, sab = 0;
        for (int k = 0; k < c.K; ++k) { acc += (double)A[i * c.K + k] * B[k * N + j]; sab += fabs((double)A[i * c.K + k] * B[k * N + j]); }
        R[i * N + j] = acc; S[i * N + j] = sab;
      }
    double scale = 0;
    for (double v : R) scale = fmax(scale, fabs(v));
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    auto report = [&](const char* what) {
      std::vector<float> D(M * N);
      hipDeviceSynchronize();
      hipMemcpy(D.data(), dD, M * N * 4, hipMemcpyDeviceToHost);
      double emax = 0, erel = 0;
      for (int i = 0; i < M * N; ++i) { emax = fmax(emax, fabs(D[i] - R[i])); erel = fmax(erel, fabs(D[i] - R[i]) / S[i]); }
      printf("  %-34s max|err|/scale = %.3e   max|err|/sum|ab| = %.3e\n", what, emax / scale, erel);
    };
    printf("%s (|out| scale %.3e)\n", c.name, scale);
    k_f32<<<1, 64>>>(dA, dB, dD, c.K); report("f32 MFMA (exact fp32 chain)");
    k_h2<<<1, 64>>>(dA, dB, dD, c.K, 1.f, 1.f, 0); report("f16x2, 3 products, no prescale");
    k_h2<<<1, 64>>>(dA, dB, dD, c.K, 1.f, 1.f, 1); report("f16x2, 4 products, no prescale");
    if (c.sa != 1) { k_h2<<<1, 64>>>(dA, dB, dD, c.K, 1048576.f * 16.f, 1.f, 0); report("f16x2, 3 products, a prescaled 2^24"); }
    if (c.sb != 1) { k_h2<<<1, 64>>>(dA, dB, dD, c.K, 1.f, 32.f, 0); report("f16x2, 3 products, b prescaled 2^5"); }
    k_h2<<<1, 64>>>(dA, dB, dD, c.K, 1.f, 1.f, 2); report("plain f16 (hh only)");
    k_b3<<<1, 64>>>(dA, dB, dD, c.K); report("bf16x3, 6 products");
  }
  return 0;
}
