// Issue cost of the vector instructions a three-way bf16 split can be built from, one wave per SIMD, independent streams.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/x3_valu_rate.hip -o /tmp/x3_valu_rate && /tmp/x3_valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float fl2 __attribute__((ext_vector_type(2)));
#define N 64
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, unsigned nl_, unsigned nh_) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.37f + i;
  unsigned nl = nl_, nh = nh_;
  asm volatile("" : "+s"(nl), "+s"(nh));
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      if (OP == 0) {          // v_cvt_pk_bf16_f32
        fl2 f = {v[i], v[i + 1]};
        bf2 h = __builtin_convertvector(f, bf2);
        v[i] = __uint_as_float(__builtin_bit_cast(unsigned, h));
      } else if (OP == 1) {   // 2 x v_dot2c_f32_bf16
        bf2 h = __builtin_bit_cast(bf2, __float_as_uint(v[i]) | 0x3f803f80u);
        v[i] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf2, nl), v[i], false);
        v[i + 1] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf2, nh), v[i + 1], false);
      } else if (OP == 2) {   // and + shl + 2 sub
        unsigned p = __float_as_uint(v[i]);
        v[i] = v[i] - __uint_as_float(p << 16);
        v[i + 1] = v[i + 1] - __uint_as_float(p & 0xffff0000u);
      } else if (OP == 3) {   // 2 x v_add_f32
        v[i] += 1.5f;
        v[i + 1] += 2.5f;
      } else if (OP == 4) {   // v_pk_add_f32
        fl2 f = {v[i], v[i + 1]};
        fl2 g = {1.5f, 2.5f};
        f = f - g * f;        // pk_fma
        v[i] = f.x; v[i + 1] = f.y;
      } else if (OP == 5) {   // v_perm_b32
        v[i] = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(v[i]), __float_as_uint(v[i + 1]), 0x07060302u));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  const char* names[] = {"v_cvt_pk_bf16_f32 (1/pair)", "v_dot2c_f32_bf16 (2/pair)", "shl+and+2 sub (4/pair)", "v_add_f32 (2/pair)", "v_pk_fma_f32 (1/pair)", "v_perm_b32 (1/pair)"};
  for (int waves = 1; waves <= 2; ++waves)
  for (int op = 0; op < 6; ++op) {
    dim3 g(256), b(256 * waves);
    for (int r = 0; r < 2; ++r) {
      switch (op) {
        case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, cyc, 0x0000bf80u, 0xbf800000u); break;
        case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, cyc, 0x0000bf80u, 0xbf800000u); break;
        case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, cyc, 0x0000bf80u, 0xbf800000u); break;
        case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, cyc, 0x0000bf80u, 0xbf800000u); break;
        case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, cyc, 0x0000bf80u, 0xbf800000u); break;
        case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, out, cyc, 0x0000bf80u, 0xbf800000u); break;
      }
      hipDeviceSynchronize();
    }
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    printf("%d wave(s)/SIMD  %-28s %.2f cycles per pair-step (8 pair-steps x 256 iterations: %.0f cycles)\n", waves, names[op], m / (256.0 * 8), m);
  }
  return 0;
}
