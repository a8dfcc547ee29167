// Set pooling over the L frames (tf.math.reduce_max(axis=1), reference nets/mj_uwyhNets_ba.py:435,451,463) and
// horizontal pyramid pooling (:468-481), forward and backward.  All HBM-bound: one coalesced pass over the frames.
#include "common.h"

namespace {

constexpr int MAXL = 32;  // frames kept in registers by the backward (L = 25 in the reference)

__global__ void setmax_fwd_kernel(const float4* __restrict__ p, const float4* __restrict__ addend,
                                  float4* __restrict__ m, float4* __restrict__ sum_out, int l, size_t s4) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= s4) return;
  const int b = blockIdx.y;
  const float4* src = p + (size_t)b * l * s4 + e;
  float4 mx = src[0];
  for (int t = 1; t < l; ++t) {
    const float4 v = src[(size_t)t * s4];
    mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
  }
  m[(size_t)b * s4 + e] = mx;
  if (addend) {
    const float4 a = addend[(size_t)b * s4 + e];
    sum_out[(size_t)b * s4 + e] = make_float4(mx.x + a.x, mx.y + a.y, mx.z + a.z, mx.w + a.w);
  }
}

__device__ __forceinline__ float sm_route(float v, float mx, float g, int lrelu) {
  float o = v == mx ? g : 0.f;
  if (lrelu) o *= ugn_lrelu_slope(v);
  return o;
}

// TF reduce_max gradient: the incoming gradient is divided equally among all maxima.
__global__ __launch_bounds__(128) void setmax_bwd_kernel(const float4* __restrict__ p, const float4* __restrict__ dm, float4* __restrict__ out,
                                  int l, size_t s4, int lrelu) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= s4) return;
  const int b = blockIdx.y;
  const float4* src = p + (size_t)b * l * s4 + e;
  float4* dst = out + (size_t)b * l * s4 + e;
  float4 v[MAXL];
#pragma unroll
  for (int t = 0; t < MAXL; ++t)
    if (t < l) v[t] = src[(size_t)t * s4];
  float4 mx = v[0];
#pragma unroll
  for (int t = 1; t < MAXL; ++t)
    if (t < l) { mx.x = fmaxf(mx.x, v[t].x); mx.y = fmaxf(mx.y, v[t].y); mx.z = fmaxf(mx.z, v[t].z); mx.w = fmaxf(mx.w, v[t].w); }
  float4 cnt = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < MAXL; ++t)
    if (t < l) {
      cnt.x += v[t].x == mx.x ? 1.f : 0.f; cnt.y += v[t].y == mx.y ? 1.f : 0.f;
      cnt.z += v[t].z == mx.z ? 1.f : 0.f; cnt.w += v[t].w == mx.w ? 1.f : 0.f;
    }
  const float4 g = dm[(size_t)b * s4 + e];
  const float4 gs = make_float4(g.x / cnt.x, g.y / cnt.y, g.z / cnt.z, g.w / cnt.w);
#pragma unroll
  for (int t = 0; t < MAXL; ++t)
    if (t < l)
      dst[(size_t)t * s4] = make_float4(sm_route(v[t].x, mx.x, gs.x, lrelu), sm_route(v[t].y, mx.y, gs.y, lrelu),
                                        sm_route(v[t].z, mx.z, gs.z, lrelu), sm_route(v[t].w, mx.w, gs.w, lrelu));
}

// ---- HPP ------------------------------------------------------------------------------------------------
// feature rows: for bins in {1,2,4,8,16}: a's strips then s3's strips.  Row offsets of the a-part per level:
__device__ __constant__ int kHppOff[5] = {0, 2, 6, 14, 30};

// one thread per (b, tensor, channel): 256 positions, coalesced over the 128 channels.
__global__ void hpp_fwd_kernel(const float* __restrict__ a, const float* __restrict__ s3, float* __restrict__ feat, int bsz) {
  const int c = threadIdx.x & 127, t = threadIdx.x >> 7;  // 256 threads: tensor 0 = a, 1 = s3
  const int b = blockIdx.x;
  const float* src = (t ? s3 : a) + (size_t)b * 256 * 128 + c;
  float sum[16], mx[16];
#pragma unroll
  for (int st = 0; st < 16; ++st) {
    float s = 0.f, m = -INFINITY;
    for (int q = 0; q < 16; ++q) {
      const float v = src[(size_t)(st * 16 + q) * 128];
      s += v;
      m = fmaxf(m, v);
    }
    sum[st] = s;
    mx[st] = m;
  }
  // level 4 (16 strips) down to level 0 (1 strip); strip means use the reference's mean over n positions
#pragma unroll
  for (int lev = 4; lev >= 0; --lev) {
    const int nb = 1 << lev;
    const float inv = 1.f / (float)(256 / nb);
#pragma unroll
    for (int st = 0; st < 16; ++st)
      if (st < nb) {
        const int row = kHppOff[lev] + t * nb + st;
        feat[((size_t)row * bsz + b) * 128 + c] = sum[st] * inv + mx[st];
      }
#pragma unroll
    for (int st = 0; st < 8; ++st)
      if (st < nb / 2) {
        sum[st] = sum[2 * st] + sum[2 * st + 1];
        mx[st] = fmaxf(mx[2 * st], mx[2 * st + 1]);
      }
  }
}

// per tensor: strip maxima, incoming strip gradients and tie counts of the 31 strips (5 levels).
__device__ __forceinline__ void hpp_bwd_one(const float* __restrict__ src, const float* __restrict__ dfeat, int t, int b,
                                            int bsz, int c, float* mxl /*31*/, float* gl /*31*/, float* cl /*31*/) {
  // level maxima: index base per level: lev 4 -> 0..15, lev 3 -> 16..23, lev 2 -> 24..27, lev 1 -> 28..29, lev 0 -> 30
#pragma unroll
  for (int st = 0; st < 16; ++st) {
    float m = -INFINITY;
#pragma unroll 1
    for (int q = 0; q < 16; ++q) m = fmaxf(m, src[(size_t)(st * 16 + q) * 128]);
    mxl[st] = m;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) mxl[16 + i] = fmaxf(mxl[2 * i], mxl[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 4; ++i) mxl[24 + i] = fmaxf(mxl[16 + 2 * i], mxl[16 + 2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 2; ++i) mxl[28 + i] = fmaxf(mxl[24 + 2 * i], mxl[24 + 2 * i + 1]);
  mxl[30] = fmaxf(mxl[28], mxl[29]);
  // incoming gradients per strip per level
#pragma unroll
  for (int lev = 0; lev < 5; ++lev) {
    const int nb = 1 << lev;
    const int base = lev == 4 ? 0 : lev == 3 ? 16 : lev == 2 ? 24 : lev == 1 ? 28 : 30;
#pragma unroll
    for (int st = 0; st < 16; ++st)
      if (st < nb) gl[base + st] = dfeat[((size_t)(kHppOff[lev] + t * nb + st) * bsz + b) * 128 + c];
  }
  // tie counts per strip per level
#pragma unroll
  for (int i = 0; i < 31; ++i) cl[i] = 0.f;
#pragma unroll
  for (int st = 0; st < 16; ++st) {
#pragma unroll 1
    for (int q = 0; q < 16; ++q) {
      const float v = src[(size_t)(st * 16 + q) * 128];
      cl[st] += v == mxl[st] ? 1.f : 0.f;
      cl[16 + st / 2] += v == mxl[16 + st / 2] ? 1.f : 0.f;
      cl[24 + st / 4] += v == mxl[24 + st / 4] ? 1.f : 0.f;
      cl[28 + st / 8] += v == mxl[28 + st / 8] ? 1.f : 0.f;
      cl[30] += v == mxl[30] ? 1.f : 0.f;
    }
  }
}

__device__ __forceinline__ float hpp_grad_at(float v, int st, const float* mxl, const float* gl, const float* cl) {
  float g = gl[st] * (1.f / 16.f) + gl[16 + st / 2] * (1.f / 32.f) + gl[24 + st / 4] * (1.f / 64.f) +
            gl[28 + st / 8] * (1.f / 128.f) + gl[30] * (1.f / 256.f);
  if (v == mxl[st]) g += gl[st] / cl[st];
  if (v == mxl[16 + st / 2]) g += gl[16 + st / 2] / cl[16 + st / 2];
  if (v == mxl[24 + st / 4]) g += gl[24 + st / 4] / cl[24 + st / 4];
  if (v == mxl[28 + st / 8]) g += gl[28 + st / 8] / cl[28 + st / 8];
  if (v == mxl[30]) g += gl[30] / cl[30];
  return g;
}

// 256 threads: thread (t, c) owns tensor t (0 = a, 1 = s3) and channel c of sample b.
// dm3 = dL/da + dL/ds3 (a also feeds s3 = b4 + a); dzb4 = dL/ds3 * LeakyReLU'(b4).
__global__ __launch_bounds__(256) void hpp_bwd_kernel(const float* __restrict__ a, const float* __restrict__ s3,
                                                      const float* __restrict__ b4, const float* __restrict__ dfeat,
                                                      float* __restrict__ dm3, float* __restrict__ dzb4, int bsz) {
  const int c = threadIdx.x & 127, t = threadIdx.x >> 7, b = blockIdx.x;
  const size_t base = (size_t)b * 256 * 128 + c;
  const float* src = t ? s3 : a;
  float mxl[31], gl[31], cl[31];
  hpp_bwd_one(src + base, dfeat, t, b, bsz, c, mxl, gl, cl);
  if (t == 1) {
#pragma unroll
    for (int st = 0; st < 16; ++st) {
#pragma unroll 1
      for (int q = 0; q < 16; ++q) {
        const size_t o = base + (size_t)(st * 16 + q) * 128;
        const float ds = hpp_grad_at(src[o], st, mxl, gl, cl);
        dm3[o] = ds;
        dzb4[o] = ds * ugn_lrelu_slope(b4[o]);
      }
    }
  }
  __syncthreads();  // the s3-half of the workgroup has written ds into dm3 (same addresses, same CU)
  if (t == 0) {
#pragma unroll
    for (int st = 0; st < 16; ++st) {
#pragma unroll 1
      for (int q = 0; q < 16; ++q) {
        const size_t o = base + (size_t)(st * 16 + q) * 128;
        dm3[o] += hpp_grad_at(src[o], st, mxl, gl, cl);
      }
    }
  }
}

}  // namespace

extern "C" int ugn_setmax_fwd(const float* p, const float* addend, float* m, float* sum_out, int b, int l, size_t s,
                              void* stream) {
  UGN_REQUIRE(p && m && b > 0 && l > 0 && s > 0 && s % 4 == 0, "ugn_setmax_fwd: bad arguments (s must be a multiple of 4)");
  UGN_REQUIRE(!addend || sum_out, "ugn_setmax_fwd: addend needs sum_out");
  const size_t s4 = s / 4;
  hipLaunchKernelGGL(setmax_fwd_kernel, dim3((unsigned)((s4 + 255) / 256), b), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)p, (const float4*)addend, (float4*)m, (float4*)sum_out, l, s4);
  UGN_CHECK_LAUNCH("setmax_fwd");
  return 0;
}

extern "C" int ugn_setmax_bwd(const float* p, const float* dm, float* out, int b, int l, size_t s, int apply_lrelu,
                              void* stream) {
  UGN_REQUIRE(p && dm && out && b > 0 && s > 0 && s % 4 == 0, "ugn_setmax_bwd: bad arguments");
  UGN_REQUIRE(l > 0 && l <= MAXL, "ugn_setmax_bwd: l must be in 1..%d (got %d)", MAXL, l);
  const size_t s4 = s / 4;
  hipLaunchKernelGGL(setmax_bwd_kernel, dim3((unsigned)((s4 + 127) / 128), b), dim3(128), 0, (hipStream_t)stream,
                     (const float4*)p, (const float4*)dm, (float4*)out, l, s4, apply_lrelu);
  UGN_CHECK_LAUNCH("setmax_bwd");
  return 0;
}

extern "C" int ugn_hpp_fwd(const float* a, const float* s3, float* feat, int b, void* stream) {
  UGN_REQUIRE(a && s3 && feat && b > 0, "ugn_hpp_fwd: bad arguments");
  hipLaunchKernelGGL(hpp_fwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, a, s3, feat, b);
  UGN_CHECK_LAUNCH("hpp_fwd");
  return 0;
}

extern "C" int ugn_hpp_bwd(const float* a, const float* s3, const float* b4, const float* dfeat, float* dm3, float* dzb4,
                           int b, void* stream) {
  UGN_REQUIRE(a && s3 && b4 && dfeat && dm3 && dzb4 && b > 0, "ugn_hpp_bwd: bad arguments");
  hipLaunchKernelGGL(hpp_bwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, a, s3, b4, dfeat, dm3, dzb4, b);
  UGN_CHECK_LAUNCH("hpp_bwd");
  return 0;
}
