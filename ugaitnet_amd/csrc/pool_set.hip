// Set pooling over the L frames (tf.math.reduce_max(axis=1), reference nets/mj_uwyhNets_ba.py:435,451,463) and
// horizontal pyramid pooling (:468-481), forward and backward.  All HBM-bound: one coalesced pass over the frames.
#include "common.h"

namespace {

constexpr int MAXL = 32;  // frames kept in registers by the backward (L = 25 in the reference)

// Up to kPoolJobs tensors of one shape per launch (blockIdx.z): the modality branches run the same pooling steps on their own
// tensors, and one launch for all of them costs what one of them costs when the tensors are small (5 clips per GPU).
constexpr int kPoolJobs = 4;
struct SetmaxJobs {
  const float4* p[kPoolJobs];
  const float4* addend[kPoolJobs];   // fwd: optional addend [b,s]; bwd: optional second gradient path [b,l,s]
  const float4* dm[kPoolJobs];       // bwd only
  float4* m[kPoolJobs];              // fwd: maxima; bwd: out
  float4* sum_out[kPoolJobs];
  int b[kPoolJobs];
};

__device__ __forceinline__ float4 max4(float4 a, float4 v) {
  return make_float4(fmaxf(a.x, v.x), fmaxf(a.y, v.y), fmaxf(a.z, v.z), fmaxf(a.w, v.w));
}

__global__ void setmax_fwd_kernel(const SetmaxJobs jt, int l, size_t s4) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.z, b = blockIdx.y;
  if (e >= s4 || b >= jt.b[j]) return;
  const float4* src = jt.p[j] + (size_t)b * l * s4 + e;
  float4 mx = src[0];
  int t = 1;
  for (; t + 6 <= l; t += 6) {    // six independent 16-byte loads in flight per lane (L = 25: 1 + 4 x 6)
    const float4 v0 = src[(size_t)t * s4], v1 = src[(size_t)(t + 1) * s4], v2 = src[(size_t)(t + 2) * s4];
    const float4 v3 = src[(size_t)(t + 3) * s4], v4 = src[(size_t)(t + 4) * s4], v5 = src[(size_t)(t + 5) * s4];
    mx = max4(max4(max4(mx, v0), max4(v1, v2)), max4(max4(v3, v4), v5));
  }
  for (; t < l; ++t) mx = max4(mx, src[(size_t)t * s4]);
  jt.m[j][(size_t)b * s4 + e] = mx;
  if (jt.addend[j]) {
    const float4 a = jt.addend[j][(size_t)b * s4 + e];
    jt.sum_out[j][(size_t)b * s4 + e] = make_float4(mx.x + a.x, mx.y + a.y, mx.z + a.z, mx.w + a.w);
  }
}

// Forward that also counts the maxima (the frames stay in registers, as in the backward): with the count the set-max
// gradient of a frame-level tensor can be formed inside the epilogue of the data gradient that consumes it
// ((p == m) ? dm / cnt : 0) instead of being materialised by setmax_bwd.
__global__ __launch_bounds__(128) void setmax_fwd_cnt_kernel(const float4* __restrict__ p, const float4* __restrict__ addend,
                                                             float4* __restrict__ m, float4* __restrict__ sum_out,
                                                             float4* __restrict__ cnt_out, int l, size_t s4) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= s4) return;
  const int b = blockIdx.y;
  const float4* src = p + (size_t)b * l * s4 + e;
  float4 v[32];
#pragma unroll
  for (int t = 0; t < 32; ++t)
    if (t < l) v[t] = src[(size_t)t * s4];
  float4 mx = v[0];
#pragma unroll
  for (int t = 1; t < 32; ++t)
    if (t < l) { mx.x = fmaxf(mx.x, v[t].x); mx.y = fmaxf(mx.y, v[t].y); mx.z = fmaxf(mx.z, v[t].z); mx.w = fmaxf(mx.w, v[t].w); }
  float4 cnt = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 32; ++t)
    if (t < l) {
      cnt.x += v[t].x == mx.x ? 1.f : 0.f; cnt.y += v[t].y == mx.y ? 1.f : 0.f;
      cnt.z += v[t].z == mx.z ? 1.f : 0.f; cnt.w += v[t].w == mx.w ? 1.f : 0.f;
    }
  m[(size_t)b * s4 + e] = mx;
  cnt_out[(size_t)b * s4 + e] = cnt;
  if (addend) {
    const float4 a = addend[(size_t)b * s4 + e];
    sum_out[(size_t)b * s4 + e] = make_float4(mx.x + a.x, mx.y + a.y, mx.z + a.z, mx.w + a.w);
  }
}

// out = g * LeakyReLU'(act) elementwise (act is a LeakyReLU OUTPUT: same sign as its input)
struct EltJobs {
  const float4* a[kPoolJobs];
  const float4* b[kPoolJobs];
  float4* out[kPoolJobs];
  size_t n4[kPoolJobs];
};
__global__ void lrelu_bwd_kernel(const EltJobs jt) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (e >= jt.n4[j]) return;
  const float4 x = jt.a[j][e], a = jt.b[j][e];
  jt.out[j][e] = make_float4(x.x * ugn_lrelu_slope(a.x), x.y * ugn_lrelu_slope(a.y), x.z * ugn_lrelu_slope(a.z), x.w * ugn_lrelu_slope(a.w));
}

__global__ void scale_kernel(float* __restrict__ x, float f, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= f;
}

__global__ void div_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, size_t n4) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n4) return;
  const float4 x = a[e], y = b[e];
  out[e] = make_float4(x.x / y.x, x.y / y.y, x.z / y.z, x.w / y.w);
}

__device__ __forceinline__ float sm_route(float v, float mx, float g, float add, int lrelu) {
  float o = (v == mx ? g : 0.f) + add;
  if (lrelu) o *= ugn_lrelu_slope(v);
  return o;
}

// TF reduce_max gradient: the incoming gradient is divided equally among all maxima.
// `addend` (optional, may alias `out`): a second gradient path into p (the data gradient of the next frame-level layer);
// out = (routed + addend) * LeakyReLU'(p): the Add of the two paths and the activation derivative cost one pass here instead
// of two extra operand streams in that data gradient's epilogue.
__global__ __launch_bounds__(128) void setmax_bwd_kernel(const SetmaxJobs jt, int l, size_t s4, int lrelu) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.z, b = blockIdx.y;
  if (e >= s4 || b >= jt.b[j]) return;
  const float4* dm = jt.dm[j];
  const float4* addend = jt.addend[j];
  const float4* src = jt.p[j] + (size_t)b * l * s4 + e;
  float4* dst = jt.m[j] + (size_t)b * l * s4 + e;
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
  const float4* asrc = addend ? addend + (size_t)b * l * s4 + e : nullptr;
#pragma unroll
  for (int t = 0; t < MAXL; ++t)
    if (t < l) {
      const float4 a = asrc ? asrc[(size_t)t * s4] : make_float4(0.f, 0.f, 0.f, 0.f);
      dst[(size_t)t * s4] = make_float4(sm_route(v[t].x, mx.x, gs.x, a.x, lrelu), sm_route(v[t].y, mx.y, gs.y, a.y, lrelu),
                                        sm_route(v[t].z, mx.z, gs.z, a.z, lrelu), sm_route(v[t].w, mx.w, gs.w, a.w, lrelu));
    }
}

// ---- set pooling with ROUTING WORDS (round 5; the fp32 twin of h2_elem.hip's routed kernels) -------------------------------
// reduce_max's gradient needs, per (clip, pixel, channel), WHICH of the l frames hold the maximum (TF splits the gradient evenly among
// them) and, for the LeakyReLU' that follows, which are positive: 2 x l bits, for which setmax_bwd_kernel reads all l frames again
// (4 l bytes per set element).  The forward pass writes them as two u32 words per element -- route[b][s4][2] as uint4: word k of
// channel c = bit t set iff frame t holds the maximum (k = 0) / is positive (k = 1) -- while it streams the frames anyway (one
// pass: a frame above the running maximum restarts the mask, an equal one joins it), and the gradient reads those 8 bytes INSTEAD of
// the frames.  Bit-identical to setmax_fwd_kernel / setmax_bwd_kernel (tests/test_kernels_gpu.py::test_setmax_routed).
struct RouteJobs {
  const float4* p[kPoolJobs];        // fwd: frames
  const float4* addend[kPoolJobs];   // fwd: optional [b,s]; bwd: optional second gradient path [b,l,s]
  const float4* dm[kPoolJobs];       // bwd
  float4* m[kPoolJobs];              // fwd: maxima; bwd: out
  float4* sum_out[kPoolJobs];
  uint4* route[kPoolJobs];
  int b[kPoolJobs];
};
__device__ __forceinline__ void route_step(float v, float& mx, unsigned& mask, unsigned& pos, unsigned bit) {
  mask = v > mx ? bit : (v == mx ? mask | bit : mask);
  mx = fmaxf(mx, v);
  pos |= v > 0.f ? bit : 0u;
}
__global__ __launch_bounds__(128) void setmax_fwd_routed_kernel(const RouteJobs jt, int l, size_t s4) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.z, b = blockIdx.y;
  if (e >= s4 || b >= jt.b[j]) return;
  const float4* src = jt.p[j] + (size_t)b * l * s4 + e;
  float4 mx = src[0];
  uint4 mask = make_uint4(1u, 1u, 1u, 1u);
  uint4 pos = make_uint4(mx.x > 0.f ? 1u : 0u, mx.y > 0.f ? 1u : 0u, mx.z > 0.f ? 1u : 0u, mx.w > 0.f ? 1u : 0u);
  int t = 1;
  for (; t + 6 <= l; t += 6) {    // six independent 16-byte loads in flight per lane (L = 25: 1 + 4 x 6)
    float4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = src[(size_t)(t + k) * s4];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const unsigned bit = 1u << (t + k);
      route_step(v[k].x, mx.x, mask.x, pos.x, bit);
      route_step(v[k].y, mx.y, mask.y, pos.y, bit);
      route_step(v[k].z, mx.z, mask.z, pos.z, bit);
      route_step(v[k].w, mx.w, mask.w, pos.w, bit);
    }
  }
  for (; t < l; ++t) {
    const float4 v = src[(size_t)t * s4];
    const unsigned bit = 1u << t;
    route_step(v.x, mx.x, mask.x, pos.x, bit);
    route_step(v.y, mx.y, mask.y, pos.y, bit);
    route_step(v.z, mx.z, mask.z, pos.z, bit);
    route_step(v.w, mx.w, mask.w, pos.w, bit);
  }
  jt.m[j][(size_t)b * s4 + e] = mx;
  uint4* r = jt.route[j] + ((size_t)b * s4 + e) * 2;
  r[0] = mask;
  r[1] = pos;
  if (jt.addend[j]) {
    const float4 a = jt.addend[j][(size_t)b * s4 + e];
    jt.sum_out[j][(size_t)b * s4 + e] = make_float4(mx.x + a.x, mx.y + a.y, mx.z + a.z, mx.w + a.w);
  }
}
__device__ __forceinline__ float route_out(unsigned mask, unsigned pos, unsigned bit, float g, float add, int lrelu) {
  float o = ((mask & bit) ? g : 0.f) + add;
  if (lrelu) o *= (pos & bit) ? 1.f : UGN_LRELU_ALPHA;
  return o;
}
__global__ __launch_bounds__(128) void setmax_bwd_routed_kernel(const RouteJobs jt, int l, size_t s4, int lrelu) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.z, b = blockIdx.y;
  if (e >= s4 || b >= jt.b[j]) return;
  const uint4* r = jt.route[j] + ((size_t)b * s4 + e) * 2;
  const uint4 mask = r[0], pos = r[1];
  const float4 g = jt.dm[j][(size_t)b * s4 + e];
  // (the division setmax_bwd_kernel does: gradient / number of maxima, counted as a float there -- same quotient)
  const float4 gs = make_float4(g.x / (float)__popc(mask.x), g.y / (float)__popc(mask.y), g.z / (float)__popc(mask.z), g.w / (float)__popc(mask.w));
  const float4* asrc = jt.addend[j] ? jt.addend[j] + (size_t)b * l * s4 + e : nullptr;
  float4* dst = jt.m[j] + (size_t)b * l * s4 + e;
  int t = 0;
  for (; t + 4 <= l; t += 4) {
    float4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = asrc ? asrc[(size_t)(t + k) * s4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned bit = 1u << (t + k);
      dst[(size_t)(t + k) * s4] = make_float4(route_out(mask.x, pos.x, bit, gs.x, a[k].x, lrelu), route_out(mask.y, pos.y, bit, gs.y, a[k].y, lrelu),
                                              route_out(mask.z, pos.z, bit, gs.z, a[k].z, lrelu), route_out(mask.w, pos.w, bit, gs.w, a[k].w, lrelu));
    }
  }
  for (; t < l; ++t) {
    const float4 a = asrc ? asrc[(size_t)t * s4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned bit = 1u << t;
    dst[(size_t)t * s4] = make_float4(route_out(mask.x, pos.x, bit, gs.x, a.x, lrelu), route_out(mask.y, pos.y, bit, gs.y, a.y, lrelu),
                                      route_out(mask.z, pos.z, bit, gs.z, a.z, lrelu), route_out(mask.w, pos.w, bit, gs.w, a.w, lrelu));
  }
}

// ---- HPP ------------------------------------------------------------------------------------------------
// feature rows: for bins in {1,2,4,8,16}: a's strips then s3's strips.  Row offsets of the a-part per level:
__device__ __constant__ int kHppOff[5] = {0, 2, 6, 14, 30};

// one thread per (b, tensor, channel): 256 positions, coalesced over the 128 channels.
struct HppJobs {
  const float* a[kPoolJobs];
  const float* s3[kPoolJobs];
  const float* b4[kPoolJobs];
  const float* dfeat[kPoolJobs];
  float* feat[kPoolJobs];      // fwd output
  float* dm3[kPoolJobs];
  float* dzb4[kPoolJobs];
  int b[kPoolJobs];
};
__global__ void hpp_fwd_kernel(const HppJobs jt) {
  const int c = threadIdx.x & 127, t = threadIdx.x >> 7;  // 256 threads: tensor 0 = a, 1 = s3
  const int b = blockIdx.x, j = blockIdx.y, bsz = jt.b[j];
  if (b >= bsz) return;
  const float* a = jt.a[j];
  const float* s3 = jt.s3[j];
  float* feat = jt.feat[j];
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

// HPP backward.  Workgroup = (sample b, 32-channel chunk), 512 threads = 16 finest strips x 32 channels; each thread
// owns the 16 positions of one finest strip for BOTH tensors (a and s3), so dm3 = dL/da + dL/ds3 needs no exchange.
// Level l (0..4) has 2^l strips of 256 / 2^l positions; strip maxima and tie counts of the coarser levels are combined
// through LDS.  mean: g / n ; max: g / (#maxima) to every maximum (TF reduce_max gradient).
// B4FMT: 0 fp32 b4, 1 H2 b4 (sign of its H half), 2 bf16 b4
template <int B4FMT>
__global__ __launch_bounds__(512) void hpp_bwd_kernel(const HppJobs jt) {
  __shared__ float sMx[2][16][32];
  __shared__ float sCnt[2][5][16][32];
  const int b = blockIdx.x >> 2, cc = blockIdx.x & 3, j = blockIdx.y, bsz = jt.b[j];
  if (b >= bsz) return;      // (whole workgroup: no barrier is skipped by a part of it)
  const float* a = jt.a[j];
  const float* s3 = jt.s3[j];
  const float* b4 = jt.b4[j];
  const float* dfeat = jt.dfeat[j];
  float* dm3 = jt.dm3[j];
  float* dzb4 = jt.dzb4[j];
  const int st = threadIdx.x >> 5, lc = threadIdx.x & 31, c = cc * 32 + lc;
  const size_t base = ((size_t)b * 256 + st * 16) * 128 + c;
  float v[2][16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    v[0][q] = a[base + (size_t)q * 128];
    v[1][q] = s3[base + (size_t)q * 128];
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float m = v[t][0];
#pragma unroll
    for (int q = 1; q < 16; ++q) m = fmaxf(m, v[t][q]);
    sMx[t][st][lc] = m;
  }
  __syncthreads();
  float mx[2][5];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int l = 0; l < 5; ++l) {
      const int span = 16 >> l, s0 = st & ~(span - 1);   // finest strips covered by my level-l strip
      float m = sMx[t][s0][lc];
      for (int k = 1; k < span; ++k) m = fmaxf(m, sMx[t][s0 + k][lc]);
      mx[t][l] = m;
      float cnt = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) cnt += v[t][q] == m ? 1.f : 0.f;
      sCnt[t][l][st][lc] = cnt;
    }
  }
  __syncthreads();
  float g[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int q = 0; q < 16; ++q) g[t][q] = 0.f;
#pragma unroll
    for (int l = 0; l < 5; ++l) {
      const int span = 16 >> l, s0 = st & ~(span - 1);
      float cnt = 0.f;
      for (int k = 0; k < span; ++k) cnt += sCnt[t][l][s0 + k][lc];
      const int nb = 1 << l;
      const float gl = dfeat[((size_t)(kHppOff[l] + t * nb + (st >> (4 - l))) * bsz + b) * 128 + c];
      const float gmean = gl * (1.f / (float)(256 >> l)), gmax = gl / cnt;
#pragma unroll
      for (int q = 0; q < 16; ++q) g[t][q] += gmean + (v[t][q] == mx[t][l] ? gmax : 0.f);
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const size_t o = base + (size_t)q * 128;
    dm3[o] = g[0][q] + g[1][q];
    if constexpr (B4FMT == 2) {
      const short hb = reinterpret_cast<const short*>(b4)[((size_t)b * 256 + st * 16 + q) * 128 + c];
      dzb4[o] = g[1][q] * (hb > 0 ? 1.f : UGN_LRELU_ALPHA);
    } else if constexpr (B4FMT == 1) {   // b4 is an H2 tensor [b][256][2][128] (ugaitnet_amd/csrc/mm_common.h): the sign of its H half is b4's sign
      const short hb = reinterpret_cast<const short*>(b4)[(((size_t)b * 256 + st * 16 + q) * 2) * 128 + c];
      dzb4[o] = g[1][q] * (hb > 0 ? 1.f : UGN_LRELU_ALPHA);
    } else {
      dzb4[o] = g[1][q] * ugn_lrelu_slope(b4[o]);
    }
  }
}

}  // namespace

extern "C" int ugn_setmax_fwd_multi(const float* const* p, const float* const* addend, float* const* m, float* const* sum_out,
                                    const int* b, int njobs, int l, size_t s, void* stream) {
  UGN_REQUIRE(p && m && b && njobs >= 1 && njobs <= kPoolJobs, "ugn_setmax_fwd_multi: bad arguments (1..%d jobs)", kPoolJobs);
  UGN_REQUIRE(l > 0 && s > 0 && s % 4 == 0, "ugn_setmax_fwd_multi: s must be a multiple of 4, l > 0");
  SetmaxJobs jt = {};
  int bmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(p[j] && m[j] && b[j] > 0, "ugn_setmax_fwd_multi: null pointer or b <= 0 in job %d", j);
    const float* ad = addend ? addend[j] : nullptr;
    UGN_REQUIRE(!ad || (sum_out && sum_out[j]), "ugn_setmax_fwd_multi: addend needs sum_out");
    jt.p[j] = (const float4*)p[j]; jt.addend[j] = (const float4*)ad; jt.m[j] = (float4*)m[j];
    jt.sum_out[j] = ad ? (float4*)sum_out[j] : nullptr; jt.b[j] = b[j];
    if (b[j] > bmax) bmax = b[j];
  }
  const size_t s4 = s / 4;
  hipLaunchKernelGGL(setmax_fwd_kernel, dim3((unsigned)((s4 + 127) / 128), bmax, njobs), dim3(128), 0, (hipStream_t)stream, jt, l, s4);
  UGN_CHECK_LAUNCH("setmax_fwd");
  return 0;
}

extern "C" int ugn_setmax_fwd(const float* p, const float* addend, float* m, float* sum_out, int b, int l, size_t s,
                              void* stream) {
  return ugn_setmax_fwd_multi(&p, &addend, &m, &sum_out, &b, 1, l, s, stream);
}

extern "C" int ugn_setmax_fwd_cnt(const float* p, const float* addend, float* m, float* sum_out, float* cnt, int b, int l,
                                  size_t s, void* stream) {
  UGN_REQUIRE(p && m && cnt && b > 0 && s > 0 && s % 4 == 0, "ugn_setmax_fwd_cnt: bad arguments (s must be a multiple of 4)");
  UGN_REQUIRE(l > 0 && l <= MAXL, "ugn_setmax_fwd_cnt: l must be in 1..%d (got %d)", MAXL, l);
  UGN_REQUIRE(!addend || sum_out, "ugn_setmax_fwd_cnt: addend needs sum_out");
  const size_t s4 = s / 4;
  hipLaunchKernelGGL(setmax_fwd_cnt_kernel, dim3((unsigned)((s4 + 127) / 128), b), dim3(128), 0, (hipStream_t)stream,
                     (const float4*)p, (const float4*)addend, (float4*)m, (float4*)sum_out, (float4*)cnt, l, s4);
  UGN_CHECK_LAUNCH("setmax_fwd_cnt");
  return 0;
}

extern "C" int ugn_lrelu_bwd_multi(const float* const* g, const float* const* act, float* const* out, const size_t* n, int njobs,
                                   void* stream) {
  UGN_REQUIRE(g && act && out && n && njobs >= 1 && njobs <= kPoolJobs, "ugn_lrelu_bwd_multi: bad arguments (1..%d jobs)", kPoolJobs);
  EltJobs jt = {};
  size_t nmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(g[j] && act[j] && out[j] && n[j] > 0 && n[j] % 4 == 0, "ugn_lrelu_bwd_multi: bad job %d (n must be a multiple of 4)", j);
    jt.a[j] = (const float4*)g[j]; jt.b[j] = (const float4*)act[j]; jt.out[j] = (float4*)out[j]; jt.n4[j] = n[j] / 4;
    if (n[j] / 4 > nmax) nmax = n[j] / 4;
  }
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)((nmax + 255) / 256), njobs), dim3(256), 0, (hipStream_t)stream, jt);
  UGN_CHECK_LAUNCH("lrelu_bwd");
  return 0;
}

extern "C" int ugn_lrelu_bwd(const float* g, const float* act, float* out, size_t n, void* stream) {
  return ugn_lrelu_bwd_multi(&g, &act, &out, &n, 1, stream);
}

extern "C" int ugn_scale(float* x, float factor, size_t n, void* stream) {
  UGN_REQUIRE(x && n > 0, "ugn_scale: bad arguments");
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, factor, n);
  UGN_CHECK_LAUNCH("scale");
  return 0;
}

extern "C" int ugn_div(const float* a, const float* b, float* out, size_t n, void* stream) {
  UGN_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "ugn_div: bad arguments (n must be a multiple of 4)");
  hipLaunchKernelGGL(div_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)a,
                     (const float4*)b, (float4*)out, n / 4);
  UGN_CHECK_LAUNCH("div");
  return 0;
}

extern "C" int ugn_setmax_bwd_multi(const float* const* p, const float* const* dm, const float* const* addend, float* const* out,
                                    const int* b, int njobs, int l, size_t s, int apply_lrelu, void* stream) {
  UGN_REQUIRE(p && dm && out && b && njobs >= 1 && njobs <= kPoolJobs, "ugn_setmax_bwd_multi: bad arguments (1..%d jobs)", kPoolJobs);
  UGN_REQUIRE(s > 0 && s % 4 == 0, "ugn_setmax_bwd_multi: s must be a multiple of 4");
  UGN_REQUIRE(l > 0 && l <= MAXL, "ugn_setmax_bwd: l must be in 1..%d (got %d)", MAXL, l);
  SetmaxJobs jt = {};
  int bmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(p[j] && dm[j] && out[j] && b[j] > 0, "ugn_setmax_bwd_multi: null pointer or b <= 0 in job %d", j);
    jt.p[j] = (const float4*)p[j]; jt.dm[j] = (const float4*)dm[j]; jt.addend[j] = (const float4*)(addend ? addend[j] : nullptr);
    jt.m[j] = (float4*)out[j]; jt.b[j] = b[j];
    if (b[j] > bmax) bmax = b[j];
  }
  const size_t s4 = s / 4;
  hipLaunchKernelGGL(setmax_bwd_kernel, dim3((unsigned)((s4 + 127) / 128), bmax, njobs), dim3(128), 0, (hipStream_t)stream, jt, l,
                     s4, apply_lrelu);
  UGN_CHECK_LAUNCH("setmax_bwd");
  return 0;
}

extern "C" int ugn_setmax_fwd_routed_multi(const float* const* p, const float* const* addend, float* const* m, float* const* sum_out,
                                           uint32_t* const* route, const int* b, int njobs, int l, size_t s, void* stream) {
  UGN_REQUIRE(p && m && route && b && njobs >= 1 && njobs <= kPoolJobs, "ugn_setmax_fwd_routed_multi: bad arguments (1..%d jobs)", kPoolJobs);
  UGN_REQUIRE(l > 0 && l <= 32 && s > 0 && s % 4 == 0, "ugn_setmax_fwd_routed_multi: s must be a multiple of 4, l in 1..32 (one bit per frame)");
  RouteJobs jt = {};
  int bmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(p[j] && m[j] && route[j] && b[j] > 0, "ugn_setmax_fwd_routed_multi: null pointer or b <= 0 in job %d", j);
    const float* ad = addend ? addend[j] : nullptr;
    UGN_REQUIRE(!ad || (sum_out && sum_out[j]), "ugn_setmax_fwd_routed_multi: addend needs sum_out");
    jt.p[j] = (const float4*)p[j]; jt.addend[j] = (const float4*)ad; jt.m[j] = (float4*)m[j];
    jt.sum_out[j] = ad ? (float4*)sum_out[j] : nullptr; jt.route[j] = (uint4*)route[j]; jt.b[j] = b[j];
    if (b[j] > bmax) bmax = b[j];
  }
  const size_t s4 = s / 4;
  hipLaunchKernelGGL(setmax_fwd_routed_kernel, dim3((unsigned)((s4 + 127) / 128), bmax, njobs), dim3(128), 0, (hipStream_t)stream, jt, l, s4);
  UGN_CHECK_LAUNCH("setmax_fwd_routed");
  return 0;
}

extern "C" int ugn_setmax_bwd_routed_multi(const uint32_t* const* route, const float* const* dm, const float* const* addend,
                                           float* const* out, const int* b, int njobs, int l, size_t s, int apply_lrelu, void* stream) {
  UGN_REQUIRE(route && dm && out && b && njobs >= 1 && njobs <= kPoolJobs, "ugn_setmax_bwd_routed_multi: bad arguments (1..%d jobs)", kPoolJobs);
  UGN_REQUIRE(l > 0 && l <= 32 && s > 0 && s % 4 == 0, "ugn_setmax_bwd_routed_multi: s must be a multiple of 4, l in 1..32");
  RouteJobs jt = {};
  int bmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(route[j] && dm[j] && out[j] && b[j] > 0, "ugn_setmax_bwd_routed_multi: null pointer or b <= 0 in job %d", j);
    jt.route[j] = (uint4*)route[j]; jt.dm[j] = (const float4*)dm[j]; jt.addend[j] = (const float4*)(addend ? addend[j] : nullptr);
    jt.m[j] = (float4*)out[j]; jt.b[j] = b[j];
    if (b[j] > bmax) bmax = b[j];
  }
  const size_t s4 = s / 4;
  hipLaunchKernelGGL(setmax_bwd_routed_kernel, dim3((unsigned)((s4 + 127) / 128), bmax, njobs), dim3(128), 0, (hipStream_t)stream, jt, l,
                     s4, apply_lrelu);
  UGN_CHECK_LAUNCH("setmax_bwd_routed");
  return 0;
}

extern "C" int ugn_setmax_bwd(const float* p, const float* dm, const float* addend, float* out, int b, int l, size_t s,
                              int apply_lrelu, void* stream) {
  return ugn_setmax_bwd_multi(&p, &dm, &addend, &out, &b, 1, l, s, apply_lrelu, stream);
}

extern "C" int ugn_hpp_fwd_multi(const float* const* a, const float* const* s3, float* const* feat, const int* b, int njobs,
                                 void* stream) {
  UGN_REQUIRE(a && s3 && feat && b && njobs >= 1 && njobs <= kPoolJobs, "ugn_hpp_fwd_multi: bad arguments (1..%d jobs)", kPoolJobs);
  HppJobs jt = {};
  int bmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(a[j] && s3[j] && feat[j] && b[j] > 0, "ugn_hpp_fwd_multi: bad job %d", j);
    jt.a[j] = a[j]; jt.s3[j] = s3[j]; jt.feat[j] = feat[j]; jt.b[j] = b[j];
    if (b[j] > bmax) bmax = b[j];
  }
  hipLaunchKernelGGL(hpp_fwd_kernel, dim3(bmax, njobs), dim3(256), 0, (hipStream_t)stream, jt);
  UGN_CHECK_LAUNCH("hpp_fwd");
  return 0;
}

extern "C" int ugn_hpp_fwd(const float* a, const float* s3, float* feat, int b, void* stream) {
  return ugn_hpp_fwd_multi(&a, &s3, &feat, &b, 1, stream);
}

extern "C" int ugn_hpp_bwd_multi(const float* const* a, const float* const* s3, const float* const* b4, const float* const* dfeat,
                                 float* const* dm3, float* const* dzb4, const int* b, int njobs, void* stream) {
  UGN_REQUIRE(a && s3 && b4 && dfeat && dm3 && dzb4 && b && njobs >= 1 && njobs <= kPoolJobs,
              "ugn_hpp_bwd_multi: bad arguments (1..%d jobs)", kPoolJobs);
  HppJobs jt = {};
  int bmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(a[j] && s3[j] && b4[j] && dfeat[j] && dm3[j] && dzb4[j] && b[j] > 0, "ugn_hpp_bwd_multi: bad job %d", j);
    jt.a[j] = a[j]; jt.s3[j] = s3[j]; jt.b4[j] = b4[j]; jt.dfeat[j] = dfeat[j]; jt.dm3[j] = dm3[j]; jt.dzb4[j] = dzb4[j];
    jt.b[j] = b[j];
    if (b[j] > bmax) bmax = b[j];
  }
  hipLaunchKernelGGL(hpp_bwd_kernel<0>, dim3(bmax * 4, njobs), dim3(512), 0, (hipStream_t)stream, jt);
  UGN_CHECK_LAUNCH("hpp_bwd");
  return 0;
}

/* the same with b4 held as an H2 tensor [b][16][16][2][128] (only its sign is used); dm3 / dzb4 stay fp32 */
static int hpp_bwd_b4fmt(const float* const* a, const float* const* s3, const uint16_t* const* b4, const float* const* dfeat,
                         float* const* dm3, float* const* dzb4, const int* b, int njobs, void* stream, bool bf);
#ifndef UGN_WITH_H2
#define UGN_WITH_H2 0
#endif
#if UGN_WITH_H2      /* entry point of the opt-in f16x2 set (build.py --h2) */
#include "../../include/ugaitnet_hip_h2.h"
extern "C" int ugn_hpp_bwd_b4h2_multi(const float* const* a, const float* const* s3, const uint16_t* const* b4,
                                      const float* const* dfeat, float* const* dm3, float* const* dzb4, const int* b, int njobs,
                                      void* stream) {
  return hpp_bwd_b4fmt(a, s3, b4, dfeat, dm3, dzb4, b, njobs, stream, false);
}
#endif
/* the same with b4 as a bf16 tensor [b][16][16][128] */
extern "C" int ugn_hpp_bwd_b4bf_multi(const float* const* a, const float* const* s3, const uint16_t* const* b4,
                                      const float* const* dfeat, float* const* dm3, float* const* dzb4, const int* b, int njobs,
                                      void* stream) {
  return hpp_bwd_b4fmt(a, s3, b4, dfeat, dm3, dzb4, b, njobs, stream, true);
}
static int hpp_bwd_b4fmt(const float* const* a, const float* const* s3, const uint16_t* const* b4, const float* const* dfeat,
                         float* const* dm3, float* const* dzb4, const int* b, int njobs, void* stream, bool bf) {
  UGN_REQUIRE(a && s3 && b4 && dfeat && dm3 && dzb4 && b && njobs >= 1 && njobs <= kPoolJobs,
              "ugn_hpp_bwd_b4h2_multi: bad arguments (1..%d jobs)", kPoolJobs);
  HppJobs jt = {};
  int bmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(a[j] && s3[j] && b4[j] && dfeat[j] && dm3[j] && dzb4[j] && b[j] > 0, "ugn_hpp_bwd_b4h2_multi: bad job %d", j);
    jt.a[j] = a[j]; jt.s3[j] = s3[j]; jt.b4[j] = reinterpret_cast<const float*>(b4[j]); jt.dfeat[j] = dfeat[j];
    jt.dm3[j] = dm3[j]; jt.dzb4[j] = dzb4[j]; jt.b[j] = b[j];
    if (b[j] > bmax) bmax = b[j];
  }
  if (bf) hipLaunchKernelGGL(hpp_bwd_kernel<2>, dim3(bmax * 4, njobs), dim3(512), 0, (hipStream_t)stream, jt);
  else hipLaunchKernelGGL(hpp_bwd_kernel<1>, dim3(bmax * 4, njobs), dim3(512), 0, (hipStream_t)stream, jt);
  UGN_CHECK_LAUNCH("hpp_bwd_b4h2");
  return 0;
}

extern "C" int ugn_hpp_bwd(const float* a, const float* s3, const float* b4, const float* dfeat, float* dm3, float* dzb4,
                           int b, void* stream) {
  return ugn_hpp_bwd_multi(&a, &s3, &b4, &dfeat, &dm3, &dzb4, &b, 1, stream);
}
