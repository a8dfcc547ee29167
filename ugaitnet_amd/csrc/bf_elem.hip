// The HBM-bound steps between the 3x3 layers on bf16 tensors (configs[4]: bf16 activations / gradients in HBM): set pooling over
// the L frames forward / backward (tf.math.reduce_max(axis=1) + Add, reference nets/mj_uwyhNets_ba.py:435,451-452,463-465),
// LeakyReLU', fp32 -> bf16.  The bf16 counterpart of h2_elem.hip: same arithmetic in fp32 on the loaded values, results rounded
// to bf16 (round to nearest even) on store; no block exponents.
#include "mm_common.h"

using namespace ugn_mm;

namespace {

constexpr int kJobs = 6;
constexpr int MAXL = 32;

// 4 channels of one pixel: one 8-byte load -> 4 values (a bf16 is the upper half of an fp32)
struct V4 { float x, y, z, w; };
__device__ __forceinline__ V4 ld4(const uint16_t* __restrict__ rec, int c, int ch) {   // rec = the pixel's record [c]
  const uint2 v = *reinterpret_cast<const uint2*>(rec + ch);
  return {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
}
__device__ __forceinline__ unsigned bfp(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
__device__ __forceinline__ void st4(uint16_t* __restrict__ rec, int c, int ch, V4 v) {
  *reinterpret_cast<uint2*>(rec + ch) = make_uint2(bfp(v.x, v.y), bfp(v.z, v.w));
}
__device__ __forceinline__ V4 max4(V4 a, V4 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }

struct SetJobs {
  const uint16_t* p[kJobs];      // frames  bf16 [b*l][pix][c]
  const H2Meta* p_meta[kJobs];
  const uint16_t* add[kJobs];    // fwd: set-level addend H2 [b][pix][2][c]; bwd: second gradient path H2 [b*l][pix][2][c]
  const H2Meta* add_meta[kJobs];
  const void* dm[kJobs];         // bwd: gradient of the maxima: H2 [b][pix][2][c], or fp32 [b][pix][c] (DM_F32)
  const H2Meta* dm_meta[kJobs];
  uint16_t* m[kJobs];            // fwd: maxima (H2, optional); bwd: out (H2, may alias add)
  H2Meta* m_meta[kJobs];
  uint16_t* sum[kJobs];          // fwd: maxima + addend (H2)
  H2Meta* sum_meta[kJobs];
  float* m_f32[kJobs];           // fwd, F32OUT: maxima / sums as fp32 [b][pix][c] (the inputs of HPP)
  float* sum_f32[kJobs];
  uint32_t* route[kJobs];        // routing words u32 [b][pix][2][c] (h2_elem.hip): written by the forward pass when set, read by the
  int b[kJobs];                  // gradient INSTEAD of the frames
};

struct Route4 { uint4 mx, sg; };

// grid: (pixel-channel quads / 128, b, jobs).  npix = pixels per image, c = channels.
template <bool F32OUT>
__global__ __launch_bounds__(128) void setmax_fwd_bf_kernel(const SetJobs jt, int l, int npix, int c) {
  const int j = blockIdx.z, b = blockIdx.y;
  const int e = blockIdx.x * 128 + threadIdx.x, q = c / 4;
  if (b >= jt.b[j]) return;
  const bool has_add = jt.add[j] != nullptr;
  if (e >= npix * q) return;
  const int pix = e / q, ch = (e - pix * q) * 4;
  const size_t rec = (size_t)c;                                       // elements per pixel record
  const uint16_t* src = jt.p[j] + ((size_t)b * l * npix + pix) * rec;
  const size_t fstride = (size_t)npix * rec;
  V4 mx = ld4(src, c, ch);
  if (jt.route[j]) {                 // one pass: a frame above the running maximum restarts its bit mask, an equal one joins it
    Route4 r = {make_uint4(1u, 1u, 1u, 1u), make_uint4(mx.x > 0.f ? 1u : 0u, mx.y > 0.f ? 1u : 0u, mx.z > 0.f ? 1u : 0u, mx.w > 0.f ? 1u : 0u)};
    auto take = [&](float v, float& m, unsigned& mb, unsigned& sb, int t) {
      const unsigned bit = 1u << t;
      mb = v > m ? bit : (v == m ? mb | bit : mb);
      m = fmaxf(m, v);
      sb |= v > 0.f ? bit : 0u;
    };
    auto take4 = [&](V4 v, int t) {
      take(v.x, mx.x, r.mx.x, r.sg.x, t); take(v.y, mx.y, r.mx.y, r.sg.y, t);
      take(v.z, mx.z, r.mx.z, r.sg.z, t); take(v.w, mx.w, r.mx.w, r.sg.w, t);
    };
    int t = 1;
    for (; t + 4 <= l; t += 4) {
      const V4 v0 = ld4(src + (size_t)t * fstride, c, ch), v1 = ld4(src + (size_t)(t + 1) * fstride, c, ch);
      const V4 v2 = ld4(src + (size_t)(t + 2) * fstride, c, ch), v3 = ld4(src + (size_t)(t + 3) * fstride, c, ch);
      take4(v0, t); take4(v1, t + 1); take4(v2, t + 2); take4(v3, t + 3);
    }
    for (; t < l; ++t) take4(ld4(src + (size_t)t * fstride, c, ch), t);
    uint32_t* rp = jt.route[j] + ((size_t)b * npix + pix) * 2 * c + ch;
    *reinterpret_cast<uint4*>(rp) = r.mx;
    *reinterpret_cast<uint4*>(rp + c) = r.sg;
  } else {
    int t = 1;
    for (; t + 4 <= l; t += 4) {       // four frames (8 loads of 8 bytes) in flight per lane
      const V4 v0 = ld4(src + (size_t)t * fstride, c, ch), v1 = ld4(src + (size_t)(t + 1) * fstride, c, ch);
      const V4 v2 = ld4(src + (size_t)(t + 2) * fstride, c, ch), v3 = ld4(src + (size_t)(t + 3) * fstride, c, ch);
      mx = max4(max4(mx, v0), max4(max4(v1, v2), v3));
    }
    for (; t < l; ++t) mx = max4(mx, ld4(src + (size_t)t * fstride, c, ch));
  }
  const size_t o = (size_t)b * npix + pix;
  V4 sm = mx;
  if (has_add) {
    const V4 a = ld4(jt.add[j] + o * rec, c, ch);
    sm = {mx.x + a.x, mx.y + a.y, mx.z + a.z, mx.w + a.w};
  }
  if constexpr (F32OUT) {
    if (jt.m_f32[j]) *reinterpret_cast<float4*>(jt.m_f32[j] + o * c + ch) = make_float4(mx.x, mx.y, mx.z, mx.w);
    if (has_add) *reinterpret_cast<float4*>(jt.sum_f32[j] + o * c + ch) = make_float4(sm.x, sm.y, sm.z, sm.w);
  } else {
    if (jt.m[j]) st4(jt.m[j] + o * rec, c, ch, mx);
    if (has_add) st4(jt.sum[j] + o * rec, c, ch, sm);
  }
}

// TF reduce_max gradient (equal split among the maxima) + the second gradient path + LeakyReLU'(p), as setmax_bwd_kernel of
// pool_set.hip:   out = ((p == max ? dm / #maxima : 0) + addend) * LeakyReLU'(p)
template <bool DM_F32, bool ROUTED>
__global__ __launch_bounds__(128) void setmax_bwd_bf_kernel(const SetJobs jt, int l, int npix, int c, int lrelu) {
  const int j = blockIdx.z, b = blockIdx.y;
  const int e = blockIdx.x * 128 + threadIdx.x, q = c / 4;
  if (b >= jt.b[j]) return;
  const bool has_add = jt.add[j] != nullptr;
  if (e >= npix * q) return;
  const int pix = e / q, ch = (e - pix * q) * 4;
  const size_t rec = (size_t)c, fstride = (size_t)npix * rec;
  const size_t o = (size_t)b * npix + pix;
  // which frames hold the maximum / are positive: the forward pass's routing words, or the frames themselves
  Route4 r = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  if constexpr (ROUTED) {
    const uint32_t* rp = jt.route[j] + o * 2 * c + ch;
    r.mx = *reinterpret_cast<const uint4*>(rp);
    r.sg = *reinterpret_cast<const uint4*>(rp + c);
  } else {
    const uint16_t* src = jt.p[j] + ((size_t)b * l * npix + pix) * rec;
    V4 v[MAXL];
#pragma unroll
    for (int t = 0; t < MAXL; ++t)
      if (t < l) v[t] = ld4(src + (size_t)t * fstride, c, ch);
    V4 mx = v[0];
#pragma unroll
    for (int t = 1; t < MAXL; ++t)
      if (t < l) mx = max4(mx, v[t]);
#pragma unroll
    for (int t = 0; t < MAXL; ++t)
      if (t < l) {
        r.mx.x |= (v[t].x == mx.x ? 1u : 0u) << t; r.mx.y |= (v[t].y == mx.y ? 1u : 0u) << t;
        r.mx.z |= (v[t].z == mx.z ? 1u : 0u) << t; r.mx.w |= (v[t].w == mx.w ? 1u : 0u) << t;
        r.sg.x |= (v[t].x > 0.f ? 1u : 0u) << t; r.sg.y |= (v[t].y > 0.f ? 1u : 0u) << t;
        r.sg.z |= (v[t].z > 0.f ? 1u : 0u) << t; r.sg.w |= (v[t].w > 0.f ? 1u : 0u) << t;
      }
  }
  const V4 cnt = {(float)__popc(r.mx.x), (float)__popc(r.mx.y), (float)__popc(r.mx.z), (float)__popc(r.mx.w)};
  V4 g;
  if constexpr (DM_F32) {
    const float4 t4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(jt.dm[j]) + o * c + ch);
    g = {t4.x, t4.y, t4.z, t4.w};
  } else {
    g = ld4(reinterpret_cast<const uint16_t*>(jt.dm[j]) + o * rec, c, ch);
  }
  const V4 gs = {g.x / cnt.x, g.y / cnt.y, g.z / cnt.z, g.w / cnt.w};
  const uint16_t* asrc = has_add ? jt.add[j] + ((size_t)b * l * npix + pix) * rec : nullptr;
  uint16_t* dst = jt.m[j] + ((size_t)b * l * npix + pix) * rec;
  auto route = [&](unsigned mbits, unsigned sbits, int t, float gg, float aa) {
    float rr = ((mbits >> t) & 1u ? gg : 0.f) + aa;
    if (lrelu) rr *= (sbits >> t) & 1u ? 1.f : UGN_LRELU_ALPHA;
    return rr;
  };
  constexpr int PF = 4;          // addends of four frames in flight
#pragma unroll
  for (int t0 = 0; t0 < MAXL; t0 += PF) {
    V4 a[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      a[k] = {0.f, 0.f, 0.f, 0.f};
      if (has_add && t0 + k < l) a[k] = ld4(asrc + (size_t)(t0 + k) * fstride, c, ch);
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int t = t0 + k;
      if (t < l)
        st4(dst + (size_t)t * fstride, c, ch, {route(r.mx.x, r.sg.x, t, gs.x, a[k].x), route(r.mx.y, r.sg.y, t, gs.y, a[k].y),
                                               route(r.mx.z, r.sg.z, t, gs.z, a[k].z), route(r.mx.w, r.sg.w, t, gs.w, a[k].w)});
    }
  }
}

struct EltJobs {
  const uint16_t* g[kJobs];
  const H2Meta* g_meta[kJobs];
  const uint16_t* act[kJobs];
  uint16_t* out[kJobs];
  H2Meta* out_meta[kJobs];
  size_t npix[kJobs];
};
// out = g * LeakyReLU'(act) (act: a LeakyReLU output, same sign as its input); same exponent, amax carried over as a bound
__global__ void lrelu_bwd_bf_kernel(const EltJobs jt, int c) {
  const int j = blockIdx.y, q = c / 4;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= jt.npix[j] * q) return;
  const size_t pix = e / q;
  const int ch = (int)(e - pix * q) * 4;
  const size_t rec = (size_t)c;
  const V4 g = ld4(jt.g[j] + pix * rec, c, ch), a = ld4(jt.act[j] + pix * rec, c, ch);
  st4(jt.out[j] + pix * rec, c, ch, {g.x * ugn_lrelu_slope(a.x), g.y * ugn_lrelu_slope(a.y), g.z * ugn_lrelu_slope(a.z),
                                     g.w * ugn_lrelu_slope(a.w)});
}

struct CvtJobs {
  const float* x[kJobs];
  uint16_t* y[kJobs];
  size_t n[kJobs];             // elements (a multiple of 4)
};
__global__ void cvt_multi_kernel(const CvtJobs jt) {
  const int j = blockIdx.y;
  const size_t nq = jt.n[j] / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(jt.x[j])[i];
    reinterpret_cast<uint2*>(jt.y[j])[i] = make_uint2(bfp(v.x, v.y), bfp(v.z, v.w));
  }
}

}  // namespace

static int fill(SetJobs& jt, const uint16_t* const* p, const uint16_t* const* add, const int* b, int njobs, int* bmax, const char* who) {
  UGN_REQUIRE(p && b && njobs >= 1 && njobs <= kJobs, "%s: bad arguments (1..%d jobs)", who, kJobs);
  *bmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(p[j] && b[j] > 0, "%s: null pointer or b <= 0 in job %d", who, j);
    jt.p[j] = p[j]; jt.add[j] = add ? add[j] : nullptr; jt.b[j] = b[j];
    if (b[j] > *bmax) *bmax = b[j];
  }
  return 0;
}

static int set_routes(SetJobs& jt, uint32_t* const* route, int njobs, int l, const char* who) {
  for (int j = 0; j < njobs; ++j) jt.route[j] = route ? route[j] : nullptr;
  UGN_REQUIRE(!route || l <= MAXL, "%s: routing words hold at most %d frames", who, MAXL);
  return 0;
}

/* bf16 set pooling: m (optional) = max over the l frames, sum = m + addend (bf16 [b][npix][c]); route (optional): routing words
 * u32 [b][npix][2][c] as ugn_h2_setmax_fwd_routed_multi writes them */
extern "C" int ugn_bf_setmax_fwd_routed_multi(const uint16_t* const* p, const uint16_t* const* addend, uint16_t* const* m,
                                              uint16_t* const* sum, uint32_t* const* route, const int* b, int njobs, int l, int npix,
                                              int c, void* stream);
extern "C" int ugn_bf_setmax_fwd_multi(const uint16_t* const* p, const uint16_t* const* addend, uint16_t* const* m,
                                       uint16_t* const* sum, const int* b, int njobs, int l, int npix, int c, void* stream) {
  return ugn_bf_setmax_fwd_routed_multi(p, addend, m, sum, nullptr, b, njobs, l, npix, c, stream);
}
extern "C" int ugn_bf_setmax_fwd_routed_multi(const uint16_t* const* p, const uint16_t* const* addend, uint16_t* const* m,
                                              uint16_t* const* sum, uint32_t* const* route, const int* b, int njobs, int l, int npix,
                                              int c, void* stream) {
  SetJobs jt = {};
  int bmax;
  if (int rc = fill(jt, p, addend, b, njobs, &bmax, "ugn_bf_setmax_fwd_multi")) return rc;
  if (int rc = set_routes(jt, route, njobs, l, "ugn_bf_setmax_fwd_routed_multi")) return rc;
  UGN_REQUIRE(l > 0 && npix > 0 && c > 0 && c % 4 == 0, "ugn_bf_setmax_fwd_multi: c must be a multiple of 4");
  for (int j = 0; j < njobs; ++j) {
    jt.m[j] = m ? m[j] : nullptr; jt.sum[j] = sum ? sum[j] : nullptr;
    UGN_REQUIRE(!jt.add[j] || jt.sum[j], "ugn_bf_setmax_fwd_multi: addend needs sum (job %d)", j);
    UGN_REQUIRE(jt.m[j] || jt.add[j], "ugn_bf_setmax_fwd_multi: nothing to write (job %d)", j);
  }
  const unsigned gx = (unsigned)(((size_t)npix * (c / 4) + 127) / 128);
  hipLaunchKernelGGL(setmax_fwd_bf_kernel<false>, dim3(gx, bmax, njobs), dim3(128), 0, (hipStream_t)stream, jt, l, npix, c);
  UGN_CHECK_LAUNCH("bf_setmax_fwd");
  return 0;
}

/* the same with fp32 outputs [b][npix][c] (the inputs of HPP) */
extern "C" int ugn_bf_setmax_fwd_f32_routed_multi(const uint16_t* const* p, const uint16_t* const* addend, float* const* m,
                                                  float* const* sum, uint32_t* const* route, const int* b, int njobs, int l,
                                                  int npix, int c, void* stream);
extern "C" int ugn_bf_setmax_fwd_f32_multi(const uint16_t* const* p, const uint16_t* const* addend, float* const* m, float* const* sum,
                                           const int* b, int njobs, int l, int npix, int c, void* stream) {
  return ugn_bf_setmax_fwd_f32_routed_multi(p, addend, m, sum, nullptr, b, njobs, l, npix, c, stream);
}
extern "C" int ugn_bf_setmax_fwd_f32_routed_multi(const uint16_t* const* p, const uint16_t* const* addend, float* const* m,
                                                  float* const* sum, uint32_t* const* route, const int* b, int njobs, int l,
                                                  int npix, int c, void* stream) {
  SetJobs jt = {};
  int bmax;
  if (int rc = fill(jt, p, addend, b, njobs, &bmax, "ugn_bf_setmax_fwd_f32_multi")) return rc;
  if (int rc = set_routes(jt, route, njobs, l, "ugn_bf_setmax_fwd_f32_routed_multi")) return rc;
  UGN_REQUIRE(l > 0 && npix > 0 && c > 0 && c % 4 == 0, "ugn_bf_setmax_fwd_f32_multi: c must be a multiple of 4");
  for (int j = 0; j < njobs; ++j) {
    jt.m_f32[j] = m ? m[j] : nullptr; jt.sum_f32[j] = sum ? sum[j] : nullptr;
    UGN_REQUIRE(!jt.add[j] || jt.sum_f32[j], "ugn_bf_setmax_fwd_f32_multi: addend needs sum (job %d)", j);
    UGN_REQUIRE(jt.m_f32[j] || jt.add[j], "ugn_bf_setmax_fwd_f32_multi: nothing to write (job %d)", j);
  }
  const unsigned gx = (unsigned)(((size_t)npix * (c / 4) + 127) / 128);
  hipLaunchKernelGGL(setmax_fwd_bf_kernel<true>, dim3(gx, bmax, njobs), dim3(128), 0, (hipStream_t)stream, jt, l, npix, c);
  UGN_CHECK_LAUNCH("bf_setmax_fwd_f32");
  return 0;
}

/* out = ((p == max over l ? dm / #maxima : 0) + addend) * (apply_lrelu ? LeakyReLU'(p) : 1); dm bf16 [b][npix][c] or (dm_is_f32) fp32.
 * With `route` (the forward pass's routing words) the frames p are not read (p may be null): same results. */
static int bf_setmax_bwd_any(const uint16_t* const* p, const uint32_t* const* route, const void* const* dm, int dm_is_f32,
                             const uint16_t* const* addend, uint16_t* const* out, const int* b, int njobs, int l, int npix, int c,
                             int apply_lrelu, void* stream, const char* who) {
  SetJobs jt = {};
  int bmax = 0;
  if (route) {
    UGN_REQUIRE(b && njobs >= 1 && njobs <= kJobs, "%s: bad arguments (1..%d jobs)", who, kJobs);
    for (int j = 0; j < njobs; ++j) {
      UGN_REQUIRE(route[j] && b[j] > 0, "%s: null routing words or b <= 0 in job %d", who, j);
      jt.route[j] = const_cast<uint32_t*>(route[j]);
      jt.add[j] = addend ? addend[j] : nullptr;
      jt.b[j] = b[j];
      if (b[j] > bmax) bmax = b[j];
    }
  } else if (int rc = fill(jt, p, addend, b, njobs, &bmax, who)) {
    return rc;
  }
  UGN_REQUIRE(dm && out, "%s: null array", who);
  UGN_REQUIRE(l > 0 && l <= MAXL && npix > 0 && c > 0 && c % 4 == 0, "%s: l must be 1..%d, c a multiple of 4", who, MAXL);
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(dm[j] && out[j], "%s: null pointer in job %d", who, j);
    jt.dm[j] = dm[j]; jt.m[j] = out[j];
  }
  const dim3 grid((unsigned)(((size_t)npix * (c / 4) + 127) / 128), bmax, njobs);
  hipStream_t st = (hipStream_t)stream;
  if (route) {
    if (dm_is_f32) hipLaunchKernelGGL((setmax_bwd_bf_kernel<true, true>), grid, dim3(128), 0, st, jt, l, npix, c, apply_lrelu);
    else hipLaunchKernelGGL((setmax_bwd_bf_kernel<false, true>), grid, dim3(128), 0, st, jt, l, npix, c, apply_lrelu);
  } else {
    if (dm_is_f32) hipLaunchKernelGGL((setmax_bwd_bf_kernel<true, false>), grid, dim3(128), 0, st, jt, l, npix, c, apply_lrelu);
    else hipLaunchKernelGGL((setmax_bwd_bf_kernel<false, false>), grid, dim3(128), 0, st, jt, l, npix, c, apply_lrelu);
  }
  UGN_CHECK_LAUNCH("bf_setmax_bwd");
  return 0;
}
extern "C" int ugn_bf_setmax_bwd_multi(const uint16_t* const* p, const void* const* dm, int dm_is_f32, const uint16_t* const* addend,
                                       uint16_t* const* out, const int* b, int njobs, int l, int npix, int c, int apply_lrelu,
                                       void* stream) {
  return bf_setmax_bwd_any(p, nullptr, dm, dm_is_f32, addend, out, b, njobs, l, npix, c, apply_lrelu, stream, "ugn_bf_setmax_bwd_multi");
}
extern "C" int ugn_bf_setmax_bwd_routed_multi(const uint32_t* const* route, const void* const* dm, int dm_is_f32,
                                              const uint16_t* const* addend, uint16_t* const* out, const int* b, int njobs, int l,
                                              int npix, int c, int apply_lrelu, void* stream) {
  UGN_REQUIRE(route, "ugn_bf_setmax_bwd_routed_multi: null routing words");
  return bf_setmax_bwd_any(nullptr, route, dm, dm_is_f32, addend, out, b, njobs, l, npix, c, apply_lrelu, stream,
                           "ugn_bf_setmax_bwd_routed_multi");
}

extern "C" int ugn_bf_lrelu_bwd_multi(const uint16_t* const* g, const uint16_t* const* act, uint16_t* const* out, const size_t* npix,
                                      int njobs, int c, void* stream) {
  UGN_REQUIRE(g && act && out && npix && njobs >= 1 && njobs <= kJobs, "ugn_bf_lrelu_bwd_multi: bad arguments");
  UGN_REQUIRE(c > 0 && c % 4 == 0, "ugn_bf_lrelu_bwd_multi: c must be a multiple of 4");
  EltJobs jt = {};
  size_t nmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(g[j] && act[j] && out[j] && npix[j] > 0, "ugn_bf_lrelu_bwd_multi: bad job %d", j);
    jt.g[j] = g[j]; jt.act[j] = act[j]; jt.out[j] = out[j]; jt.npix[j] = npix[j];
    if (npix[j] > nmax) nmax = npix[j];
  }
  hipLaunchKernelGGL(lrelu_bwd_bf_kernel, dim3((unsigned)((nmax * (c / 4) + 255) / 256), njobs), dim3(256), 0, (hipStream_t)stream, jt, c);
  UGN_CHECK_LAUNCH("bf_lrelu_bwd");
  return 0;
}

/* fp32 -> bf16 (round to nearest even) for up to 6 tensors of n[j] elements (multiples of 4) */
extern "C" int ugn_bf_convert_multi(const float* const* x, uint16_t* const* y, const size_t* n, int njobs, void* stream) {
  UGN_REQUIRE(x && y && n && njobs >= 1 && njobs <= kJobs, "ugn_bf_convert_multi: bad arguments");
  CvtJobs jt = {};
  size_t nmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(x[j] && y[j] && n[j] > 0 && n[j] % 4 == 0, "ugn_bf_convert_multi: bad job %d", j);
    jt.x[j] = x[j]; jt.y[j] = y[j]; jt.n[j] = n[j];
    if (n[j] > nmax) nmax = n[j];
  }
  const size_t blocks = (nmax / 4 + 255) / 256;
  hipLaunchKernelGGL(cvt_multi_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), njobs), dim3(256), 0, (hipStream_t)stream, jt);
  UGN_CHECK_LAUNCH("bf_convert");
  return 0;
}
