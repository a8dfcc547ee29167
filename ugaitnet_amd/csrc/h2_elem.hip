// The HBM-bound steps between the 3x3 layers on H2 tensors (mm_common.h): set pooling over the L frames forward / backward
// (tf.math.reduce_max(axis=1) + Add, reference nets/mj_uwyhNets_ba.py:435,451-452,463-465), LeakyReLU', and the conversions at
// the edges of the H2 part of the path.  Same arithmetic as pool_set.hip on the values the halves hold; what is new is the
// block-exponent bookkeeping: an output's exponent comes from a bound formed from the inputs' metas, and its `amax` is that
// bound (these kernels gather no maximum: the next convolution measures its own output).
//
// A stored value v = H + L is an fp32 number with 22 significant bits, and splitting v again returns halves with the same
// sum (|L| <= ulp(H) / 2), so the kernels work on v in fp32 and re-split what they write.
#include "mm_common.h"
#include "../../include/ugaitnet_hip_h2.h"

using namespace ugn_mm;

namespace {

constexpr int kJobs = 6;
constexpr int MAXL = 32;
#ifndef UGN_SETMAX_PF
#define UGN_SETMAX_PF 4
#endif

// 4 channels of one pixel: two 8-byte loads (H, L) -> 4 stored values
struct V4 { float x, y, z, w; };
__device__ __forceinline__ V4 ld4(const uint16_t* __restrict__ rec, int c, int ch) {   // rec = the pixel's record [2][c]
  const uint2 hi = *reinterpret_cast<const uint2*>(rec + ch), lo = *reinterpret_cast<const uint2*>(rec + c + ch);
  return {h2_join0(hi.x, lo.x), h2_join1(hi.x, lo.x), h2_join0(hi.y, lo.y), h2_join1(hi.y, lo.y)};
}
__device__ __forceinline__ void st4(uint16_t* __restrict__ rec, int c, int ch, V4 v) {
  _Float16 h0, l0, h1, l1, h2, l2, h3, l3;
  h2_split(v.x, h0, l0); h2_split(v.y, h1, l1); h2_split(v.z, h2, l2); h2_split(v.w, h3, l3);
  *reinterpret_cast<uint2*>(rec + ch) = make_uint2(h2_pack(h0, h1), h2_pack(h2, h3));
  *reinterpret_cast<uint2*>(rec + c + ch) = make_uint2(h2_pack(l0, l1), h2_pack(l2, l3));
}
__device__ __forceinline__ V4 max4(V4 a, V4 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }

struct SetJobs {
  const uint16_t* p[kJobs];      // frames  H2 [b*l][pix][2][c]
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
  uint32_t* route[kJobs];        // routing words [b][pix][2][c] (fwd: written when set; bwd: read INSTEAD of p), see setmax_route
  int b[kJobs];
};

// Routing words of one (clip, pixel, channel): plane 0 = bit t set where frame t holds the maximum, plane 1 = bit t set where
// frame t is positive (its LeakyReLU' slope is 1).  8 bytes per 4 * l bytes of frames: with them the gradient kernel does not
// read the frames again -- a third (with an addend) to a half (without) of its bytes.
struct Route4 { uint4 mx, sg; };
__device__ __forceinline__ void route_bits(Route4& r, int t, V4 v, V4 m) {
  r.mx.x |= (v.x == m.x ? 1u : 0u) << t; r.mx.y |= (v.y == m.y ? 1u : 0u) << t;
  r.mx.z |= (v.z == m.z ? 1u : 0u) << t; r.mx.w |= (v.w == m.w ? 1u : 0u) << t;
  r.sg.x |= (v.x > 0.f ? 1u : 0u) << t; r.sg.y |= (v.y > 0.f ? 1u : 0u) << t;
  r.sg.z |= (v.z > 0.f ? 1u : 0u) << t; r.sg.w |= (v.w > 0.f ? 1u : 0u) << t;
}

// grid: (pixel-channel quads / 128, b, jobs).  npix = pixels per image, c = channels.
template <bool F32OUT>
__global__ __launch_bounds__(128) void setmax_fwd_h2_kernel(const SetJobs jt, int l, int npix, int c) {
  const int j = blockIdx.z, b = blockIdx.y;
  const int e = blockIdx.x * 128 + threadIdx.x, q = c / 4;
  if (b >= jt.b[j]) return;
  const int ep = jt.p_meta[j]->e;
  const bool has_add = jt.add[j] != nullptr;
  const int ea = has_add ? jt.add_meta[j]->e : 0;
  // exponent of the sum from the bound max|p| + max|addend|
  const float bp = h2_true_amax(ep, jt.p_meta[j]->amax), ba = has_add ? h2_true_amax(ea, jt.add_meta[j]->amax) : 0.f;
  const int eo = h2_exp_for_bound(bp + ba);
  if (e == 0 && b == 0 && !F32OUT) {
    if (jt.m[j]) *jt.m_meta[j] = *jt.p_meta[j];
    if (has_add) { H2Meta mo; mo.e = eo; mo.amax = __float_as_uint(ldexpf(bp + ba, eo)); *jt.sum_meta[j] = mo; }
  }
  if (e >= npix * q) return;
  const int pix = e / q, ch = (e - pix * q) * 4;
  const size_t rec = (size_t)2 * c;                                   // halves per pixel record
  const uint16_t* src = jt.p[j] + ((size_t)b * l * npix + pix) * rec;
  const size_t fstride = (size_t)npix * rec;
  V4 mx;
  if (jt.route[j]) {                 // one pass: a frame above the running maximum restarts its bit mask, an equal one joins it
    Route4 r = {make_uint4(1u, 1u, 1u, 1u), make_uint4(0u, 0u, 0u, 0u)};
    mx = ld4(src, c, ch);
    r.sg = make_uint4(mx.x > 0.f ? 1u : 0u, mx.y > 0.f ? 1u : 0u, mx.z > 0.f ? 1u : 0u, mx.w > 0.f ? 1u : 0u);
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
    for (; t + UGN_SETMAX_PF <= l; t += UGN_SETMAX_PF) {       // UGN_SETMAX_PF frames (2 loads of 8 bytes each) in flight per lane
      V4 vv[UGN_SETMAX_PF];
#pragma unroll
      for (int i = 0; i < UGN_SETMAX_PF; ++i) vv[i] = ld4(src + (size_t)(t + i) * fstride, c, ch);
#pragma unroll
      for (int i = 0; i < UGN_SETMAX_PF; ++i) take4(vv[i], t + i);
    }
    for (; t < l; ++t) take4(ld4(src + (size_t)t * fstride, c, ch), t);
    uint32_t* rp = jt.route[j] + ((size_t)b * npix + pix) * 2 * c + ch;
    *reinterpret_cast<uint4*>(rp) = r.mx;
    *reinterpret_cast<uint4*>(rp + c) = r.sg;
  } else {
    mx = ld4(src, c, ch);
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
    if constexpr (F32OUT) {
      const float fp = ldexpf(1.f, -ep), fa = ldexpf(1.f, -ea);
      sm = {mx.x * fp + a.x * fa, mx.y * fp + a.y * fa, mx.z * fp + a.z * fa, mx.w * fp + a.w * fa};
    } else {
      const float fp = ldexpf(1.f, eo - ep), fa = ldexpf(1.f, eo - ea);
      sm = {mx.x * fp + a.x * fa, mx.y * fp + a.y * fa, mx.z * fp + a.z * fa, mx.w * fp + a.w * fa};
    }
  }
  if constexpr (F32OUT) {
    const float fp = ldexpf(1.f, -ep);
    if (jt.m_f32[j]) *reinterpret_cast<float4*>(jt.m_f32[j] + o * c + ch) = make_float4(mx.x * fp, mx.y * fp, mx.z * fp, mx.w * fp);
    if (has_add) *reinterpret_cast<float4*>(jt.sum_f32[j] + o * c + ch) = make_float4(sm.x, sm.y, sm.z, sm.w);
  } else {
    if (jt.m[j]) st4(jt.m[j] + o * rec, c, ch, mx);
    if (has_add) st4(jt.sum[j] + o * rec, c, ch, sm);
  }
}

// TF reduce_max gradient (equal split among the maxima) + the second gradient path + LeakyReLU'(p), as setmax_bwd_kernel of
// pool_set.hip:   out = ((p == max ? dm / #maxima : 0) + addend) * LeakyReLU'(p)
template <bool DM_F32, bool ROUTED>
__global__ __launch_bounds__(128) void setmax_bwd_h2_kernel(const SetJobs jt, int l, int npix, int c, int lrelu) {
  const int j = blockIdx.z, b = blockIdx.y;
  const int e = blockIdx.x * 128 + threadIdx.x, q = c / 4;
  if (b >= jt.b[j]) return;
  const bool has_add = jt.add[j] != nullptr;
  const int edm = jt.dm_meta[j]->e, ea = has_add ? jt.add_meta[j]->e : 0;
  const float bd = h2_true_amax(edm, jt.dm_meta[j]->amax), ba = has_add ? h2_true_amax(ea, jt.add_meta[j]->amax) : 0.f;
  const int eo = h2_exp_for_bound(bd + ba);
  if (e == 0 && b == 0) { H2Meta mo; mo.e = eo; mo.amax = __float_as_uint(ldexpf(bd + ba, eo)); *jt.m_meta[j] = mo; }
  if (e >= npix * q) return;
  const int pix = e / q, ch = (e - pix * q) * 4;
  const size_t rec = (size_t)2 * c, fstride = (size_t)npix * rec;
  const size_t o = (size_t)b * npix + pix;
  // which frames hold the maximum / are positive: from the forward pass's routing words, or from the frames themselves
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
      if (t < l) route_bits(r, t, v[t], mx);
  }
  const V4 cnt = {(float)__popc(r.mx.x), (float)__popc(r.mx.y), (float)__popc(r.mx.z), (float)__popc(r.mx.w)};
  V4 g;
  if constexpr (DM_F32) {
    const float4 t4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(jt.dm[j]) + o * c + ch);
    g = {t4.x, t4.y, t4.z, t4.w};
  } else {
    g = ld4(reinterpret_cast<const uint16_t*>(jt.dm[j]) + o * rec, c, ch);
  }
  const float fg = ldexpf(1.f, eo - edm), fa = ldexpf(1.f, eo - ea);
  const V4 gs = {g.x * fg / cnt.x, g.y * fg / cnt.y, g.z * fg / cnt.z, g.w * fg / cnt.w};
  const uint16_t* asrc = has_add ? jt.add[j] + ((size_t)b * l * npix + pix) * rec : nullptr;
  uint16_t* dst = jt.m[j] + ((size_t)b * l * npix + pix) * rec;
  auto route = [&](unsigned mbits, unsigned sbits, int t, float gg, float aa) {
    float rr = ((mbits >> t) & 1u ? gg : 0.f) + aa;
    if (lrelu) rr *= (sbits >> t) & 1u ? 1.f : UGN_LRELU_ALPHA;
    return rr;
  };
  // (the addends of four frames in flight: with the routing words nothing else hides the latency of this read)
  constexpr int PF = 4;
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
        st4(dst + (size_t)t * fstride, c, ch, {route(r.mx.x, r.sg.x, t, gs.x, a[k].x * fa), route(r.mx.y, r.sg.y, t, gs.y, a[k].y * fa),
                                               route(r.mx.z, r.sg.z, t, gs.z, a[k].z * fa), route(r.mx.w, r.sg.w, t, gs.w, a[k].w * fa)});
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
__global__ void lrelu_bwd_h2_kernel(const EltJobs jt, int c) {
  const int j = blockIdx.y, q = c / 4;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) *jt.out_meta[j] = *jt.g_meta[j];
  if (e >= jt.npix[j] * q) return;
  const size_t pix = e / q;
  const int ch = (int)(e - pix * q) * 4;
  const size_t rec = (size_t)2 * c;
  const V4 g = ld4(jt.g[j] + pix * rec, c, ch), a = ld4(jt.act[j] + pix * rec, c, ch);
  st4(jt.out[j] + pix * rec, c, ch, {g.x * ugn_lrelu_slope(a.x), g.y * ugn_lrelu_slope(a.y), g.z * ugn_lrelu_slope(a.z),
                                     g.w * ugn_lrelu_slope(a.w)});
}

// ---- fp32 -> H2 for several tensors per launch (the gradients entering the convolution stack) ------------------------
struct CvtJobs {
  const float* x[kJobs];
  uint16_t* y[kJobs];          // encode only
  H2Meta* scratch[kJobs];      // {0, bits(max|x|)}: written by absmax_multi, read by encode_multi
  H2Meta* meta[kJobs];         // encode only: the tensor's final meta
  size_t n[kJobs];             // elements
};
__global__ void absmax_multi_kernel(const CvtJobs jt) {
  const int j = blockIdx.y;
  const float* x = jt.x[j];
  __shared__ float red[4];
  float m = 0.f;
  const size_t n4 = jt.n[j] / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < jt.n[j]; i += blockDim.x) m = fmaxf(m, fabsf(x[i]));
  h2_publish_amax_block(jt.scratch[j], m, red, threadIdx.x, 4);
}
__global__ void encode_multi_kernel(const CvtJobs jt, int c) {
  const int j = blockIdx.y;
  const float amax = __uint_as_float(jt.scratch[j]->amax);
  const int e = h2_exp_for_bound(amax);
  if (blockIdx.x == 0 && threadIdx.x == 0) { H2Meta mo; mo.e = e; mo.amax = __float_as_uint(ldexpf(amax, e)); *jt.meta[j] = mo; }
  const float f = ldexpf(1.f, e);
  const int q = c / 4;
  const size_t nq = jt.n[j] / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / q;
    const int ch = (int)(i - pix * q) * 4;
    const float4 v = *reinterpret_cast<const float4*>(jt.x[j] + pix * c + ch);
    st4(jt.y[j] + pix * 2 * c, c, ch, {v.x * f, v.y * f, v.z * f, v.w * f});
  }
}

}  // namespace

static int fill_set(SetJobs& jt, const uint16_t* const* p, const void* const* p_meta, const uint16_t* const* add,
                    const void* const* add_meta, int njobs, const int* b, int* bmax, const char* who) {
  UGN_REQUIRE(p && p_meta && b && njobs >= 1 && njobs <= kJobs, "%s: bad arguments (1..%d jobs)", who, kJobs);
  *bmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(p[j] && p_meta[j] && b[j] > 0, "%s: null pointer or b <= 0 in job %d", who, j);
    jt.p[j] = p[j]; jt.p_meta[j] = (const H2Meta*)p_meta[j];
    jt.add[j] = add ? add[j] : nullptr;
    jt.add_meta[j] = (add && add[j]) ? (const H2Meta*)add_meta[j] : nullptr;
    UGN_REQUIRE(!jt.add[j] || jt.add_meta[j], "%s: addend without its meta in job %d", who, j);
    jt.b[j] = b[j];
    if (b[j] > *bmax) *bmax = b[j];
  }
  return 0;
}

static int set_routes(SetJobs& jt, uint32_t* const* route, int njobs, int l, const char* who) {
  for (int j = 0; j < njobs; ++j) jt.route[j] = route ? route[j] : nullptr;
  UGN_REQUIRE(!route || l <= MAXL, "%s: routing words hold at most %d frames", who, MAXL);
  return 0;
}

/* maxima over the l frames of each clip (+ optional set-level addend): H2 outputs m (optional) and sum = m + addend.
 * route (optional; l <= 32): routing words u32 [b][npix][2][c] for ugn_h2_setmax_bwd_routed_multi */
extern "C" int ugn_h2_setmax_fwd_routed_multi(const uint16_t* const* p, const void* const* p_meta, const uint16_t* const* addend,
                                              const void* const* addend_meta, uint16_t* const* m, void* const* m_meta,
                                              uint16_t* const* sum, void* const* sum_meta, uint32_t* const* route, const int* b,
                                              int njobs, int l, int npix, int c, void* stream) {
  SetJobs jt = {};
  int bmax;
  if (int rc = fill_set(jt, p, p_meta, addend, addend_meta, njobs, b, &bmax, "ugn_h2_setmax_fwd_multi")) return rc;
  if (int rc = set_routes(jt, route, njobs, l, "ugn_h2_setmax_fwd_routed_multi")) return rc;
  UGN_REQUIRE(l > 0 && npix > 0 && c > 0 && c % 4 == 0, "ugn_h2_setmax_fwd_multi: c must be a multiple of 4");
  for (int j = 0; j < njobs; ++j) {
    jt.m[j] = m ? m[j] : nullptr; jt.m_meta[j] = (m && m[j]) ? (H2Meta*)m_meta[j] : nullptr;
    jt.sum[j] = sum ? sum[j] : nullptr; jt.sum_meta[j] = (sum && sum[j]) ? (H2Meta*)sum_meta[j] : nullptr;
    UGN_REQUIRE(!jt.m[j] || jt.m_meta[j], "ugn_h2_setmax_fwd_multi: m without meta (job %d)", j);
    UGN_REQUIRE(!jt.add[j] || (jt.sum[j] && jt.sum_meta[j]), "ugn_h2_setmax_fwd_multi: addend needs sum + meta (job %d)", j);
    UGN_REQUIRE(jt.m[j] || jt.add[j], "ugn_h2_setmax_fwd_multi: nothing to write (job %d)", j);
  }
  const unsigned gx = (unsigned)(((size_t)npix * (c / 4) + 127) / 128);
  hipLaunchKernelGGL(setmax_fwd_h2_kernel<false>, dim3(gx, bmax, njobs), dim3(128), 0, (hipStream_t)stream, jt, l, npix, c);
  UGN_CHECK_LAUNCH("h2_setmax_fwd");
  return 0;
}
extern "C" int ugn_h2_setmax_fwd_multi(const uint16_t* const* p, const void* const* p_meta, const uint16_t* const* addend,
                                       const void* const* addend_meta, uint16_t* const* m, void* const* m_meta,
                                       uint16_t* const* sum, void* const* sum_meta, const int* b, int njobs, int l, int npix,
                                       int c, void* stream) {
  return ugn_h2_setmax_fwd_routed_multi(p, p_meta, addend, addend_meta, m, m_meta, sum, sum_meta, nullptr, b, njobs, l, npix, c, stream);
}

/* the same with fp32 outputs [b][npix][c] (true values): the last set pooling feeds HPP, which stays fp32 */
extern "C" int ugn_h2_setmax_fwd_f32_routed_multi(const uint16_t* const* p, const void* const* p_meta, const uint16_t* const* addend,
                                                  const void* const* addend_meta, float* const* m, float* const* sum,
                                                  uint32_t* const* route, const int* b, int njobs, int l, int npix, int c,
                                                  void* stream) {
  SetJobs jt = {};
  int bmax;
  if (int rc = fill_set(jt, p, p_meta, addend, addend_meta, njobs, b, &bmax, "ugn_h2_setmax_fwd_f32_multi")) return rc;
  if (int rc = set_routes(jt, route, njobs, l, "ugn_h2_setmax_fwd_f32_routed_multi")) return rc;
  UGN_REQUIRE(l > 0 && npix > 0 && c > 0 && c % 4 == 0, "ugn_h2_setmax_fwd_f32_multi: c must be a multiple of 4");
  for (int j = 0; j < njobs; ++j) {
    jt.m_f32[j] = m ? m[j] : nullptr; jt.sum_f32[j] = sum ? sum[j] : nullptr;
    UGN_REQUIRE(!jt.add[j] || jt.sum_f32[j], "ugn_h2_setmax_fwd_f32_multi: addend needs sum (job %d)", j);
    UGN_REQUIRE(jt.m_f32[j] || jt.add[j], "ugn_h2_setmax_fwd_f32_multi: nothing to write (job %d)", j);
  }
  const unsigned gx = (unsigned)(((size_t)npix * (c / 4) + 127) / 128);
  hipLaunchKernelGGL(setmax_fwd_h2_kernel<true>, dim3(gx, bmax, njobs), dim3(128), 0, (hipStream_t)stream, jt, l, npix, c);
  UGN_CHECK_LAUNCH("h2_setmax_fwd_f32");
  return 0;
}
extern "C" int ugn_h2_setmax_fwd_f32_multi(const uint16_t* const* p, const void* const* p_meta, const uint16_t* const* addend,
                                           const void* const* addend_meta, float* const* m, float* const* sum, const int* b,
                                           int njobs, int l, int npix, int c, void* stream) {
  return ugn_h2_setmax_fwd_f32_routed_multi(p, p_meta, addend, addend_meta, m, sum, nullptr, b, njobs, l, npix, c, stream);
}

/* out = ((p == max over l ? dm / #maxima : 0) + addend) * (apply_lrelu ? LeakyReLU'(p) : 1).  dm: H2 [b][npix][2][c], or with
 * dm_is_f32 an fp32 tensor [b][npix][c] whose dm_meta is {0, bits(max|dm|)} (ugn_absmax_multi).  out may alias addend's data
 * (its meta must be a different record).  Which frames hold the maximum and which are positive comes from the frames p, or --
 * `route` given, p ignored (may be null) -- from the routing words the forward pass wrote: bit-identical results without reading
 * the l frames again. */
static int setmax_bwd_any(const uint16_t* const* p, const void* const* p_meta, const uint32_t* const* route, const void* const* dm,
                          const void* const* dm_meta, int dm_is_f32, const uint16_t* const* addend, const void* const* addend_meta,
                          uint16_t* const* out, void* const* out_meta, const int* b, int njobs, int l, int npix, int c, int apply_lrelu,
                          void* stream, const char* who) {
  SetJobs jt = {};
  int bmax = 0;
  if (route) {
    UGN_REQUIRE(b && njobs >= 1 && njobs <= kJobs, "%s: bad arguments (1..%d jobs)", who, kJobs);
    for (int j = 0; j < njobs; ++j) {
      UGN_REQUIRE(route[j] && b[j] > 0, "%s: null routing words or b <= 0 in job %d", who, j);
      jt.route[j] = const_cast<uint32_t*>(route[j]);
      jt.add[j] = addend ? addend[j] : nullptr;
      jt.add_meta[j] = (addend && addend[j]) ? (const H2Meta*)addend_meta[j] : nullptr;
      UGN_REQUIRE(!jt.add[j] || jt.add_meta[j], "%s: addend without its meta in job %d", who, j);
      jt.b[j] = b[j];
      if (b[j] > bmax) bmax = b[j];
    }
  } else if (int rc = fill_set(jt, p, p_meta, addend, addend_meta, njobs, b, &bmax, who)) {
    return rc;
  }
  UGN_REQUIRE(dm && dm_meta && out && out_meta, "%s: null array", who);
  UGN_REQUIRE(l > 0 && l <= MAXL && npix > 0 && c > 0 && c % 4 == 0, "%s: l must be 1..%d, c a multiple of 4", who, MAXL);
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(dm[j] && dm_meta[j] && out[j] && out_meta[j], "%s: null pointer in job %d", who, j);
    UGN_REQUIRE(out_meta[j] != (addend_meta ? addend_meta[j] : nullptr), "%s: out_meta must not be addend_meta (job %d)", who, j);
    jt.dm[j] = dm[j]; jt.dm_meta[j] = (const H2Meta*)dm_meta[j]; jt.m[j] = out[j]; jt.m_meta[j] = (H2Meta*)out_meta[j];
  }
  const dim3 grid((unsigned)(((size_t)npix * (c / 4) + 127) / 128), bmax, njobs);
  hipStream_t st = (hipStream_t)stream;
  if (route) {
    if (dm_is_f32) hipLaunchKernelGGL((setmax_bwd_h2_kernel<true, true>), grid, dim3(128), 0, st, jt, l, npix, c, apply_lrelu);
    else hipLaunchKernelGGL((setmax_bwd_h2_kernel<false, true>), grid, dim3(128), 0, st, jt, l, npix, c, apply_lrelu);
  } else {
    if (dm_is_f32) hipLaunchKernelGGL((setmax_bwd_h2_kernel<true, false>), grid, dim3(128), 0, st, jt, l, npix, c, apply_lrelu);
    else hipLaunchKernelGGL((setmax_bwd_h2_kernel<false, false>), grid, dim3(128), 0, st, jt, l, npix, c, apply_lrelu);
  }
  UGN_CHECK_LAUNCH("h2_setmax_bwd");
  return 0;
}
extern "C" int ugn_h2_setmax_bwd_multi(const uint16_t* const* p, const void* const* p_meta, const void* const* dm,
                                       const void* const* dm_meta, int dm_is_f32, const uint16_t* const* addend,
                                       const void* const* addend_meta, uint16_t* const* out, void* const* out_meta, const int* b,
                                       int njobs, int l, int npix, int c, int apply_lrelu, void* stream) {
  return setmax_bwd_any(p, p_meta, nullptr, dm, dm_meta, dm_is_f32, addend, addend_meta, out, out_meta, b, njobs, l, npix, c, apply_lrelu,
                        stream, "ugn_h2_setmax_bwd_multi");
}
extern "C" int ugn_h2_setmax_bwd_routed_multi(const uint32_t* const* route, const void* const* dm, const void* const* dm_meta,
                                              int dm_is_f32, const uint16_t* const* addend, const void* const* addend_meta,
                                              uint16_t* const* out, void* const* out_meta, const int* b, int njobs, int l, int npix,
                                              int c, int apply_lrelu, void* stream) {
  UGN_REQUIRE(route, "ugn_h2_setmax_bwd_routed_multi: null routing words");
  return setmax_bwd_any(nullptr, nullptr, route, dm, dm_meta, dm_is_f32, addend, addend_meta, out, out_meta, b, njobs, l, npix, c,
                        apply_lrelu, stream, "ugn_h2_setmax_bwd_routed_multi");
}

extern "C" int ugn_h2_lrelu_bwd_multi(const uint16_t* const* g, const void* const* g_meta, const uint16_t* const* act,
                                      uint16_t* const* out, void* const* out_meta, const size_t* npix, int njobs, int c,
                                      void* stream) {
  UGN_REQUIRE(g && g_meta && act && out && out_meta && npix && njobs >= 1 && njobs <= kJobs, "ugn_h2_lrelu_bwd_multi: bad arguments");
  UGN_REQUIRE(c > 0 && c % 4 == 0, "ugn_h2_lrelu_bwd_multi: c must be a multiple of 4");
  EltJobs jt = {};
  size_t nmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(g[j] && g_meta[j] && act[j] && out[j] && out_meta[j] && npix[j] > 0, "ugn_h2_lrelu_bwd_multi: bad job %d", j);
    jt.g[j] = g[j]; jt.g_meta[j] = (const H2Meta*)g_meta[j]; jt.act[j] = act[j]; jt.out[j] = out[j];
    jt.out_meta[j] = (H2Meta*)out_meta[j]; jt.npix[j] = npix[j];
    if (npix[j] > nmax) nmax = npix[j];
  }
  hipLaunchKernelGGL(lrelu_bwd_h2_kernel, dim3((unsigned)((nmax * (c / 4) + 255) / 256), njobs), dim3(256), 0, (hipStream_t)stream, jt, c);
  UGN_CHECK_LAUNCH("h2_lrelu_bwd");
  return 0;
}

/* meta[j] <- {0, bits(max|x[j]|)} for up to 6 fp32 tensors (metas zero on entry) */
extern "C" int ugn_absmax_multi(const float* const* x, const size_t* n, void* const* meta, int njobs, void* stream) {
  UGN_REQUIRE(x && n && meta && njobs >= 1 && njobs <= kJobs, "ugn_absmax_multi: bad arguments (1..%d jobs)", kJobs);
  CvtJobs jt = {};
  size_t nmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(x[j] && meta[j] && n[j] > 0, "ugn_absmax_multi: bad job %d", j);
    jt.x[j] = x[j]; jt.scratch[j] = (H2Meta*)meta[j]; jt.n[j] = n[j];
    if (n[j] > nmax) nmax = n[j];
  }
  const size_t blocks = (nmax + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL(absmax_multi_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256), njobs), dim3(256), 0, (hipStream_t)stream, jt);
  UGN_CHECK_LAUNCH("absmax_multi");
  return 0;
}

/* fp32 [npix][c] -> H2 for up to 6 tensors: amax_meta[j] = {0, bits(max|x[j]|)} from ugn_absmax_multi; meta[j] receives the
 * tensor's exponent and stored maximum */
extern "C" int ugn_h2_encode_multi(const float* const* x, const void* const* amax_meta, uint16_t* const* y, void* const* meta,
                                   const size_t* npix, int njobs, int c, void* stream) {
  UGN_REQUIRE(x && amax_meta && y && meta && npix && njobs >= 1 && njobs <= kJobs, "ugn_h2_encode_multi: bad arguments");
  UGN_REQUIRE(c > 0 && c % 4 == 0, "ugn_h2_encode_multi: c must be a multiple of 4");
  CvtJobs jt = {};
  size_t nmax = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(x[j] && amax_meta[j] && y[j] && meta[j] && npix[j] > 0 && amax_meta[j] != meta[j], "ugn_h2_encode_multi: bad job %d", j);
    jt.x[j] = x[j]; jt.scratch[j] = (H2Meta*)amax_meta[j]; jt.y[j] = y[j]; jt.meta[j] = (H2Meta*)meta[j]; jt.n[j] = npix[j] * (size_t)c;
    if (jt.n[j] > nmax) nmax = jt.n[j];
  }
  const size_t blocks = (nmax / 4 + 255) / 256;
  hipLaunchKernelGGL(encode_multi_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), njobs), dim3(256), 0, (hipStream_t)stream, jt, c);
  UGN_CHECK_LAUNCH("h2_encode_multi");
  return 0;
}
