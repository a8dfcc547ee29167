// Direct 3x3 convolution (forward / data gradient) of fp32 tensors on v_mfma_f32_16x16x32_bf16 through the exact three-way bf16
// split of both operands (x3_common.h): Conv2D + LeakyReLU (+ MaxPool) and Conv2DBackpropInput (+ MaxPoolGrad, LeakyReluGrad) of
// nets/mj_uwyhNets_ba.py:431-462 with IEEE fp32 tensors in HBM and fp32-grade results at 6/16 of the fp32 matrix time.
//
// Implicit GEMM, M = output channels (the MFMA's A side), N = pixels (B side), K = 9 taps x input channels in 32-channel chunks.
// A 512-thread persistent workgroup (8 waves, two per SIMD, one workgroup per CU) owns a 16x16-pixel region and ALL NC output
// channels; wave = ONE 16-channel tile x ROWS = 4 / 8 / 16 rows of 16 pixels (NC = 32 / 64 / 128), so a lane's accumulator of a
// row holds FOUR CONSECUTIVE CHANNELS of one pixel: LeakyReLU, LeakyReLU' and the stores are 16 bytes per lane, a 2x2 pooling
// window is two rows of the lane and its neighbour (one DPP exchange).
//   * the chunk's 18x18 halo tile lies in LDS as three bf16 PLANES [plane][pixel][32 channels] (64 B per pixel and plane), double
//     buffered.  It gets there through registers: fp32 global loads a chunk ahead (two float4 = 8 channels of a pixel per unit,
//     three units per thread), the three-way split (11 vector instructions per channel pair) and three ds_write_b128 behind the
//     second tap column -- about a tenth of the chunk's matrix time, beside it.  A pooled input (MaxPool backward) is scattered
//     on the way: the pooled pixel's value goes where its argmax byte points, zeros elsewhere.
//   * a B fragment = 16 pixels of ONE halo row shifted by dx: one ds_read_b128 per plane, 1 KB contiguous per wave; the 16-byte
//     k group kg of the pixel in tile column c sits in slot kg ^ (((c >> 2) & 1) << 1), which makes the lane groups the LDS
//     services together ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md) hit 16 different 16-byte bank columns
//     for every dx.  A row's three planes serve the taps dy = 0, 1, 2 of three output rows: 3 reads per 18 MFMAs.
//   * the A fragments (filter: 16 channels x 32 k x 3 planes = 3 KB per tap and wave) come from L2 straight into registers, the
//     three taps of a column one column ahead (ugn_x3_pack_multi stores them lane-linear in exactly that order); layers with ONE
//     K chunk (32 input channels) keep all nine taps in registers for a whole job.  No filter in LDS, ONE barrier per K chunk.
#include "x3_common.h"

using namespace ugn_x3;

namespace {

constexpr int X3_PLANE = 18 * 18 * 64;     // 20,736 B: 324 halo pixels x 32 bf16
constexpr int X3_BUF = 3 * X3_PLANE;       // 62,208
constexpr int X3_LDS = 2 * X3_BUF;         // 124,416
constexpr int X3_UNITS = 18 * 18 * 4;      // staging units: (halo pixel, 8-channel group)

__host__ __device__ constexpr int x3_slot(int kg, int col) { return kg ^ (((col >> 2) & 1) << 1); }

// timing-only ablations (WRONG results): 1 no MFMA, 2 no split / LDS writes of the next tile, 4 no epilogue, 8 no tile loads, 16 no barrier
#ifndef UGN_X3_ABL
#define UGN_X3_ABL 0
#endif
#ifndef UGN_X3_ORDER
#define UGN_X3_ORDER 0      /* 1: the six products of a row's three taps product-major (consecutive MFMAs on different accumulators) */
#endif
#ifndef UGN_X3_SGB
#define UGN_X3_SGB 4        /* vector instructions of the split scheduled behind each MFMA of a staging row (0: hipcc's order) */
#endif

// diagnostic build only (-DUGN_X3_STAMP=1, tools/stamp_x3.py): s_memtime of wave 0 at the phases of its first items, workgroups 0..31
#ifndef UGN_X3_STAMP
#define UGN_X3_STAMP 0
#endif
#if UGN_X3_STAMP
constexpr int kStampWgs = 32, kStampItems = 48;
__device__ unsigned long long ugn_x3_stamp_buf[kStampWgs * kStampItems * 8];
#define X3_STAMP(k_)                                                                                                      \
  do {                                                                                                                    \
    if (tid == 0 && blockIdx.x < kStampWgs && nstamp < kStampItems)                                                       \
      ugn_x3_stamp_buf[(blockIdx.x * kStampItems + nstamp) * 8 + (k_)] = (k_) == 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define X3_STAMP(k_) do { } while (0)
#endif

enum { EPI_LRELU = 0, EPI_LRELU_POOL = 1, EPI_DGRAD = 2, EPI_DGRAD_ACT = 3 };

struct X3Job {
  const float* in;          // [n][hw][hw][kc]   (pooled input: [n][hw/2][hw/2][kc])
  const uint8_t* in_idx;    // pooled input: argmax bytes [n][hw/2][hw/2][kc]
  const uint16_t* wpk;      // packed filter planes (ugn_x3_pack_multi)
  float* out;               // [n][ho][ho][nc]
  uint8_t* out_idx;         // pooled epilogue: argmax bytes
  const float* act;         // data gradient: the layer's input activation (LeakyReLU')
};
struct X3Jobs {
  X3Job job[kMaxJobs];
  int start[kMaxJobs + 1];
};
__device__ __forceinline__ int job_of(const X3Jobs& jt, int it) {
  int jb = 0;
#pragma unroll
  for (int j = 1; j < kMaxJobs; ++j) jb += it >= jt.start[j] ? 1 : 0;
  return jb;
}

__device__ __forceinline__ float dpp_swap1(float v) {       // value of lane ^ 1 (quad_perm [1, 0, 3, 2])
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
}

// ---- filter packing ---------------------------------------------------------------------------------------------------
// HWIO fp32 -> [chunk][dx][dy][16-channel tile][plane][lane-linear 1 KB]: lane l of a fragment holds output channel
// 16 tile + (l & 15) at the 8 reduction channels 32 chunk + 8 (l >> 4) + e.  dgrad = 1: the flipped, transposed filter of the
// data gradient (reduction over the layer's output channels).
constexpr int kPackJobs = 64;
struct PackTable {
  const float* w[kPackJobs];
  uint16_t* pk[kPackJobs];
  int cin[kPackJobs], cout[kPackJobs], dgrad[kPackJobs];
  int start[kPackJobs + 1];          // first thread of each job
};
__global__ __launch_bounds__(256) void x3_pack_kernel(const PackTable t, int njobs) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= t.start[njobs]) return;
  int j = 0;
  while (j + 1 < njobs && g >= t.start[j + 1]) ++j;
  const int r = g - t.start[j];
  const int cin = t.cin[j], cout = t.cout[j], dg = t.dgrad[j];
  const int nc = dg ? cin : cout;
  const int nt = nc / 16;
  const int l = r & 63, f = r >> 6;
  const int ng = f % nt, tap = (f / nt) % 9, chunk = f / (nt * 9);
  const int dx = tap / 3, dy = tap % 3;            // [dx][dy] order
  const int col = 16 * ng + (l & 15);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * chunk + 8 * (l >> 4) + e;
    v[e] = dg ? t.w[j][(((2 - dy) * 3 + (2 - dx)) * cin + col) * cout + k] : t.w[j][((dy * 3 + dx) * cin + k) * cout + col];
  }
  uint4 p0, p1, p2;
  split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p0, p1, p2);
  char* dst = reinterpret_cast<char*>(t.pk[j]) + (size_t)f * 3072 + l * 16;
  *reinterpret_cast<uint4*>(dst) = p0;
  *reinterpret_cast<uint4*>(dst + 1024) = p1;
  *reinterpret_cast<uint4*>(dst + 2048) = p2;
}

// the split itself, for tests and for anyone who wants to see the format: planes[k][i] = bf16 bit pattern of x_k[i], k = 0, 1, 2
__global__ __launch_bounds__(256) void x3_split_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes, size_t n) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= n) return;
  const float a = x[i], b = i + 1 < n ? x[i + 1] : 0.f;
  unsigned p0, p1, p2;
  split2(a, b, p0, p1, p2);
  const unsigned p[3] = {p0, p1, p2};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    planes[k * n + i] = (uint16_t)(p[k] & 0xffffu);
    if (i + 1 < n) planes[k * n + i + 1] = (uint16_t)(p[k] >> 16);
  }
}

// ---- the convolution --------------------------------------------------------------------------------------------------
// WAVES = 8: one workgroup per CU, the halo tile double buffered, the next tile split and written between the MFMAs of the current one.
// WAVES = 4: TWO independent workgroups per CU (62 KB of LDS each, one halo buffer): a workgroup multiplies a tile, then -- behind a
// second barrier -- splits and writes the next one and runs its epilogue while the OTHER workgroup multiplies.  Inside one workgroup the
// matrix work and everything else add up (the waves move in lock-step from barrier to barrier; re-ordering the same work changes
// nothing: tools/bench_x3.py ablations in profiles/r05_x3_experiments.txt); two workgroups drift into anti-phase.
// FORM: 0 = 8 waves, one workgroup per CU, double-buffered halo; 1 = 4 waves, two workgroups per CU, one halo buffer each;
//       2 = 8 waves, TWO workgroups per CU (four waves per SIMD, <= 128 registers each), one halo buffer each
template <int KC, int NC, int HW, int EPI, int IN_POOLED, int FORM, int NP>
__global__ __launch_bounds__(FORM == 1 ? 256 : 512, FORM == 2 ? 4 : 2) void conv_x3_kernel(const X3Jobs jt) {
  constexpr int WAVES = FORM == 1 ? 4 : 8;
  constexpr int NTHR = 64 * WAVES;
  constexpr bool DB = FORM == 0;               // double-buffered halo, staging interleaved with the MFMAs
  constexpr int NT = NC / 16, MPARTS = WAVES / NT, ROWS = 16 / MPARTS, NCHUNK = KC / 32;
  constexpr bool RES = NCHUNK == 1 && DB;      // the job's whole filter tile stays in registers
  constexpr int RPX = HW / 16, RPI = RPX * RPX;
  static_assert((NC == 32 || NC == 64 || NC == 128) && MPARTS >= 1 && ROWS * MPARTS == 16, "a wave = one 16-channel tile x ROWS rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ng = wave % NT, mh = wave / NT;           // channel tile, row part
  const int x = lane & 15, kg = lane >> 4;            // B side: pixel x of a row, k group kg.  D side: pixel x, channels 4 kg .. + 3
  int ab[3];                                          // [dx]: plane 0 of the current buffer
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) ab[dx] = ((mh * ROWS) * 18 + x + dx) * 64 + (x3_slot(kg, x + dx) << 4);

  const int nitems = jt.start[kMaxJobs];
  const ItemRange ir = xcd_items(nitems);
  int item = ir.item;
  if (item >= ir.end) return;
  int jb = job_of(jt, item), lit = item - jt.start[jb];

  // ---- staging of an un-pooled tile: unit u = tid + NTHR s -> halo pixel u >> 2, channels 8 (u & 3) .. + 7 of the chunk.  The 240
  // slots beyond the 1296 units repeat units 1056 .. 1295 (same loads, same values to the same addresses): every thread runs the
  // same straight-line code, so the split can be scheduled between the MFMAs of a row instead of behind a branch
  constexpr int NSLOT = IN_POOLED ? (400 + NTHR - 1) / NTHR : (X3_UNITS + NTHR - 1) / NTHR;      // 3 (8 waves) or 6; pooled: 1 or 2
  int st_lds[IN_POOLED ? 1 : NSLOT], st_g[IN_POOLED ? 1 : NSLOT], st_rc[IN_POOLED ? 1 : NSLOT];
  if constexpr (!IN_POOLED) {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      int u = tid + NTHR * s;
      if (u >= X3_UNITS) u -= NSLOT * NTHR - X3_UNITS;
      const int hp = u >> 2, skg = u & 3;
      const int hr = hp / 18, hc = hp - hr * 18;
      st_lds[s] = hp * 64 + (x3_slot(skg, hc) << 4);
      st_g[s] = ((hr - 1) * HW + (hc - 1)) * KC + skg * 8;
      st_rc[s] = (hr << 8) | hc;
    }
  }
  float4 sv[NSLOT][2];      // (4 waves, rounds of SROUND slots: the compiler reuses the registers of a finished round)
  auto stage_load = [&](const X3Job& J, int lit_, int chunk, int sa, int sb) {
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    const float* base = J.in + ((size_t)img * HW * HW + (size_t)(ry0 * HW + rx0)) * KC + chunk * 32;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      if (s < sa || s >= sb) continue;
      const int y = ry0 + (st_rc[IN_POOLED ? 0 : s] >> 8) - 1, xx = rx0 + (st_rc[IN_POOLED ? 0 : s] & 255) - 1;
      const bool ok = (unsigned)y < (unsigned)HW && (unsigned)xx < (unsigned)HW;
      sv[s][0] = make_float4(0.f, 0.f, 0.f, 0.f);
      sv[s][1] = sv[s][0];
      if (ok) {
        sv[s][0] = *reinterpret_cast<const float4*>(base + st_g[IN_POOLED ? 0 : s]);
        sv[s][1] = *reinterpret_cast<const float4*>(base + st_g[IN_POOLED ? 0 : s] + 4);
      }
    }
  };
  auto stage_store_unit = [&](int buf, int s) {            // split + write unit s of the staged tile
    char* dst = smem + buf * X3_BUF;
    uint4 p0, p1, p2;
    split8(sv[s][0], sv[s][1], p0, p1, p2);
    *reinterpret_cast<uint4*>(dst + st_lds[IN_POOLED ? 0 : s]) = p0;
    *reinterpret_cast<uint4*>(dst + X3_PLANE + st_lds[IN_POOLED ? 0 : s]) = p1;
    *reinterpret_cast<uint4*>(dst + 2 * X3_PLANE + st_lds[IN_POOLED ? 0 : s]) = p2;
  };
  // ---- staging of a pooled tile (MaxPool backward): unit = (one of the 10 x 10 pooled pixels under the halo, 8 channels); slots
  // beyond the 400 units repeat units 288 .. 399, and a window position outside the halo repeats the unit's position inside it
  int sprow[IN_POOLED ? NSLOT : 1], spcol[IN_POOLED ? NSLOT : 1], scg[IN_POOLED ? NSLOT : 1];
  int pl_lds[IN_POOLED ? NSLOT : 1][4], pl_pos[IN_POOLED ? NSLOT : 1][4];
  uint2 pidx[IN_POOLED ? NSLOT : 1];
  if constexpr (IN_POOLED) {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      int pu = tid + NTHR * s;
      if (pu >= 400) pu -= NSLOT * NTHR - 400;
      const int spp = pu >> 2;
      scg[s] = pu & 3;
      sprow[s] = (spp * 205) >> 11;
      spcol[s] = spp - sprow[s] * 10;
#pragma unroll
      for (int pos = 0; pos < 4; ++pos) {
        int hy = 2 * sprow[s] - 1 + (pos >> 1), hx = 2 * spcol[s] - 1 + (pos & 1);
        hy = hy < 0 ? 0 : (hy > 17 ? 17 : hy);
        hx = hx < 0 ? 0 : (hx > 17 ? 17 : hx);
        pl_lds[s][pos] = (hy * 18 + hx) * 64 + (x3_slot(scg[s], hx) << 4);
        pl_pos[s][pos] = (((hy + 1) & 1) << 1) | ((hx + 1) & 1);
      }
      pidx[s] = make_uint2(0u, 0u);
    }
  }
  auto pool_load = [&](const X3Job& J, int lit_, int chunk, int sa, int sb) {
    constexpr int HP = HW / 2;
    const int img = lit_ / RPI, rrem = lit_ % RPI;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      if (s < sa || s >= sb) continue;
      const int pr = ((rrem / RPX) * 16) / 2 - 1 + sprow[IN_POOLED ? s : 0], pc = ((rrem % RPX) * 16) / 2 - 1 + spcol[IN_POOLED ? s : 0];
      const bool ok = (unsigned)pr < (unsigned)HP && (unsigned)pc < (unsigned)HP;
      sv[s][0] = make_float4(0.f, 0.f, 0.f, 0.f);      // (outside the image the gradient is zero: the scatter still overwrites the stale halo)
      sv[s][1] = sv[s][0];
      pidx[IN_POOLED ? s : 0] = make_uint2(0u, 0u);
      if (ok) {
        const size_t o = (size_t)img * HP * HP + (size_t)(pr * HP + pc);
        const float* v = J.in + o * KC + chunk * 32 + scg[IN_POOLED ? s : 0] * 8;
        sv[s][0] = *reinterpret_cast<const float4*>(v);
        sv[s][1] = *reinterpret_cast<const float4*>(v + 4);
        pidx[IN_POOLED ? s : 0] = *reinterpret_cast<const uint2*>(J.in_idx + o * KC + chunk * 32 + scg[IN_POOLED ? s : 0] * 8);
      }
    }
  };
  auto pool_store = [&](int buf, int s) {
    uint4 q0, q1, q2;
    split8(sv[s][0], sv[s][1], q0, q1, q2);
    const unsigned a0[4] = {q0.x, q0.y, q0.z, q0.w}, a1[4] = {q1.x, q1.y, q1.z, q1.w}, a2[4] = {q2.x, q2.y, q2.z, q2.w};
    const uint2 pix = pidx[IN_POOLED ? s : 0];
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      const unsigned ps = (unsigned)pl_pos[IN_POOLED ? s : 0][pos];
      unsigned m[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned w = d < 2 ? pix.x : pix.y;
        const unsigned b0 = (w >> (16 * (d & 1))) & 0xffu, b1 = (w >> (16 * (d & 1) + 8)) & 0xffu;
        m[d] = (b0 == ps ? 0x0000ffffu : 0u) | (b1 == ps ? 0xffff0000u : 0u);
      }
      char* rec = smem + buf * X3_BUF + pl_lds[IN_POOLED ? s : 0][pos];
      *reinterpret_cast<uint4*>(rec) = make_uint4(a0[0] & m[0], a0[1] & m[1], a0[2] & m[2], a0[3] & m[3]);
      *reinterpret_cast<uint4*>(rec + X3_PLANE) = make_uint4(a1[0] & m[0], a1[1] & m[1], a1[2] & m[2], a1[3] & m[3]);
      *reinterpret_cast<uint4*>(rec + 2 * X3_PLANE) = make_uint4(a2[0] & m[0], a2[1] & m[1], a2[2] & m[2], a2[3] & m[3]);
    }
  };
  // ---- the tile sequence of this workgroup: (item, chunk), chunk fastest.  Tile t + 1 is split and written while (8 waves) or after
  // (4 waves) tile t is multiplied; the fp32 values of tile t + 2 are fetched right behind that (two tiles of latency cover).
  struct Tl { int item, chunk; };
  auto tl_next = [&](Tl t) {
    Tl r;
    r.item = t.chunk + 1 < NCHUNK ? t.item : t.item + ir.stride;
    r.chunk = t.chunk + 1 < NCHUNK ? t.chunk + 1 : 0;
    return r;
  };
  auto tile_load = [&](Tl t, int sa = 0, int sb = 64) {
    if (t.item >= ir.end) return;
    const int j = job_of(jt, t.item);
    if constexpr (IN_POOLED) pool_load(jt.job[j], t.item - jt.start[j], t.chunk, sa, sb);
    else stage_load(jt.job[j], t.item - jt.start[j], t.chunk, sa, sb);
  };
  auto tile_store_unit = [&](int buf, int u) {
    if constexpr (IN_POOLED) pool_store(buf, u);
    else stage_store_unit(buf, u);
  };

  // 4 waves: a tile is fetched, split and written in rounds of SROUND slots (the staging registers of a round are reused by the next)
  constexpr int SROUND = (!DB && !IN_POOLED && NC >= 64) ? 3 : NSLOT;
  auto tile_fetch_write = [&](Tl t) {
#pragma unroll
    for (int r0 = 0; r0 < NSLOT; r0 += SROUND) {
      tile_load(t, r0, r0 + SROUND);
      __builtin_amdgcn_sched_barrier(0);
#if !(UGN_X3_ABL & 2)
#pragma unroll
      for (int u = r0; u < r0 + SROUND && u < NSLOT; ++u) tile_store_unit(0, u);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- the wave's filter fragments: [dy][plane] of one tap column (streamed), or [dx][dy][plane] of the whole chunk (RES)
  // (form 2 -- four waves per SIMD, 128 registers -- does not prefetch the next column's fragments: the other waves hide the latency)
  constexpr bool WPF = !RES && FORM != 2;
  uint4 wr[RES ? 3 : 1][3][3];
  uint4 nb[WPF ? 3 : 1][3];
  auto w_ptr = [&](const uint16_t* wpk, int chunk, int dx) {
    return reinterpret_cast<const char*>(wpk) + ((size_t)((chunk * 3 + dx) * 3) * NT + ng) * 3072 + lane * 16;
  };
  auto load_col = [&](const uint16_t* wpk, int chunk, int dx, uint4 (&dst)[3][3]) {
    const char* p = w_ptr(wpk, chunk, dx);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) dst[dy][pl] = *reinterpret_cast<const uint4*>(p + dy * (NT * 3072) + pl * 1024);
  };
  int w_jb = -1;
  if constexpr (RES) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) load_col(jt.job[jb].wpk, 0, dx, wr[dx]);
    w_jb = jb;
  } else if constexpr (WPF) {
    load_col(jt.job[jb].wpk, 0, 0, nb);
  }

  Tl t_cur = {item, 0};
  tile_load(t_cur);
#pragma unroll
  for (int u = 0; u < NSLOT; ++u) tile_store_unit(0, u);
  if constexpr (DB) tile_load(tl_next(t_cur));
  int hbuf = 0;

#if UGN_X3_STAMP
  int nstamp = 0;
#endif
  for (; item < ir.end; item += ir.stride) {
    X3_STAMP(0);
    X3_STAMP(6);
    const int next_item = item + ir.stride;
    const bool more = next_item < ir.end;
    const int jn = more ? job_of(jt, next_item) : jb, nlit = more ? next_item - jt.start[jn] : lit;
    if constexpr (RES) {
      if (jb != w_jb) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) load_col(jt.job[jb].wpk, 0, dx, wr[dx]);
        w_jb = jb;
      }
    }
    f32x4 acc[ROWS];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int img = lit / RPI, rrem = lit % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    const int py0 = ry0 + mh * ROWS, ch0 = 16 * ng + 4 * kg;
    // LeakyReLU' operands of the first rows, fetched behind the last MFMAs (the rest in the epilogue)
    constexpr int APF = EPI == EPI_DGRAD_ACT && ROWS <= 8 && DB ? (IN_POOLED ? 2 : ROWS) : 0;      // (where registers are left)
    float4 actv[APF > 0 ? APF : 1];

#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool last_chunk = chunk + 1 == NCHUNK;
      const int n_chunk = last_chunk ? 0 : chunk + 1;
      const int nx_job = last_chunk ? jn : jb;
      const Tl t1 = tl_next(Tl{item, chunk});
      const bool have_t1 = t1.item < ir.end;
#if !(UGN_X3_ABL & 16)
      __syncthreads();                                      // the chunk's tile is complete (8 waves: nobody reads the other buffer any more)
#endif
      if (chunk == 0) X3_STAMP(1);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        uint4 cb[RES ? 1 : 3][3];
        if constexpr (WPF) {
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) cb[dy][pl] = nb[dy][pl];
          // (unconditional -- behind the last chunk of the last item it re-reads chunk 0: a conditional load is its own basic block)
          if (dx < 2) load_col(jt.job[jb].wpk, chunk, dx + 1, nb);
          else load_col(jt.job[nx_job].wpk, n_chunk, 0, nb);
        } else if constexpr (!RES) {
          load_col(jt.job[jb].wpk, chunk, dx, cb);
        }
        if constexpr (DB) {
          if (dx == 1 && have_t1 && !(UGN_X3_ABL & 8)) tile_load(tl_next(t1));     // the fp32 values of the tile after next (registers free again)
        }
        if constexpr (APF > 0) {
          if (dx == 2 && last_chunk) {
            const float* act = jt.job[jb].act + (size_t)img * HW * HW * NC + ch0;
#pragma unroll
            for (int m = 0; m < APF; ++m) actv[m] = *reinterpret_cast<const float4*>(act + (size_t)((py0 + m) * HW + rx0 + x) * NC);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int a0 = ab[dx];
        uint4 fa[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fa[0][pl] = *reinterpret_cast<const uint4*>(smem + a0 + pl * X3_PLANE);
#pragma unroll
        for (int j = 0; j < ROWS + 2; ++j) {
          if (j + 1 < ROWS + 2) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              fa[(j + 1) & 1][pl] = *reinterpret_cast<const uint4*>(smem + a0 + (j + 1) * (18 * 64) + pl * X3_PLANE);
          }
          __builtin_amdgcn_sched_barrier(0);
          int nmf = 0;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int m = j - dy;
            if (m >= 0 && m < ROWS) {
              nmf += NP;
#pragma unroll
              for (int i = 0; i < NP; ++i) {
#if UGN_X3_ABL & 1
                if constexpr (RES) acc[m][i & 3] += __uint_as_float(wr[dx][dy][prod_w<NP>(i)].x ^ fa[j & 1][prod_x<NP>(i)].x);
                else acc[m][i & 3] += __uint_as_float(cb[dy][prod_w<NP>(i)].x ^ fa[j & 1][prod_x<NP>(i)].x);
#else
                if constexpr (RES) acc[m] = mfma_bf(wr[dx][dy][prod_w<NP>(i)], fa[j & 1][prod_x<NP>(i)], acc[m]);
                else acc[m] = mfma_bf(cb[dy][prod_w<NP>(i)], fa[j & 1][prod_x<NP>(i)], acc[m]);
#endif
              }
            }
          }
          // 8 waves: the next tile's units are split and written behind the MFMAs of rows 1, 1 + ROWS / 4, ... of the FIRST column,
          // their vector instructions interleaved with the matrix ones (one MFMA, four VALU, ...)
          if constexpr (DB) {
            constexpr int USTEP = ROWS / 4;
            if (dx == 0 && j >= 1 && (j - 1) % USTEP == 0 && (j - 1) / USTEP < NSLOT) {
#if !(UGN_X3_ABL & 2)
              tile_store_unit(hbuf ^ 1, (j - 1) / USTEP);      // (no tile after this one: stale values into the idle buffer)
#endif
#pragma unroll
              for (int k = 0; k < 18; ++k) {
                if (k < nmf) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (DB) {
        hbuf ^= 1;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) ab[dx] += hbuf ? X3_BUF : -X3_BUF;
      } else {
        // 4 waves: everybody has finished reading the tile -> fetch, split and write the next one into the same buffer (the other
        // workgroup of the CU multiplies meanwhile; no staging register lives through the MFMA loop).  Behind the LAST chunk the
        // fetch is issued before the epilogue and the split follows it.
        __syncthreads();
        if (!last_chunk) tile_fetch_write(t1);
      }
    }
    // (forms 1: the next item's first tile is fetched before the epilogue and split behind it; form 2 -- 128 registers -- fetches it
    //  behind the epilogue, whose latency the CU's other waves cover)
    X3_STAMP(2);
    constexpr bool PRE = !DB && SROUND == NSLOT && FORM != 2;
    if constexpr (PRE) {
      if (more) tile_load(Tl{next_item, 0});
      __builtin_amdgcn_sched_barrier(0);
    }
    X3_STAMP(3);

#if UGN_X3_ABL & 4
    if (ry0 < 0) {
#else
    {
#endif
    // ---- epilogue: a lane holds channels 16 ng + 4 kg .. + 3 of pixel (py0 + m, rx0 + x)
    const X3Job& J = jt.job[jb];
    if constexpr (EPI == EPI_LRELU_POOL) {
      constexpr int HO = HW / 2;
      const int odd = x & 1;
      float* out = J.out + (size_t)img * HO * HO * NC + ch0 + 2 * odd;
      uint8_t* oi = J.out_idx + (size_t)img * HO * HO * NC + ch0 + 2 * odd;
#pragma unroll
      for (int m2 = 0; m2 < ROWS / 2; ++m2) {
        const f32x4 r0 = acc[2 * m2], r1 = acc[2 * m2 + 1];
        // the even lane keeps channels 0, 1 and receives the odd pixel's; the odd lane keeps channels 2, 3 and receives the even pixel's
        float own[2][2], got[2][2];                       // [channel][row]
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          own[c][0] = odd ? r0[2 + c] : r0[c];
          own[c][1] = odd ? r1[2 + c] : r1[c];
          got[c][0] = dpp_swap1(odd ? r0[c] : r0[2 + c]);
          got[c][1] = dpp_swap1(odd ? r1[c] : r1[2 + c]);
        }
        float best[2];
        unsigned bi[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          // window in row-major order: (row 0, even pixel), (row 0, odd pixel), (row 1, even), (row 1, odd)
          const float cand[4] = {odd ? got[c][0] : own[c][0], odd ? own[c][0] : got[c][0], odd ? got[c][1] : own[c][1],
                                 odd ? own[c][1] : got[c][1]};
          best[c] = cand[0];
          bi[c] = 0;
#pragma unroll
          for (int i = 1; i < 4; ++i)
            if (cand[i] > best[c]) { best[c] = cand[i]; bi[c] = i; }       // strict >: the FIRST maximum wins (TF MaxPoolGrad)
          best[c] = ugn_lrelu(best[c]);                                    // (LeakyReLU is increasing: max first, activation once)
        }
        const unsigned pix = (unsigned)((py0 / 2 + m2) * HO + ((rx0 + x) >> 1));
        *reinterpret_cast<float2*>(out + (size_t)pix * NC) = make_float2(best[0], best[1]);
        *reinterpret_cast<uint16_t*>(oi + (size_t)pix * NC) = (uint16_t)(bi[0] | (bi[1] << 8));
      }
    } else {
      float* out = J.out + (size_t)img * HW * HW * NC + ch0;
      const float* act = nullptr;
      if constexpr (EPI == EPI_DGRAD_ACT) act = J.act + (size_t)img * HW * HW * NC + ch0;
#pragma unroll
      for (int m = 0; m < ROWS; ++m) {
        const size_t pix = (size_t)((py0 + m) * HW + rx0 + x) * NC;
        f32x4 v = acc[m];
        if constexpr (EPI == EPI_LRELU) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = ugn_lrelu(v[c]);
        } else if constexpr (EPI == EPI_DGRAD_ACT) {
          float4 a;
          if (m < APF) a = actv[m < APF ? m : 0];
          else a = *reinterpret_cast<const float4*>(act + pix);
          v[0] *= ugn_lrelu_slope(a.x);
          v[1] *= ugn_lrelu_slope(a.y);
          v[2] *= ugn_lrelu_slope(a.z);
          v[3] *= ugn_lrelu_slope(a.w);
        }
        *reinterpret_cast<float4*>(out + pix) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    }
    X3_STAMP(4);
    if constexpr (PRE) {
#if !(UGN_X3_ABL & 2)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NSLOT; ++u) tile_store_unit(0, u);      // (no next item: stale values into a buffer nobody reads)
#endif
    } else if constexpr (!DB) {
      if (more) tile_fetch_write(Tl{next_item, 0});
    }
    jb = jn;
    lit = nlit;
    X3_STAMP(5);
#if UGN_X3_STAMP
    ++nstamp;
#endif
  }
}

// persistent workgroups per CU-filling launch: ugn_set_persistent_wgs (conv3x3_mm.hip; the library's one process-wide setting), rounded
// down to a multiple of 8 (items are handed out per XCD).  Results do not depend on it (the items are the same, only who runs them).
int x3_grid() {
  const int g = ugn_mm::persistent_wgs() / 8 * 8;
  return g < 8 ? 8 : (g > kGrid ? kGrid : g);
}

inline int make_table(X3Jobs& jt, const X3Job* jobs, const int* n, int njobs, int per_img) {
  int total = 0;
  for (int j = 0; j < kMaxJobs; ++j) {
    jt.job[j] = jobs[j < njobs ? j : njobs - 1];
    jt.start[j] = total;
    if (j < njobs) total += n[j] * per_img;
  }
  jt.start[kMaxJobs] = total;
  return total;
}

// 4-wave form (two workgroups per CU) for the launches with <= 64 output columns, 8-wave form for 128 (a wave would own two column
// tiles: 128 accumulator registers beside 144 of filter fragments)
#ifndef UGN_X3_W4
#define UGN_X3_W4 1
#endif

#ifndef UGN_X3_FORM32
#define UGN_X3_FORM32 1      /* launches with 32 output columns (form 2, four waves per SIMD at <= 128 registers, measured 7 % slower) */
#endif
#ifndef UGN_X3_FORM64
#define UGN_X3_FORM64 -1     /* launches with 64 output columns: -1 = form 1 where it fits its registers, else form 0 */
#endif
template <int KC, int NC, int EPI, int IN_POOLED>
constexpr int x3_form() {
  if (!UGN_X3_W4 || NC > 64) return 0;
  if (NC == 32) return UGN_X3_FORM32;
  if (UGN_X3_FORM64 >= 0) return UGN_X3_FORM64;
  // (64 columns: 16 rows per wave = 64 accumulator registers; with the pooling epilogue, or four K chunks of streamed filters, the
  //  4-wave form spills 40-50 registers)
  if (!IN_POOLED && (EPI == EPI_LRELU_POOL || KC > 64)) return 0;
  return 1;
}

template <int KC, int NC, int HW, int EPI, int IN_POOLED, int NP>
int launch_x3_np(const X3Job* jobs, const int* n, int njobs, hipStream_t stream) {
  constexpr int FORM = x3_form<KC, NC, EPI, IN_POOLED>();
  constexpr int WAVES = FORM == 1 ? 4 : 8, WGS_PER_CU = FORM == 0 ? 1 : 2;
  constexpr int LDS = FORM == 0 ? X3_LDS : X3_BUF;
  static bool attr = false;
  auto* kern = conv_x3_kernel<KC, NC, HW, EPI, IN_POOLED, FORM, NP>;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      ugn_set_error("conv_x3_kernel: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(e));
      return (int)e;
    }
    attr = true;
  }
  X3Jobs jt;
  const int total = make_table(jt, jobs, n, njobs, (HW / 16) * (HW / 16));
  if (total == 0) return 0;
  hipLaunchKernelGGL(kern, dim3(x3_grid() * WGS_PER_CU), dim3(64 * WAVES), LDS, stream, jt);
  UGN_CHECK_LAUNCH("conv_x3_kernel");
  return 0;
}

template <int KC, int NC, int HW, int EPI, int IN_POOLED>
int launch_x3(const X3Job* jobs, const int* n, int njobs, int products, hipStream_t stream) {
  if (products == 9) return launch_x3_np<KC, NC, HW, EPI, IN_POOLED, 9>(jobs, n, njobs, stream);
  return launch_x3_np<KC, NC, HW, EPI, IN_POOLED, kProducts>(jobs, n, njobs, stream);
}

}  // namespace


#if UGN_X3_STAMP
extern "C" int ugn_x3_debug_stamps(unsigned long long* host_dst, int n) {
  hipError_t e = hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ugn_x3_stamp_buf), (size_t)n * sizeof(unsigned long long));
  void* p = nullptr;
  if (e == hipSuccess) e = hipGetSymbolAddress(&p, HIP_SYMBOL(ugn_x3_stamp_buf));
  if (e == hipSuccess) e = hipMemset(p, 0, sizeof(unsigned long long) * kStampWgs * kStampItems * 8);      // (the next launch starts from zeros)
  return (int)e;
}
#endif

extern "C" int ugn_x3_pack_multi(const float* const* w_hwio_host, uint16_t* const* wpk_host, const int* cin_host, const int* cout_host,
                                 const int* dgrad_host, int njobs, void* stream) {
  UGN_REQUIRE(njobs >= 1 && njobs <= kPackJobs, "ugn_x3_pack_multi: 1..%d jobs, got %d", kPackJobs, njobs);
  PackTable t;
  int total = 0;
  for (int j = 0; j < njobs; ++j) {
    const int cin = cin_host[j], cout = cout_host[j];
    UGN_REQUIRE(w_hwio_host[j] && wpk_host[j], "ugn_x3_pack_multi: null pointer in job %d", j);
    UGN_REQUIRE((cin == 32 || cin == 64 || cin == 128) && (cout == 32 || cout == 64 || cout == 128),
                "ugn_x3_pack_multi: channels must be 32 / 64 / 128, got %d -> %d", cin, cout);
    t.w[j] = w_hwio_host[j];
    t.pk[j] = wpk_host[j];
    t.cin[j] = cin;
    t.cout[j] = cout;
    t.dgrad[j] = dgrad_host[j] ? 1 : 0;
    t.start[j] = total;
    total += 9 * cin * cout / 8;          // one thread per 8 reduction channels of a (tap, output channel)
  }
  for (int j = njobs; j <= kPackJobs; ++j) t.start[j] = total;
  hipLaunchKernelGGL(x3_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, njobs);
  UGN_CHECK_LAUNCH("x3_pack_kernel");
  return 0;
}

extern "C" int ugn_x3_split(const float* x, uint16_t* planes, size_t n, void* stream) {
  UGN_REQUIRE(x && planes, "ugn_x3_split: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((n / 2 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, x, planes, n);
  UGN_CHECK_LAUNCH("x3_split_kernel");
  return 0;
}

#define X3F(KC_, NC_, HW_, P_)                                                                                          \
  if (cin == KC_ && cout == NC_ && hw == HW_ && (pool != 0) == (P_ != 0))                                               \
    return launch_x3<KC_, NC_, HW_, P_ ? EPI_LRELU_POOL : EPI_LRELU, 0>(jobs, n, njobs, products, (hipStream_t)stream);

extern "C" int ugn_x3_conv3x3_fwd_multi(const float* const* in, const uint16_t* const* wpk, float* const* out, uint8_t* const* out_idx,
                                        const int* n, int njobs, int hw, int cin, int cout, int pool, int products, void* stream) {
  UGN_REQUIRE(products == 6 || products == 9, "ugn_x3_conv3x3_fwd_multi: products must be 6 (default) or 9 (all partial products), got %d", products);
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_x3_conv3x3_fwd_multi: 1..%d jobs, got %d", kMaxJobs, njobs);
  X3Job jobs[kMaxJobs];
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(in[j] && wpk[j] && out[j] && (!pool || (out_idx && out_idx[j])) && n[j] >= 0,
                "ugn_x3_conv3x3_fwd_multi: null pointer or negative image count in job %d", j);
    jobs[j] = X3Job{in[j], nullptr, wpk[j], out[j], pool ? out_idx[j] : nullptr, nullptr};
  }
  X3F(32, 32, 64, 1)
  X3F(32, 64, 32, 0)
  X3F(64, 64, 32, 1)
  X3F(64, 128, 16, 0)
  X3F(128, 128, 16, 0)
  ugn_set_error("ugn_x3_conv3x3_fwd_multi: unsupported shape hw=%d cin=%d cout=%d pool=%d", hw, cin, cout, pool);
  return UGN_EINVAL;
}

// data gradient of the layer cin -> cout at hw x hw: reduction over cout, result [n][hw][hw][cin]
#define X3D(CI_, CO_, HW_, P_)                                                                                          \
  if (cin == CI_ && cout == CO_ && hw == HW_ && pooled == (P_ != 0)) {                                                  \
    if (with_act) return launch_x3<CO_, CI_, HW_, EPI_DGRAD_ACT, P_>(jobs, n, njobs, products, (hipStream_t)stream);              \
    return launch_x3<CO_, CI_, HW_, EPI_DGRAD, P_>(jobs, n, njobs, products, (hipStream_t)stream);                                \
  }

extern "C" int ugn_x3_conv3x3_dgrad_multi(const float* const* dz, const uint8_t* const* dz_idx, const uint16_t* const* wpk,
                                          const float* const* act, float* const* out, const int* n, int njobs, int hw, int cin,
                                          int cout, int products, void* stream) {
  UGN_REQUIRE(products == 6 || products == 9, "ugn_x3_conv3x3_dgrad_multi: products must be 6 (default) or 9 (all partial products), got %d", products);
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_x3_conv3x3_dgrad_multi: 1..%d jobs, got %d", kMaxJobs, njobs);
  const bool pooled = dz_idx != nullptr && dz_idx[0] != nullptr;
  const bool with_act = act != nullptr && act[0] != nullptr;
  X3Job jobs[kMaxJobs];
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(dz[j] && wpk[j] && out[j] && n[j] >= 0, "ugn_x3_conv3x3_dgrad_multi: null pointer or negative image count in job %d", j);
    UGN_REQUIRE((dz_idx != nullptr && dz_idx[j] != nullptr) == pooled && (act != nullptr && act[j] != nullptr) == with_act,
                "ugn_x3_conv3x3_dgrad_multi: all jobs or none take dz_idx / act (job %d differs)", j);
    jobs[j] = X3Job{dz[j], pooled ? dz_idx[j] : nullptr, wpk[j], out[j], nullptr, with_act ? act[j] : nullptr};
  }
  X3D(32, 32, 64, 1)
  X3D(32, 64, 32, 0)
  X3D(64, 64, 32, 1)
  X3D(64, 128, 16, 0)
  X3D(128, 128, 16, 0)
  ugn_set_error("ugn_x3_conv3x3_dgrad_multi: unsupported shape hw=%d cin=%d cout=%d pooled=%d", hw, cin, cout, (int)pooled);
  return UGN_EINVAL;
}
