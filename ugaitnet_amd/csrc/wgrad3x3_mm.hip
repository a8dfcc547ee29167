// Weight gradient of the 3x3 layers on H2 tensors (mm_common.h), v_mfma_f32_32x32x16_f16.  Replaces the implicit TF
// Conv2DBackpropFilter (+ MaxPoolGrad) of reference nets/mj_uwyhNets_ba.py:431-462, like wgrad3x3_wino.hip.
//
//   dW[tap][ci][co] = sum over images and pixels of  in[y + dy - 1][x + dx - 1][ci] * dz[y][x][co]
//
// Per tap a GEMM with M = ci, N = co and K = PIXELS.  Both operands sit in LDS exactly as HBM holds them (pixel-major, channels
// contiguous), so the MFMA fragments -- 8 consecutive k = 8 consecutive pixels of ONE channel per lane -- are read with
// ds_read_b64_tr_b16, gfx950's transposing LDS read (a 4-pixel x 16-channel block per 16 lanes, delivered channel-major).
//   * A 512-thread workgroup owns a block combination of 32 input x COW (32 or 64) output channels and all 9 taps: wave =
//     (pair of 32x32 blocks, K split): 9 accumulators = 144 registers; the K split of a strip is by pixel rows.
//   * Work items are 8-row x 16-column strips of an image (all jobs of the launch form one list; the groups of a combination
//     own equal contiguous shares of it).  Per strip: the 10 x 18 halo of the input chunk and the 8 x 16 gradient tile,
//     PLANAR in LDS ([plane][pixel][32 channels] = 64-byte rows: a half wave's transposed read then covers 256 contiguous
//     bytes -- conflict-free), fetched by LDS-DMA one strip ahead.  One k-step = one pixel row of 16: the gradient fragments
//     are read once and meet the 9 shifted input fragments: 27 MFMAs (H*H + H*L + L*H per tap).
//   * Pooled layers: the pooled gradient + argmax bytes are staged and scattered LDS -> LDS (MaxPool backward) per strip.
//   * At the end of a share (or a job boundary inside it) the K-split waves are added through LDS in a fixed order and the
//     partial sums leave as a slab; wgrad_mm_finish adds the slabs in a fixed order and applies 2^-(e_in + e_dz).  No atomics.
//   * Round 4: the un-pooled 64-wide launches multiply on v_mfma_f32_16x16x32_f16 (M16), and the two POOLED layers on the SPARSE
//     matrix pipe, v_smfmac_f32_16x16x64_f16 (SP = 1; 16-row strips): MaxPool backward is 2:4 sparse along a pixel row and the
//     compressed operand is the pooled gradient as stored -- see the SP branch of the strip loop.
#include "mm_common.h"
#include "../../include/ugaitnet_hip_h2.h"

using namespace ugn_mm;

namespace {

typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
#define LDS_PTR(T) __attribute__((address_space(3))) T*

constexpr int kWgMaxJobs = 6;
// Rows of a strip.  32 output channels (a2): one 32x32 block pair, the 8 waves split K, so an 8-row strip is ONE k-step (27 MFMAs,
// 864 cycles) per wave between two barriers, a tile wait and the DMA issue -- 1,730 of a 4,490-cycle strip period on the matrix
// pipe (tools/stamp_wgrad.py).  A 16-row strip gives the wave two k-steps for the same fixed costs.
#ifndef UGN_WG_SR32
#define UGN_WG_SR32 16
#endif

struct WgJob {
  const uint16_t* in;        // H2 [n][hw][hw][2][ci]
  const uint16_t* dz;        // H2 [n][hw][hw][2][co]   (pooled: [n][hw/2][hw/2][2][co])
  const uint8_t* dz_idx;     // pooled: argmax bytes
  float* slab;               // [combo][ng][9][32][COW]
  int g0, ng;                // the groups (of every combination) that touch this job
};
struct WgJobs {
  WgJob job[kWgMaxJobs];
  int start[kWgMaxJobs + 1]; // first strip of job j; [njobs..] = total
  int ngroups;               // groups per block combination (a multiple of 8): fixed per kernel shape
  int gstep;                 // groups per combination that the grid holds at once (= ngroups on the full grid)
};

// Pooled layers on the SPARSE matrix pipe (wgrad_mm_kernel<..., SP = 1>; see the strip loop): 16-row strips, a wave = 4 rows = K 64.
#ifndef UGN_WG_SPARSE
#define UGN_WG_SPARSE 1
#endif
#ifndef UGN_SP_ABL
#define UGN_SP_ABL 0          /* timing-only ablations of the sparse path (WRONG results): 1 no MFMA, 2 no dense-fragment reads, 4 no LDS-DMA */
#endif
#ifndef UGN_WG_SPD
#define UGN_WG_SPD 3          /* micro-steps (6 MFMAs) the dense fragments are read ahead: 32 output channels per workgroup ... */
#endif
#ifndef UGN_WG_SPD64
#define UGN_WG_SPD64 0        /* ... and 64 (144 accumulator registers: one step ahead spills 10 and measures the same) */
#endif
constexpr int wg_default_sr(int co) { return co >= 64 ? 8 : UGN_WG_SR32; }
template <int CO, int SRV = wg_default_sr(CO)>
struct WGeo {
  static constexpr int COW = CO >= 64 ? 64 : 32;     // output channels of a workgroup
  static constexpr int PW = COW / 32;                // 32x32 block pairs
  static constexpr int KS = 8 / PW;                  // waves sharing a pair (K split)
  static constexpr int SR = SRV;                     // pixel rows of a strip
  static constexpr int RPW = SR / KS;                // pixel rows of a strip per wave
  static constexpr int IN_PIX = (SR + 2) * 18;       // halo of the input chunk
  static constexpr int IN_PLANE = IN_PIX * 64;       // 11,520 B (8 rows)
  static constexpr int IN_SLOTS = 2 * IN_PIX * 4;    // 2 planes x pixels x 4 quarters: 1440 slots (8 rows) -> 22.5 pieces
  static constexpr int IN_PIECES = (IN_SLOTS + 63) / 64;
  static constexpr int IN_BYTES = IN_PIECES * 1024;
  static constexpr int DZ_BLOCK = SR * 1024;         // [plane][block][SR x 16 pixels][64 B]
  static constexpr int DZ_PLANE = PW * DZ_BLOCK;
  static constexpr int DZ_BYTES = 2 * DZ_PLANE;
  // pooled layers: the gradient tile is the POOLED one, [plane][block][SR / 2 x 8 pooled pixels][64 B], + [pooled pixels][COW] argmax bytes
  static constexpr int PZ_PIX = SR * 4;
  static constexpr int PZ_BLOCK = PZ_PIX * 64;
  static constexpr int PZ_PLANE = PW * PZ_BLOCK;
  static constexpr int PZ_VAL = 2 * PZ_PLANE;
  static constexpr int PZ_PIECES = PZ_VAL / 1024 + PZ_PIX * COW / 1024;      // values + argmax bytes (8 rows: 4 + 1 per block)
  static constexpr int PZ_BYTES = PZ_PIECES * 1024;
};
// buffer set = input halo + gradient tile (pooled: the pooled gradient tile); + 32 KB of scratch for the K-split combine where a
// set is smaller than that
template <int CO, int POOLED, int SRV = wg_default_sr(CO)>
constexpr int wg_set_bytes() { return WGeo<CO, SRV>::IN_BYTES + (POOLED ? WGeo<CO, SRV>::PZ_BYTES : WGeo<CO, SRV>::DZ_BYTES); }
// buffer sets: two (the next strip streams in while this one is multiplied); THREE for the sparse 32-channel kernel, which waits for
// HBM rather than for the matrix pipe (tile wait 800-1,600 cycles of a 6,500-cycle strip with one strip in flight): two strips ahead
#ifndef UGN_WG_SP3
#define UGN_WG_SP3 0       /* measured: 395 against 397 us -- the launch is not short of bytes in flight */
#endif
template <int CO, int SP>
constexpr int wg_nsets() { return (SP && CO < 64 && UGN_WG_SP3) ? 3 : 2; }
template <int CO, int POOLED, int SRV = wg_default_sr(CO), int NSET = 2>
constexpr int wg_lds_bytes() { return NSET * wg_set_bytes<CO, POOLED, SRV>() + (wg_set_bytes<CO, POOLED, SRV>() < 32768 ? 32768 : 0); }

__device__ __forceinline__ h8 tr_pair(const LDS_PTR(char) base, int off0, int off1) {
  const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(base + off0));
  const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(base + off1));
  const s8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(h8, v);
}
__device__ __forceinline__ f32x16 mfma_h8(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

#ifdef UGN_WG_STAMP
__device__ unsigned long long* g_wg_stamp = nullptr;
constexpr int kWgStampPerWave = 4 + 6 * 60;
#endif

typedef _Float16 h16 __attribute__((ext_vector_type(16)));

template <int CI, int CO, int HW, int POOLED, int SP = 0>
__global__ __launch_bounds__(512, 2) void wgrad_mm_kernel(const WgJobs jt, const void* __restrict__ zeros) {
  static_assert(!SP || POOLED, "the sparse form is the MaxPool-backward one");
  constexpr int SRV = SP ? 16 : wg_default_sr(CO);
  using G = WGeo<CO, SRV>;
  constexpr int COW = G::COW, PW = G::PW, KS = G::KS, RPW = G::RPW, SR = G::SR, SET = wg_set_bytes<CO, POOLED, SRV>();
  constexpr int IN_PIX = G::IN_PIX, IN_PLANE = G::IN_PLANE, IN_PIECES = G::IN_PIECES, IN_BYTES = G::IN_BYTES;
  constexpr int NCOC = CO / COW, NCOMBO = (CI / 32) * NCOC;
  constexpr int SPX = HW / 16, SPI = (HW / SR) * SPX;       // strips per image row / per image
  static_assert(HW % SR == 0 && (SR == 8 || SR == 16) && (SP || RPW * KS == SR) && IN_PIX < 400, "strip geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const LDS_PTR(char) lds = (LDS_PTR(char))smem;
  const unsigned sbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (combination, group): all combinations of a group sit on ONE XCD (blockIdx % 8), so the strip both read
  // comes from HBM once per XCD and from that XCD's L2 afterwards
  const int bid = blockIdx.x, xcd = bid & 7, rest = bid >> 3;
  const int combo = rest % NCOMBO;
  int grp = (rest / NCOMBO) * 8 + xcd;      // the first group of this workgroup; further ones gstep apart (reduced grids)
  const int cic = combo / NCOC, coc = combo % NCOC;
  const int total = jt.start[kWgMaxJobs];
  const int pair = wave % PW, ks = wave / PW;
  // M16 (64 output channels per workgroup): v_mfma_f32_16x16x32_f16 -- K = 32 pixels (the wave's two rows) per step, the 32 x 32
  // block of a tap as four 16 x 16 tiles (see the strip loop).  Its tiles are SWIZZLED in LDS: on odd tile rows the two 32-byte
  // halves of a pixel's 64-byte plane record are swapped (SWZ), see `in_m` below.
#ifndef UGN_WG_POOLED16
#define UGN_WG_POOLED16 1
#endif
#ifndef UGN_WG_SWZ
#define UGN_WG_SWZ 1
#endif
#ifndef UGN_WG_PER_TAP
#define UGN_WG_PER_TAP 1
#endif
#ifndef UGN_WG_PIPE
#define UGN_WG_PIPE 1         /* a tap's transposed reads pinned one tap ahead of its MFMAs (round-4 experiment 3: equal in isolation,
                                 40-110 us better over the five launches inside the step, same box, both orders) */
#endif
  constexpr bool M16 = PW == 2 && (!POOLED || UGN_WG_POOLED16) && !SP;
  constexpr bool SWZ = M16 && UGN_WG_SWZ;
  // transposed-read role of the lane: 16-lane group (channel half gh, k half h), row q of the 4-pixel block, columns 4p..
  const int h = lane >> 5, gh = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const int lane_off = (8 * h + q) * 64 + (16 * gh + 4 * p) * 2;

  auto job_of = [&](int s) {
    int jb = 0;
#pragma unroll
    for (int j = 1; j < kWgMaxJobs; ++j) jb += s >= jt.start[j] ? 1 : 0;
    return jb;
  };
  // ---- LDS-DMA pieces of this wave (piece pi = wave + 8 j of a strip's tiles).  What depends only on the lane is computed
  // once: the byte offset of the lane's 16-byte slot relative to the strip's first pixel and, for the input halo, the slot's
  // (row, column) for the image-border test -- packed off << 12 | row << 5 | column (pad slots: row 127, never inside an
  // image); for the pooled staging tile bit 0 says "argmax bytes" (a base of their own).  Per strip a piece then costs a dozen
  // instructions, and the pieces are issued BETWEEN the taps of the MFMA loop (they were a phase of 1,200-3,700 cycles per
  // strip in which no wave multiplied; in-kernel stamps, tools/stamp_wgrad.py).
  constexpr int NPIECE = IN_PIECES + (POOLED ? G::PZ_PIECES : G::DZ_BYTES / 1024);
  constexpr int NJ = (NPIECE + 7) / 8;
  int pk[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int pi = wave + 8 * j;
    pk[j] = 127 << 5;
    if (pi < IN_PIECES) {
      const int sg = pi * 64 + lane;                         // slot: [plane][pixel 0..IN_PIX - 1][quarter]
      const int plane = sg >= IN_PIX * 4 ? 1 : 0, rem = sg - IN_PIX * 4 * plane;
      const int pix = rem >> 2, c4 = rem & 3;
      const int row = (pix * 3641) >> 16, px = pix - row * 18;         // pix / 18 for pix < 400
      // (the quarter this LDS slot holds; SP: the 32-byte halves of a plane record are swapped for tile columns 8-15, see the strip loop)
      const int c4s = SP ? c4 ^ (((px >> 3) & 1) << 1) : (SWZ ? c4 ^ ((row & 1) << 1) : c4);
      const int off = ((row - 1) * HW + (px - 1)) * (CI * 4) + plane * CI * 2 + c4s * 16;
      if (sg < G::IN_SLOTS) pk[j] = (int)(((unsigned)off << 12) | (unsigned)(row << 5) | (unsigned)px);
    } else if (pi < NPIECE) {
      const int pd = pi - IN_PIECES;
      const int sg = pd * 64 + lane;
      if constexpr (!POOLED) {                               // slot: [plane][block][pixel 0..SR * 16 - 1][quarter]
        const int plane = sg / (PW * SR * 64), rem = sg - plane * (PW * SR * 64);
        const int nb = rem / (SR * 64), rem2 = rem - nb * (SR * 64);
        const int pix = rem2 >> 2, c4 = rem2 & 3;
        const int c4s = SWZ ? c4 ^ (((pix >> 4) & 1) << 1) : c4;
        pk[j] = ((pix >> 4) * HW + (pix & 15)) * (CO * 4) + plane * CO * 2 + (coc * COW + nb * 32) * 2 + c4s * 16;
      } else {                     // slots: [plane][block][pooled pixel][quarter], then [pooled pixel][COW / 16] of argmax bytes
        constexpr int HP = HW / 2, PZ_PIX = G::PZ_PIX;
        if (sg < 2 * PW * PZ_PIX * 4) {
          const int plane = sg / (PW * PZ_PIX * 4), rem = sg - plane * (PW * PZ_PIX * 4);
          const int nb = rem / (PZ_PIX * 4), rem2 = rem - nb * (PZ_PIX * 4);
          const int pp = rem2 >> 2, c4 = rem2 & 3;
          pk[j] = (((pp >> 3) * HP + (pp & 7)) * (CO * 4) + plane * CO * 2 + (coc * COW + nb * 32) * 2 + c4 * 16) << 1;
        } else {
          const int si = sg - 2 * PW * PZ_PIX * 4;
          const int pp = si / (COW / 16), part = si - pp * (COW / 16);
          pk[j] = pp < PZ_PIX ? ((((pp >> 3) * HP + (pp & 7)) * CO + coc * COW + part * 16) << 1) | 1 : -2;     // (-2: pad slot)
        }
      }
    }
  }
  // per-strip bases (wave-uniform)
  struct StripSrc { const char* in; const char* dz; const char* ix; int sy0, sx0; };
  auto strip_src = [&](int s) {
    const int jb = job_of(s), ls = s - jt.start[jb];
    const int img = ls / SPI, r = ls % SPI;
    StripSrc S;
    S.sy0 = (r / SPX) * SR;
    S.sx0 = (r % SPX) * 16;
    S.in = reinterpret_cast<const char*>(jt.job[jb].in) + (size_t)img * HW * HW * CI * 4 + cic * 64 +
           (size_t)(S.sy0 * HW + S.sx0) * (CI * 4);
    if constexpr (POOLED) {
      constexpr int HP = HW / 2;
      const size_t o = (size_t)img * HP * HP + (size_t)((S.sy0 / 2) * HP + S.sx0 / 2);
      S.dz = reinterpret_cast<const char*>(jt.job[jb].dz) + o * (CO * 4);
      S.ix = reinterpret_cast<const char*>(jt.job[jb].dz_idx) + o * CO;
    } else {
      S.dz = reinterpret_cast<const char*>(jt.job[jb].dz) + ((size_t)img * HW * HW + (size_t)(S.sy0 * HW + S.sx0)) * (CO * 4);
      S.ix = nullptr;
    }
    return S;
  };
  // piece j of this wave: input halo and gradient tile (pooled: pooled gradient + argmax bytes) of strip `Sin` -> buffer set b
  auto issue = [&](int j, const StripSrc& Sin, int b) {
    const int pi = wave + 8 * j;
    if (pi < IN_PIECES) {
      const int v = pk[j];
      const int gy = Sin.sy0 - 1 + ((v >> 5) & 127), gx = Sin.sx0 - 1 + (v & 31);
      const bool ok = (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
      const void* src = ok ? (const void*)(Sin.in + (ptrdiff_t)(v >> 12)) : zeros;
      dma16(src, sbase + (unsigned)(b * SET) + (unsigned)pi * 1024u);
    } else if (pi < NPIECE) {
      const int pd = pi - IN_PIECES;
      if constexpr (!POOLED) {
        dma16(Sin.dz + pk[j], sbase + (unsigned)(b * SET + IN_BYTES) + (unsigned)pd * 1024u);
      } else {
        const int v = pk[j];
        const char* base = (v & 1) ? Sin.ix : Sin.dz;
        const void* src = v == -2 ? zeros : (const void*)(base + (v >> 1));
        dma16(src, sbase + (unsigned)(b * SET + IN_BYTES) + (unsigned)pd * 1024u);
      }
    }
  };
  // MaxPool backward is part of the fragment build (pooled layers).  The B fragment of a k-step -- pixel row y of the strip, lane
  // (channel c, k half h) = pixels 8h .. 8h+7 -- holds, per pixel, the pooled gradient of its window where the window's argmax
  // byte names the pixel's position, else zero: ONE transposed read per plane (pooled pixels 4h .. 4h+3 of pooled row y / 2), the
  // four argmax bytes, and a dozen selects.  (The first version scattered the staged tile LDS -> LDS into a full-resolution
  // gradient tile: 950-1,450 cycles per strip and a second barrier, tools/stamp_wgrad.py.)
  const int lane_off_p = (4 * h + q) * 64 + (16 * gh + 4 * p) * 2;
  auto pooled_frag = [&](int b, int y, h8& bh, h8& bl) {
    const LDS_PTR(char) pv = lds + b * SET + IN_BYTES + pair * G::PZ_BLOCK + (y >> 1) * 512 + lane_off_p;
    const s4 ph = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))pv);
    const s4 pl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(pv + G::PZ_PLANE));
    const unsigned char* pi8 = reinterpret_cast<const unsigned char*>(smem) + b * SET + IN_BYTES + G::PZ_VAL +
                               ((y >> 1) * 8 + 4 * h) * COW + pair * 32 + (lane & 31);
    const unsigned posa = 2u * (unsigned)(y & 1);
    const uint2 hv = __builtin_bit_cast(uint2, ph), lv = __builtin_bit_cast(uint2, pl);
    unsigned fh[4], fl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned ix = pi8[j * COW];
      const unsigned m = (ix == posa ? 0x0000ffffu : 0u) | (ix == posa + 1u ? 0xffff0000u : 0u);
      const unsigned h2w = j < 2 ? hv.x : hv.y, l2w = j < 2 ? lv.x : lv.y;
      const unsigned hs = (j & 1) ? (h2w >> 16) : (h2w & 0xffffu), ls = (j & 1) ? (l2w >> 16) : (l2w & 0xffffu);
      fh[j] = (hs | (hs << 16)) & m;
      fl[j] = (ls | (ls << 16)) & m;
    }
    bh = __builtin_bit_cast(h8, make_uint4(fh[0], fh[1], fh[2], fh[3]));
    bl = __builtin_bit_cast(h8, make_uint4(fl[0], fl[1], fl[2], fl[3]));
  };
  // the same for the 16x16x32 form: lane (channel 16 cot + (lane & 15), k group) = pixels 8 xh .. + 7 of strip row y.  (SWZ: the two
  // k groups of a pass are rows 2k, 2k + 1 of the same columns -- the same pooled pixels, identical addresses, a broadcast.)
  auto pooled_frag16 = [&](int b, int y, int xh, int cot, h8& bh, h8& bl) {
    const LDS_PTR(char) pv = lds + b * SET + IN_BYTES + pair * G::PZ_BLOCK + (y >> 1) * 512 + (4 * xh + q) * 64 + (16 * cot + 4 * p) * 2;
    const s4 ph = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))pv);
    const s4 pl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(pv + G::PZ_PLANE));
    const unsigned char* pi8 = reinterpret_cast<const unsigned char*>(smem) + b * SET + IN_BYTES + G::PZ_VAL +
                               ((y >> 1) * 8 + 4 * xh) * COW + pair * 32 + 16 * cot + (lane & 15);
    const unsigned posa = 2u * (unsigned)(y & 1);
    const uint2 hv = __builtin_bit_cast(uint2, ph), lv = __builtin_bit_cast(uint2, pl);
    unsigned fh[4], fl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned ix = pi8[j * COW];
      const unsigned m = (ix == posa ? 0x0000ffffu : 0u) | (ix == posa + 1u ? 0xffff0000u : 0u);
      const unsigned h2w = j < 2 ? hv.x : hv.y, l2w = j < 2 ? lv.x : lv.y;
      const unsigned hs = (j & 1) ? (h2w >> 16) : (h2w & 0xffffu), ls = (j & 1) ? (l2w >> 16) : (l2w & 0xffffu);
      fh[j] = (hs | (hs << 16)) & m;
      fl[j] = (ls | (ls << 16)) & m;
    }
    bh = __builtin_bit_cast(h8, make_uint4(fh[0], fh[1], fh[2], fh[3]));
    bl = __builtin_bit_cast(h8, make_uint4(fl[0], fl[1], fl[2], fl[3]));
  };

  // M16 (64 output channels per workgroup, un-pooled gradient): v_mfma_f32_16x16x32_f16 -- K = 32 pixels (the wave's two rows) per
  // step, the 32 x 32 block of a tap as four 16 x 16 tiles.  Same operand reads, same FLOPs and accumulator registers as two
  // 32x32x16 steps, but the chip holds a higher clock on this shape: 6-12 % on these launches (profiles/r03_stage_stamps.txt).
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // SP: [tap][co tile of the wave's 32-channel block]
  f32x16 acc[(M16 || SP) ? 1 : 9];
  constexpr int SPT = SP ? 2 * PW : 4;        // SP: tiles of a wave per tap, [co tile * PW + ci tile of the wave]
  f32x4 a4[(M16 || SP) ? 9 : 1][SPT];         // [tap][ci tile * 2 + co tile]
#pragma unroll
  for (int t = 0; t < ((M16 || SP) ? 1 : 9); ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
  for (int t = 0; t < ((M16 || SP) ? 9 : 1); ++t)
#pragma unroll
    for (int k = 0; k < SPT; ++k) a4[t][k] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef UGN_WG_STAMP
  unsigned long long* stamp = g_wg_stamp ? g_wg_stamp + ((size_t)blockIdx.x * 8 + wave) * kWgStampPerWave : nullptr;
  int nstamp = 0;
  if (stamp && lane == 0) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[1] = __builtin_amdgcn_s_memrealtime(); }
#define WG_STAMP(k_) do { if (stamp && lane == 0 && nstamp < 60) stamp[4 + 6 * nstamp + (k_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG_STAMP(k_) do { } while (0)
#endif
  // A launch always has jt.ngroups groups per block combination -- the shares, the slabs and the order of every sum are fixed by
  // the job sizes alone -- and a workgroup takes the groups grp, grp + gstep, ...: on the full grid (gstep = ngroups) exactly one,
  // on a reduced grid (ugn_set_persistent_wgs: CUs left to RCCL) several in turn.  Results are bit-identical on every grid.
  for (bool first_group = true; grp < jt.ngroups; grp += jt.gstep, first_group = false) {
  const int s0 = (int)((long long)grp * total / jt.ngroups), s1 = (int)((long long)(grp + 1) * total / jt.ngroups);
  if (s0 >= s1) continue;
  if (!first_group) __syncthreads();        // the previous group's slab combine has finished reading its scratch
  // ---- prologue: tiles of strip s0 -> set 0 (three sets: and of strip s0 + 1 -> set 1)
  constexpr int NSET = wg_nsets<CO, SP>(), AHEAD = NSET - 1;
  {
    const StripSrc S0 = strip_src(s0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) issue(j, S0, 0);
    if (NSET == 3 && s0 + 1 < s1) {
      const StripSrc S1 = strip_src(s0 + 1);
#pragma unroll
      for (int j = 0; j < NJ; ++j) issue(j, S1, 1);
    }
  }
  int b = 0;
  int jb = job_of(s0);
  // pieces this wave issues per strip (LDS-DMA retires in order: a counted wait that leaves the newest strip's pieces in flight)
  const bool seven = wave + 8 * (NJ - 1) < NPIECE;
  bool newer_in_flight = NSET == 3 && s0 + 1 < s1;
  for (int s = s0; s < s1; ++s) {
    WG_STAMP(0);
    if (NSET == 3 && newer_in_flight) {
      static_assert(NSET == 2 || NJ == 7, "vmcnt immediates");
      if (seven) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    WG_STAMP(1);
    __syncthreads();      // strip s is in set b; nobody reads the set the next pieces go to any more
    WG_STAMP(2);
    // the tiles of strip s + AHEAD -> the set that strip s - 1 left, issued between the taps below
    const bool have_in = s + AHEAD < s1;
    newer_in_flight = NSET == 3 && s + 1 < s1 && have_in;      // (at the top of strip s + 1: were pieces issued behind its own?)
    const StripSrc Sin = strip_src(have_in ? s + AHEAD : s);
    const int bn = NSET == 3 ? (b + 2 >= 3 ? b - 1 : b + 2) : (b ^ 1);
    WG_STAMP(3);
    WG_STAMP(4);
    const LDS_PTR(char) in_b = lds + b * SET + ks * RPW * (18 * 64) + lane_off;
    const LDS_PTR(char) dz_b = lds + b * SET + IN_BYTES + pair * G::DZ_BLOCK + ks * RPW * (16 * 64) + lane_off;
    if constexpr (SP) {
      // ---- pooled layers on the SPARSE matrix pipe.  The un-pooled gradient has ONE non-zero per 2x2 window, so along a pixel row
      // every pair (2j, 2j + 1) holds at most one: 2:4 structured along K = pixels.  v_smfmac_f32_16x16x64_f16 takes the sparse
      // operand compressed -- per lane (m, k block kb) the 8 kept values of its 16 k and a 2-bit position for each -- and issues at
      // the rate of the dense 16x16x32 (tools/experiments/smfmac_rate.hip: 2.81 against 2.67 ms): HALF the matrix time.  Operand
      // layout (tools/experiments/smfmac_probe.hip): A lane (m = lane & 15, kb = lane >> 4): slots s = 0..7 = dense k
      // 16 kb + 4 (s >> 1) + position; B lane (n = lane & 15, kb): elements 0-7 = k 8 kb .., 8-15 = k 32 + 8 kb ..
      // So: M = output channels (dz^T, sparse), N = input channels (x, dense), K 64 = 4 strip rows x 16 pixels, k = 16 row + pixel.
      //   * the compressed operand IS the pooled tensor: slot s of row y = the pooled gradient of window (y >> 1, s) where the
      //     window's argmax sits in row y & 1 (else 0), position 2 (s & 1) + (argmax & 1): no un-pooling at all, built once per strip;
      //   * the dense operand of a tap = two of the 16x16x32 fragments (rows kb >> 1 and 2 + (kb >> 1), pixels 8 (kb & 1) .. + 7).
      // wave = (row group rg: strip rows 4 rg .. 4 rg + 3 = the K 64 of one v_smfmac, unit).  32 output channels: unit = input-channel
      // tile (2 x 1 tiles per tap, 72 accumulator registers); 64: unit = 32-channel output block, both input-channel tiles (2 x 2
      // tiles, 144 registers -- the form with 72, two row groups of 8 rows, builds the sparse operands twice as often and measured
      // 262 -> 279 us).
      const int rg = wave & 3, unit = wave >> 2;
      const int blk = PW == 2 ? unit : 0;
      const int kb = lane >> 4, mi = lane & 15;
      h8 sah[2], sal[2];
      int sidx[2];
      {
        const int y = 4 * rg + kb;
#pragma unroll
        for (int cot = 0; cot < 2; ++cot) {
          const LDS_PTR(char) pv = lds + b * SET + IN_BYTES + blk * G::PZ_BLOCK + (y >> 1) * 512 + q * 64 + (16 * cot + 4 * p) * 2;
          const uint2 h0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))pv));
          const uint2 h1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(pv + 4 * 64)));
          const uint2 l0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(pv + G::PZ_PLANE)));
          const uint2 l1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(pv + G::PZ_PLANE + 4 * 64)));
          const unsigned char* pi8 = reinterpret_cast<const unsigned char*>(smem) + b * SET + IN_BYTES + G::PZ_VAL +
                                     ((y >> 1) * 8) * COW + blk * 32 + 16 * cot + mi;
          const unsigned hv[4] = {h0.x, h0.y, h1.x, h1.y}, lv[4] = {l0.x, l0.y, l1.x, l1.y};
          unsigned hm[4], lm[4], iw = 0;
          const unsigned rowpar = (unsigned)(y & 1);
#pragma unroll
          for (int d = 0; d < 4; ++d) {            // dword d = slots 2d, 2d + 1 = pooled columns 2d, 2d + 1
            const unsigned i0 = pi8[(2 * d) * COW], i1 = pi8[(2 * d + 1) * COW];
            const unsigned m = ((i0 >> 1) == rowpar ? 0x0000ffffu : 0u) | ((i1 >> 1) == rowpar ? 0xffff0000u : 0u);
            hm[d] = hv[d] & m;
            lm[d] = lv[d] & m;
            iw |= ((i0 & 1u) | ((2u | (i1 & 1u)) << 2)) << (4 * d);
          }
          sah[cot] = __builtin_bit_cast(h8, make_uint4(hm[0], hm[1], hm[2], hm[3]));
          sal[cot] = __builtin_bit_cast(h8, make_uint4(lm[0], lm[1], lm[2], lm[3]));
          sidx[cot] = (int)iw;
        }
      }
      // Dense fragments: lanes 0-31 of a transposed read (k blocks 0, 1) are pixels 0-7 | 8-15 of ONE row, 512 B apart = the same
      // banks if both read the same 32-byte half of their records.  The halo tile therefore holds the two halves of a plane record
      // SWAPPED for tile columns 8-15 (16, 17: not): the k blocks of a pass then sit in different halves for every tap -- a lane's
      // half is cit ^ ((column >> 3) & 1), column = 8 (kb & 1) + q + dx (+ 4 for the second read of a pair), which leaves the lane's
      // 8-column block only in that second read and only for q + dx >= 4: three lane bases.
      const int hsel = kb & 1;            // the half that holds input-channel tile 0 at this lane's first column
      const LDS_PTR(char) in_r = lds + b * SET + ((4 * rg + (kb >> 1)) * 18 + 8 * (kb & 1) + q) * 64 + 4 * p * 2;
      // [dx]: byte offset of tile 0's half for the SECOND read of a pair (the first: 32 hsel)
      const int sec0[3] = {32 * hsel, 32 * (hsel ^ (q + 1 >= 4 ? 1 : 0)), 32 * (hsel ^ (q + 2 >= 4 ? 1 : 0))};
      // the dense fragments of micro-step u + SPD are read before the 6 MFMAs of micro-step u = (tap, input-channel tile) (a ring of
      // SPD + 1 register sets)
      constexpr int NCW = PW == 2 ? 2 : 1;        // input-channel tiles of a wave
      constexpr int SPD = PW == 2 ? UGN_WG_SPD64 : UGN_WG_SPD, NS = SPD + 1, NU = 9 * NCW;
      h16 fbh[NS], fbl[NS];
      auto tr2 = [&](const LDS_PTR(char) first, const LDS_PTR(char) second, int off) {
        const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(first + off));
        const s4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(second + off + 4 * 64));
        const s8 v = __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(h8, v);
      };
      auto load_b = [&](int set, int u) {
        const int t = u / NCW, cit = PW == 2 ? u % NCW : unit;
        const int o = ((t / 3) * 18 + (t % 3)) * 64;
        // tile cit's half = tile 0's half ^ cit
        const LDS_PTR(char) fst = in_r + ((32 * hsel) ^ (32 * cit));
        const LDS_PTR(char) sec = in_r + (sec0[t % 3] ^ (32 * cit));
        const h8 a0 = tr2(fst, sec, o), a1 = tr2(fst, sec, o + 2 * 18 * 64);
        const h8 c0 = tr2(fst, sec, IN_PLANE + o), c1 = tr2(fst, sec, IN_PLANE + o + 2 * 18 * 64);
        fbh[set] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        fbl[set] = __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      };
#pragma unroll
      for (int u = 0; u < SPD; ++u) load_b(u % NS, u);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int t = u / NCW, cw = u % NCW;
#if UGN_SP_ABL & 2
        if (u + SPD < NU && u + SPD < 2) load_b((u + SPD) % NS, u + SPD);
#else
        if (u + SPD < NU) load_b((u + SPD) % NS, u + SPD);
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cot = 0; cot < 2; ++cot) {
          f32x4 c = a4[t][cot * NCW + cw];
#if UGN_SP_ABL & 1
          c[0] += (float)fbh[u % NS][0] * (float)sah[cot][0] + (float)fbl[u % NS][1] * (float)sal[cot][1] + (float)sidx[cot];
#else
          c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(sah[cot], fbh[u % NS], c, sidx[cot], 0, 0);
          c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(sah[cot], fbl[u % NS], c, sidx[cot], 0, 0);
          c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(sal[cot], fbh[u % NS], c, sidx[cot], 0, 0);
#endif
          a4[t][cot * NCW + cw] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (cw == NCW - 1 && t < NJ && have_in && !(UGN_SP_ABL & 4)) {      // one LDS-DMA piece of the next strip per tap
          issue(t, Sin, bn);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if constexpr (M16) {
      // lane (i = lane & 15, k group kg = lane >> 4): 8 consecutive pixels of the wave's 32.  SWZ (default): row kr = kg & 1, columns
      // 8 (kg >> 1) .. + 7 -- a ds_read_b64_tr_b16 is serviced as lanes 0-31, then 32-63 (MI355X_MICROARCH.md, LDS), i.e. k groups
      // {0, 1} together.  A 16-lane group reads 32 of the 64 bytes of four consecutive pixels (banks 16 q + 8 half + 2 p), so two groups
      // of a pass must sit in DIFFERENT halves: with the two k groups on rows r, r + 1 and the halves swapped on odd tile rows (the
      // DMA fetches quarter c4 ^ 2 (row & 1) into slot c4) they do, for every tap, at lane base + immediate: the half a lane reads
      // is cit ^ ((kr + dy) & 1), i.e. one of two lane bases.  (Round 3 had k groups {0, 1} = columns 0-7 | 8-15 of one row: 512 B
      // apart, the same banks -- SQ_LDS_BANK_CONFLICT 0.49 per active LDS cycle on all four M16 launches, profiles/r03_lds_conflicts.csv.)
      const int kg = lane >> 4;
      const int kr = SWZ ? (kg & 1) : (kg >> 1), kx = SWZ ? (kg >> 1) : (kg & 1);
      const LDS_PTR(char) in_m0 = lds + b * SET + ks * RPW * (18 * 64) + (kr * 18 + 8 * kx + q) * 64 + 4 * p * 2;
      const LDS_PTR(char) dz_m0 = lds + b * SET + IN_BYTES + pair * G::DZ_BLOCK + ks * RPW * (16 * 64) + (kr * 16 + 8 * kx + q) * 64 + 4 * p * 2;
      // half c of the pixel record as this lane finds it on an even / odd tap row: [c ^ kr] (SWZ) or [c]
      const LDS_PTR(char) in_m[2] = {in_m0 + (SWZ ? 32 * kr : 0), in_m0 + (SWZ ? 32 * (kr ^ 1) : 32)};
      const LDS_PTR(char) dz_m[2] = {dz_m0 + (SWZ ? 32 * kr : 0), dz_m0 + (SWZ ? 32 * (kr ^ 1) : 32)};
      h8 bh[2], bl[2];
#pragma unroll
      for (int cot = 0; cot < 2; ++cot) {
        if constexpr (POOLED) {
          pooled_frag16(b, ks * RPW + kr, kx, cot, bh[cot], bl[cot]);
        } else {
          bh[cot] = tr_pair(dz_m[cot], 0, 4 * 64);
          bl[cot] = tr_pair(dz_m[cot], G::DZ_PLANE, G::DZ_PLANE + 4 * 64);
        }
      }
      // software pipeline over the taps: the 8 transposed reads of tap t + 1 (both input-channel tiles, both planes) go out before
      // the 12 MFMAs of tap t, pinned with sched_barrier -- hipcc on its own reads a fragment one or two instructions ahead of the
      // MFMA that needs it and the matrix pipe waits for LDS (see conv_mm16_kernel)
      h8 fah[2][2], fal[2][2];           // [register set][input-channel tile]
      auto load_a = [&](int set, int t) {
        const int o = ((t / 3) * 18 + (t % 3)) * 64;
#pragma unroll
        for (int cit = 0; cit < 2; ++cit) {
          const LDS_PTR(char) im = in_m[SWZ ? cit ^ ((t / 3) & 1) : cit];
          fah[set][cit] = tr_pair(im, o, o + 4 * 64);
          fal[set][cit] = tr_pair(im, IN_PLANE + o, IN_PLANE + o + 4 * 64);
        }
      };
      if (UGN_WG_PIPE) load_a(0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (UGN_WG_PIPE) {
          if (t + 1 < 9) load_a((t + 1) & 1, t + 1);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          load_a(t & 1, t);
        }
#pragma unroll
        for (int cit = 0; cit < 2; ++cit) {
          const h8 ah = fah[t & 1][cit], al = fal[t & 1][cit];
#pragma unroll
          for (int cot = 0; cot < 2; ++cot) {
            f32x4 c = a4[t][cit * 2 + cot];
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[cot], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[cot], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[cot], c, 0, 0, 0);
            a4[t][cit * 2 + cot] = c;
          }
        }
        if (UGN_WG_PIPE) __builtin_amdgcn_sched_barrier(0);
        // the next strip's pieces go out in the FIRST taps (UGN_WG_PER_TAP per tap): the later the last one is issued, the more of
        // its latency the top-of-strip wait sees
        if (t * UGN_WG_PER_TAP < NJ && have_in) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < UGN_WG_PER_TAP; ++u)
            if (t * UGN_WG_PER_TAP + u < NJ) issue(t * UGN_WG_PER_TAP + u, Sin, bn);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      // k-step = pixel row rr of the wave: lane half h covers pixels 8h .. 8h+7 (two 4-pixel blocks)
      h8 bh, bl;
      if constexpr (POOLED) {
        pooled_frag(b, ks * RPW + rr, bh, bl);
      } else {
        bh = tr_pair(dz_b, (rr * 16) * 64, (rr * 16 + 4) * 64);
        bl = tr_pair(dz_b, G::DZ_PLANE + (rr * 16) * 64, G::DZ_PLANE + (rr * 16 + 4) * 64);
      }
      h8 fah[2], fal[2];                 // the input fragments of a tap, read one tap ahead (as in the M16 loop above)
      auto load_a = [&](int set, int t) {
        const int o = ((rr + t / 3) * 18 + (t % 3)) * 64;
        fah[set] = tr_pair(in_b, o, o + 4 * 64);
        fal[set] = tr_pair(in_b, IN_PLANE + o, IN_PLANE + o + 4 * 64);
      };
      if (UGN_WG_PIPE) load_a(0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (UGN_WG_PIPE) {
          if (t + 1 < 9) load_a((t + 1) & 1, t + 1);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          load_a(t & 1, t);
        }
        acc[t] = mfma_h8(fah[t & 1], bh, acc[t]);
        acc[t] = mfma_h8(fah[t & 1], bl, acc[t]);
        acc[t] = mfma_h8(fal[t & 1], bh, acc[t]);
        if (UGN_WG_PIPE) __builtin_amdgcn_sched_barrier(0);
        // one LDS-DMA piece after every tap (every second tap where there are two k-steps) until the wave's pieces are out
        constexpr int EVERY = RPW == 1 ? 1 : 2;
        const int slot = rr * 9 + t;
        if (slot % EVERY == 0 && slot / EVERY < NJ && have_in) {
          __builtin_amdgcn_sched_barrier(0);
          issue(slot / EVERY, Sin, bn);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    WG_STAMP(5);
#ifdef UGN_WG_STAMP
    ++nstamp;
#endif
    const int jn = s + 1 < s1 ? job_of(s + 1) : -1;
    if (jn != jb) {
      // ---- job (or share) finished: add the K-split waves of a pair through LDS in a fixed order, write the slab.  The
      // scratch is the buffer set just multiplied (the next strip streams into the other one).
      float* slab = jt.job[jb].slab + ((size_t)combo * jt.job[jb].ng + (grp - jt.job[jb].g0)) * (9 * 32 * COW);
      float* scr = reinterpret_cast<float*>(smem + (SET < 32768 ? 2 * SET : b * SET));
#pragma unroll 1
      for (int t = 0; t < 9; ++t) {
        __syncthreads();
        f32x16 a;
        if constexpr (SP && PW == 1) {       // two tiles per wave
          f32x4 q0 = a4[0][0], q1 = a4[0][1];
#pragma unroll
          for (int u = 1; u < 9; ++u) if (t == u) { q0 = a4[u][0]; q1 = a4[u][1]; }
#pragma unroll
          for (int i = 0; i < 4; ++i) { a[i] = q0[i]; a[4 + i] = q1[i]; a[8 + i] = 0.f; a[12 + i] = 0.f; }
        } else if constexpr (M16 || SP) {
          f32x4 q0 = a4[0][0], q1 = a4[0][1], q2 = a4[0][2], q3 = a4[0][3];
#pragma unroll
          for (int u = 1; u < 9; ++u) if (t == u) { q0 = a4[u][0]; q1 = a4[u][1]; q2 = a4[u][2]; q3 = a4[u][3]; }
#pragma unroll
          for (int i = 0; i < 4; ++i) { a[i] = q0[i]; a[4 + i] = q1[i]; a[8 + i] = q2[i]; a[12 + i] = q3[i]; }
        } else {
          a = acc[0];
#pragma unroll
          for (int u = 1; u < 9; ++u) if (t == u) a = acc[u];
        }
#pragma unroll
        for (int i = 0; i < (SP ? 4 * SPT : 16); ++i) scr[wave * 1024 + i * 64 + lane] = a[i];
        __syncthreads();
        if constexpr (SP) {
          // the tiles of unit u from its four row-group waves (wave = 4 u + rg), in order.  Tile register r of lane l = output channel
          // 4 (l >> 4) + r (M), input channel l & 15 (N) of the tile; tile = co tile * PW + ci tile of the wave
#pragma unroll
          for (int k = 0; k < 2 * PW; ++k) {
            const int e = tid + 512 * k, u = e / (SPT * 256), rem = e - u * (SPT * 256);
            const int tile = rem >> 8, r = (rem >> 6) & 3, ln = rem & 63;
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) sum += scr[(u * 4 + g) * 1024 + (tile * 4 + r) * 64 + ln];
            const int cot = tile / PW, cw = tile % PW;
            const int co = (PW == 2 ? 32 * u : 0) + 16 * cot + 4 * (ln >> 4) + r;
            const int ci = 16 * (PW == 2 ? cw : u) + (ln & 15);
            slab[(t * 32 + ci) * COW + co] = sum;
          }
        } else
#pragma unroll
        for (int k = 0; k < 2 * PW; ++k) {
          const int e = tid + 512 * k, pr = e >> 10, idx = e & 1023;
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < KS; ++w) sum += scr[(w * PW + pr) * 1024 + idx];
          const int reg = idx >> 6, ln = idx & 63;
          // 32x32 block: register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31; four 16x16 tiles (M16): register
          // 4 tile + i of lane l = row 16 (tile >> 1) + 4 (l >> 4) + i, column 16 (tile & 1) + (l & 15)
          const int ci = M16 ? 16 * (reg >> 3) + 4 * (ln >> 4) + (reg & 3) : (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5);
          const int co = pr * 32 + (M16 ? 16 * ((reg >> 2) & 1) + (ln & 15) : (ln & 31));
          slab[(t * 32 + ci) * COW + co] = sum;
        }
      }
#pragma unroll
      for (int t = 0; t < ((M16 || SP) ? 1 : 9); ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
      for (int t = 0; t < ((M16 || SP) ? 9 : 1); ++t)
#pragma unroll
        for (int k = 0; k < SPT; ++k) a4[t][k] = f32x4{0.f, 0.f, 0.f, 0.f};
      jb = jn;
    }
    b = NSET == 3 ? (b == 2 ? 0 : b + 1) : (b ^ 1);
  }
  }
#ifdef UGN_WG_STAMP
  if (stamp && lane == 0) { stamp[2] = __builtin_amdgcn_s_memtime(); stamp[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

struct WgFinish {
  const float* slab[kWgMaxJobs];
  float* dw[kWgMaxJobs];
  const H2Meta* in_meta[kWgMaxJobs];
  const H2Meta* dz_meta[kWgMaxJobs];
  int ng[kWgMaxJobs];
};
// dW[tap][ci][co] (HWIO) = 2^-(e_in + e_dz) * sum over the job's groups, in order.  One thread per element; blockIdx.y = job.
__global__ __launch_bounds__(256) void wgrad_mm_finish(const WgFinish ft, int CI, int CO, int COW) {
  const int j = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 9 * CI * CO) return;
  const int co = e % CO, ci = (e / CO) % CI, tap = e / (CO * CI);
  const int ncoc = CO / COW, combo = (ci >> 5) * ncoc + co / COW;
  const int ng = ft.ng[j];
  const float* sl = ft.slab[j] + (size_t)combo * ng * (9 * 32 * COW) + (tap * 32 + (ci & 31)) * COW + (co % COW);
  float sum = 0.f;
  for (int g = 0; g < ng; ++g) sum += sl[(size_t)g * (9 * 32 * COW)];
  ft.dw[j][e] = ldexpf(sum, -(ft.in_meta[j]->e + ft.dz_meta[j]->e));
}

template <int CI, int CO>
constexpr int wg_ngroups() { return 256 / ((CI / 32) * (CO / WGeo<CO>::COW)); }

template <int CI, int CO, int HW, int POOLED, int SP = 0>
int launch_wgrad(const uint16_t* const* in, const void* const* in_meta, const uint16_t* const* dz, const uint8_t* const* dz_idx,
                 const void* const* dz_meta, float* const* dw, const int* n, int njobs, float* ws, size_t ws_floats, hipStream_t st) {
  constexpr int SRV = SP ? 16 : wg_default_sr(CO);
  using G = WGeo<CO, SRV>;
  constexpr int NCOMBO = (CI / 32) * (CO / G::COW), NG = wg_ngroups<CI, CO>();
  constexpr int SPI = (HW / G::SR) * (HW / 16);
  constexpr int LDS = wg_lds_bytes<CO, POOLED, SRV, wg_nsets<CO, SP>()>();
  static_assert(LDS <= 163840 && NG % 8 == 0, "geometry");
  auto kern = wgrad_mm_kernel<CI, CO, HW, POOLED, SP>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { ugn_set_error("wgrad_mm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const void* zeros = zero_block();
  if (!zeros) { ugn_set_error("wgrad_mm: cannot allocate the zero block"); return UGN_EINVAL; }
  WgJobs jt = {};
  WgFinish ft = {};
  int total = 0;
  for (int j = 0; j < kWgMaxJobs; ++j) {
    jt.start[j] = total;
    if (j < njobs) total += n[j] * SPI;
  }
  jt.start[kWgMaxJobs] = total;
  // (ugn_set_persistent_wgs: a grid of NGE < NG groups per block combination leaves CUs to RCCL; every workgroup then walks
  //  several of the NG shares -- same shares, same slabs, same sums)
  const int NGE = persistent_groups(NG, NCOMBO);
  jt.ngroups = NG;
  jt.gstep = NGE;
  const size_t slab_floats = (size_t)9 * 32 * G::COW;
  size_t used = 0;
  for (int j = 0; j < kWgMaxJobs; ++j) {
    const int jj = j < njobs ? j : njobs - 1;
    jt.job[j].in = in[jj]; jt.job[j].dz = dz[jj]; jt.job[j].dz_idx = POOLED ? dz_idx[jj] : nullptr;
    if (j >= njobs) { jt.job[j].slab = jt.job[jj].slab; jt.job[j].g0 = jt.job[jj].g0; jt.job[j].ng = jt.job[jj].ng; continue; }
    // groups whose share [g*T/NG, (g+1)*T/NG) meets the job's strips [a, b)
    const long long a = jt.start[j], bnd = (long long)jt.start[j] + (long long)n[j] * SPI;
    int g0 = (int)(a * NG / total);
    while ((long long)(g0 + 1) * total / NG <= a) ++g0;            // first group whose share ends after a
    while (g0 > 0 && (long long)g0 * total / NG > a) --g0;
    int g1 = g0;
    while (g1 + 1 < NG && (long long)(g1 + 1) * total / NG < bnd) ++g1;
    jt.job[j].g0 = g0; jt.job[j].ng = g1 - g0 + 1;
    jt.job[j].slab = ws + used;
    used += (size_t)NCOMBO * jt.job[j].ng * slab_floats;
    ft.slab[j] = jt.job[j].slab; ft.dw[j] = dw[j]; ft.ng[j] = jt.job[j].ng;
    ft.in_meta[j] = (const H2Meta*)in_meta[j]; ft.dz_meta[j] = (const H2Meta*)dz_meta[j];
  }
  if (used > ws_floats) { ugn_set_error("wgrad_mm: workspace too small (%zu floats needed, %zu given)", used, ws_floats); return UGN_EINVAL; }
  // a group with an EMPTY share writes nothing, but may lie between two groups of a job's slab list: such a slab must add 0
  // (only possible when there are fewer strips than groups)
  if (total < NG) {
    hipError_t me = hipMemsetAsync(ws, 0, used * sizeof(float), st);
    if (me != hipSuccess) { ugn_set_error("wgrad_mm: memset: %s", hipGetErrorString(me)); return (int)me; }
  }
  hipLaunchKernelGGL(kern, dim3(NGE * NCOMBO), dim3(512), LDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("wgrad_mm");
  hipLaunchKernelGGL(wgrad_mm_finish, dim3((9 * CI * CO + 255) / 256, njobs), dim3(256), 0, st, ft, CI, CO, G::COW);
  UGN_CHECK_LAUNCH("wgrad_mm finish");
  return 0;
}

template <int CI, int CO>
size_t ws_floats_for(int njobs) {
  using G = WGeo<CO>;
  constexpr int NCOMBO = (CI / 32) * (CO / G::COW), NG = wg_ngroups<CI, CO>();
  return (size_t)NCOMBO * (NG + njobs) * 9 * 32 * G::COW;
}

}  // namespace

#ifdef UGN_WG_STAMP
extern "C" int ugn_wg_debug_stamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_stamp), &p, sizeof(p));
}
#endif

extern "C" size_t ugn_mm_conv3x3_wgrad_ws(int hw, int cin, int cout) {
#define WS(CI_, CO_, HW_) if (cin == CI_ && cout == CO_ && hw == HW_) return ws_floats_for<CI_, CO_>(kWgMaxJobs) * sizeof(float);
  WS(32, 32, 64) WS(32, 64, 32) WS(64, 64, 32) WS(64, 128, 16) WS(128, 128, 16)
#undef WS
  return 0;
}

extern "C" int ugn_mm_conv3x3_wgrad_multi(const uint16_t* const* in, const void* const* in_meta, const uint16_t* const* dz,
                                          const uint8_t* const* dz_idx, const void* const* dz_meta, float* const* dw,
                                          const int* n, int njobs, int hw, int cin, int cout, void* ws, size_t ws_bytes,
                                          void* stream) {
  UGN_REQUIRE(in && in_meta && dz && dz_meta && dw && n && ws, "ugn_mm_conv3x3_wgrad_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kWgMaxJobs, "ugn_mm_conv3x3_wgrad_multi: njobs must be 1..%d (got %d)", kWgMaxJobs, njobs);
  const bool pooled = dz_idx != nullptr && dz_idx[0] != nullptr;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(in[j] && in_meta[j] && dz[j] && dz_meta[j] && dw[j] && n[j] > 0, "ugn_mm_conv3x3_wgrad_multi: null pointer or n <= 0 in job %d", j);
    UGN_REQUIRE(pooled == (dz_idx != nullptr && dz_idx[j] != nullptr), "ugn_mm_conv3x3_wgrad_multi: dz_idx for all jobs or none");
  }
  hipStream_t st = (hipStream_t)stream;
  float* wsf = (float*)ws;
  const size_t wfl = ws_bytes / sizeof(float);
#define WG(CI_, CO_, HW_, P_)                                        \
  if (cin == CI_ && cout == CO_ && hw == HW_ && pooled == (P_ != 0)) \
    return launch_wgrad<CI_, CO_, HW_, P_, (P_ && UGN_WG_SPARSE) ? 1 : 0>(in, in_meta, dz, dz_idx, dz_meta, dw, n, njobs, wsf, wfl, st);
  WG(32, 32, 64, 1) WG(32, 64, 32, 0) WG(64, 64, 32, 1) WG(64, 128, 16, 0) WG(128, 128, 16, 0)
#undef WG
  ugn_set_error("ugn_mm_conv3x3_wgrad_multi: unsupported shape cin=%d cout=%d hw=%d pooled=%d", cin, cout, hw, (int)pooled);
  return UGN_EINVAL;
}
