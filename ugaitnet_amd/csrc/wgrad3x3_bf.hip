// Weight gradient of the 3x3 layers on bf16 tensors (v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 result): the configs[4]
// counterpart of wgrad3x3_mm.hip -- read its header; same structure (K = pixels, both operands by ds_read_b64_tr_b16 from planar
// 64-byte-row tiles, strips of 8 x 16 pixels, K-split waves added through LDS, slabs, fixed-order finish), with ONE plane per
// tile and ONE MFMA per tap.  Replaces Conv2DBackpropFilter (+ MaxPoolGrad) of reference nets/mj_uwyhNets_ba.py:431-462.
#include "mm_common.h"

using namespace ugn_mm;

namespace {

typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
#define LDS_PTR(T) __attribute__((address_space(3))) T*

constexpr int kWgMaxJobs = 6;
// Rows of a strip.  With one bf16 MFMA per tap a strip's matrix work is a third of the f16x2 kernels' while its fixed costs (two
// barriers, the strip's base arithmetic, the tile wait) are the same: the POOLED layers -- whose gradient tile is a quarter of the
// un-pooled one, so the buffers stay small -- take 32-row strips (UGN_BF_SR_POOLED); the un-pooled 64-wide ones keep 8 (the 16x16x32
// loop multiplies exactly a wave's two rows).
#ifndef UGN_BF_SR_POOLED
#define UGN_BF_SR_POOLED 32
#endif
#ifndef UGN_BF_SR_M16
#define UGN_BF_SR_M16 8       /* un-pooled 64-wide layers (16x16x32 loop); 16 = two K = 32 steps per wave and strip: measured the same (1.998 | 1.997 ms) */
#endif

struct WgJob {
  const uint16_t* in;        // bf16 [n][hw][hw][ci]
  const uint16_t* dz;        // bf16 [n][hw][hw][co]   (pooled: [n][hw/2][hw/2][co])
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

template <int CO, int POOLED>
struct WGeo {
  static constexpr int COW = CO >= 64 ? 64 : 32;     // output channels of a workgroup
  static constexpr int PW = COW / 32;                // 32x32 block pairs
  static constexpr int KS = 8 / PW;                  // waves sharing a pair (K split)
  static constexpr int SR = POOLED ? UGN_BF_SR_POOLED : (CO >= 64 ? UGN_BF_SR_M16 : 8);      // pixel rows of a strip
  static constexpr int RPW = SR / KS;                // pixel rows of a strip per wave
  static constexpr int IN_PIX = (SR + 2) * 18;
  static constexpr int IN_SLOTS = IN_PIX * 4;        // 180 pixels x 4 slots = 720 slots -> 11.25 pieces (8 rows)
  static constexpr int IN_PIECES = (IN_SLOTS + 63) / 64;
  static constexpr int IN_BYTES = IN_PIECES * 1024;
  static constexpr int DZ_BLOCK = SR * 1024;         // [block][SR x 16 pixels][64 B]
  static constexpr int DZ_BYTES = PW * DZ_BLOCK;
  // pooled layers: the gradient tile is the POOLED one, [block][SR / 2 x 8 pooled pixels][64 B], + [pooled pixels][COW] argmax bytes
  static constexpr int PZ_PIX = SR * 4;
  static constexpr int PZ_BLOCK = PZ_PIX * 64;
  static constexpr int PZ_VAL = PW * PZ_BLOCK;
  static constexpr int PZ_PIECES = (PZ_VAL + PZ_PIX * COW) / 1024;      // values + argmax bytes (8 rows: 2 + 1 per block)
  static constexpr int PZ_BYTES = PZ_PIECES * 1024;
  static_assert(SR % KS == 0 && (PZ_VAL + PZ_PIX * COW) % 1024 == 0, "strip geometry");
};
// buffer set = input halo + gradient tile (pooled: the pooled gradient tile); the K-split combine needs 8 waves x 4 KB of scratch:
// the set just multiplied where a set is that large, 32 KB of its own behind the two sets otherwise
template <int CO, int POOLED>
constexpr int wg_set_bytes() { return WGeo<CO, POOLED>::IN_BYTES + (POOLED ? WGeo<CO, POOLED>::PZ_BYTES : WGeo<CO, POOLED>::DZ_BYTES); }
template <int CO, int POOLED>
constexpr int wg_lds_bytes() { return 2 * wg_set_bytes<CO, POOLED>() + (wg_set_bytes<CO, POOLED>() < 32768 ? 32768 : 0); }

__device__ __forceinline__ h8 tr_pair(const LDS_PTR(char) base, int off0, int off1) {
  const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(base + off0));
  const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(base + off1));
  const s8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(h8, v);
}
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_b8(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}

template <int CI, int CO, int HW, int POOLED>
__global__ __launch_bounds__(512, 2) void wgrad_bf_kernel(const WgJobs jt, const void* __restrict__ zeros) {
  using G = WGeo<CO, POOLED>;
  constexpr int COW = G::COW, PW = G::PW, KS = G::KS, RPW = G::RPW, SR = G::SR, SET = wg_set_bytes<CO, POOLED>();
  constexpr int IN_PIECES = G::IN_PIECES, IN_BYTES = G::IN_BYTES;
  constexpr int NCOC = CO / COW, NCOMBO = (CI / 32) * NCOC;
  constexpr int SPX = HW / 16, SPI = (HW / SR) * SPX;       // strips per image row / per image
  static_assert(HW % SR == 0, "strip geometry");
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
      const int sg = pi * 64 + lane;                         // slot: [pixel 0..IN_PIX - 1][quarter]
      const int pix = sg >> 2, c4 = sg & 3;
      const int row = pix / 18, px = pix - row * 18;
      const int off = ((row - 1) * HW + (px - 1)) * (CI * 2) + c4 * 16;
      if (sg < G::IN_SLOTS) pk[j] = (int)(((unsigned)off << 12) | (unsigned)(row << 5) | (unsigned)px);
    } else if (pi < NPIECE) {
      const int pd = pi - IN_PIECES;
      const int sg = pd * 64 + lane;
      if constexpr (!POOLED) {                               // slot: [block][pixel 0..SR * 16 - 1][quarter]
        const int nb = sg / (SR * 64), rem2 = sg - nb * (SR * 64);
        const int pix = rem2 >> 2, c4 = rem2 & 3;
        pk[j] = ((pix >> 4) * HW + (pix & 15)) * (CO * 2) + (coc * COW + nb * 32) * 2 + c4 * 16;
      } else {                     // slots: [block][pooled pixel][quarter], then [pooled pixel][COW / 16] of argmax bytes
        constexpr int HP = HW / 2, PZ_PIX = G::PZ_PIX;
        if (sg < PW * PZ_PIX * 4) {
          const int nb = sg / (PZ_PIX * 4), rem2 = sg - nb * (PZ_PIX * 4);
          const int pp = rem2 >> 2, c4 = rem2 & 3;
          pk[j] = (((pp >> 3) * HP + (pp & 7)) * (CO * 2) + (coc * COW + nb * 32) * 2 + c4 * 16) << 1;
        } else {
          const int si = sg - PW * PZ_PIX * 4;
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
    S.in = reinterpret_cast<const char*>(jt.job[jb].in) + (size_t)img * HW * HW * CI * 2 + cic * 64 +
           (size_t)(S.sy0 * HW + S.sx0) * (CI * 2);
    if constexpr (POOLED) {
      constexpr int HP = HW / 2;
      const size_t o = (size_t)img * HP * HP + (size_t)((S.sy0 / 2) * HP + S.sx0 / 2);
      S.dz = reinterpret_cast<const char*>(jt.job[jb].dz) + o * (CO * 2);
      S.ix = reinterpret_cast<const char*>(jt.job[jb].dz_idx) + o * CO;
    } else {
      S.dz = reinterpret_cast<const char*>(jt.job[jb].dz) + ((size_t)img * HW * HW + (size_t)(S.sy0 * HW + S.sx0)) * (CO * 2);
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
  // MaxPool backward is part of the fragment build (pooled layers): see wgrad3x3_mm.hip.  One transposed read (pooled pixels
  // 4h .. 4h+3 of pooled row y / 2), the four argmax bytes, and the selects.
  const int lane_off_p = (4 * h + q) * 64 + (16 * gh + 4 * p) * 2;
  auto pooled_frag = [&](int b, int y, h8& bh) {
    const LDS_PTR(char) pv = lds + b * SET + IN_BYTES + pair * G::PZ_BLOCK + (y >> 1) * 512 + lane_off_p;
    const s4 ph = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))pv);
    const unsigned char* pi8 = reinterpret_cast<const unsigned char*>(smem) + b * SET + IN_BYTES + G::PZ_VAL +
                               ((y >> 1) * 8 + 4 * h) * COW + pair * 32 + (lane & 31);
    const unsigned posa = 2u * (unsigned)(y & 1);
    const uint2 hv = __builtin_bit_cast(uint2, ph);
    unsigned fh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned ix = pi8[j * COW];
      const unsigned m = (ix == posa ? 0x0000ffffu : 0u) | (ix == posa + 1u ? 0xffff0000u : 0u);
      const unsigned h2w = j < 2 ? hv.x : hv.y;
      const unsigned hs = (j & 1) ? (h2w >> 16) : (h2w & 0xffffu);
      fh[j] = (hs | (hs << 16)) & m;
    }
    bh = __builtin_bit_cast(h8, make_uint4(fh[0], fh[1], fh[2], fh[3]));
  };

  // M16 (64 output channels per workgroup, un-pooled gradient): v_mfma_f32_16x16x32_bf16, K = 32 pixels (the wave's two rows) per step,
  // the 32 x 32 block of a tap as four 16 x 16 tiles -- see wgrad3x3_mm.hip
  constexpr bool M16 = PW == 2 && !POOLED;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x16 acc[M16 ? 1 : 9];
  f32x4 a4[M16 ? 9 : 1][4];          // [tap][ci tile * 2 + co tile]
#pragma unroll
  for (int t = 0; t < (M16 ? 1 : 9); ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
  for (int t = 0; t < (M16 ? 9 : 1); ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) a4[t][k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A launch always has jt.ngroups groups per block combination -- the shares, the slabs and the order of every sum are fixed by
  // the job sizes alone -- and a workgroup takes the groups grp, grp + gstep, ...: on the full grid (gstep = ngroups) exactly one,
  // on a reduced grid (ugn_set_persistent_wgs: CUs left to RCCL) several in turn.  Results are bit-identical on every grid.
  for (bool first_group = true; grp < jt.ngroups; grp += jt.gstep, first_group = false) {
  const int s0 = (int)((long long)grp * total / jt.ngroups), s1 = (int)((long long)(grp + 1) * total / jt.ngroups);
  if (s0 >= s1) continue;
  if (!first_group) __syncthreads();        // the previous group's slab combine has finished reading its scratch
  // ---- prologue: tiles of strip s0 -> set 0
  {
    const StripSrc S0 = strip_src(s0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) issue(j, S0, 0);
  }
  int b = 0;
  int jb = job_of(s0);
  for (int s = s0; s < s1; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();      // strip s is in set b; nobody reads the other set any more
    // the tiles of strip s + 1 -> the other set, issued between the taps below
    const bool have_in = s + 1 < s1;
    const StripSrc Sin = strip_src(have_in ? s + 1 : s);
    const LDS_PTR(char) in_b = lds + b * SET + ks * RPW * (18 * 64) + lane_off;
    const LDS_PTR(char) dz_b = lds + b * SET + IN_BYTES + pair * G::DZ_BLOCK + ks * RPW * (16 * 64) + lane_off;
    if constexpr (M16) {
      const int kg = lane >> 4;
      static_assert(!M16 || RPW % 2 == 0, "a K = 32 step is two rows of the wave");
#pragma unroll
      for (int kk = 0; kk < RPW / 2; ++kk) {        // K = 32 steps of the wave: its rows 2 kk, 2 kk + 1
        const LDS_PTR(char) in_m = lds + b * SET + (ks * RPW + 2 * kk) * (18 * 64) + ((kg >> 1) * 18 + 8 * (kg & 1) + q) * 64 + 4 * p * 2;
        const LDS_PTR(char) dz_m = lds + b * SET + IN_BYTES + pair * G::DZ_BLOCK + (ks * RPW + 2 * kk) * (16 * 64) +
                                   ((kg >> 1) * 16 + 8 * (kg & 1) + q) * 64 + 4 * p * 2;
        h8 bh[2];
#pragma unroll
        for (int cot = 0; cot < 2; ++cot) bh[cot] = tr_pair(dz_m, cot * 32, cot * 32 + 4 * 64);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int o = ((t / 3) * 18 + (t % 3)) * 64;
#pragma unroll
          for (int cit = 0; cit < 2; ++cit) {
            const h8 ah = tr_pair(in_m, o + cit * 32, o + cit * 32 + 4 * 64);
#pragma unroll
            for (int cot = 0; cot < 2; ++cot)
              a4[t][cit * 2 + cot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, ah), __builtin_bit_cast(b8, bh[cot]),
                                                                             a4[t][cit * 2 + cot], 0, 0, 0);
          }
          if (kk * 9 + t < NJ && have_in) {
            __builtin_amdgcn_sched_barrier(0);
            issue(kk * 9 + t, Sin, b ^ 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      // k-step = pixel row rr of the wave: lane half h covers pixels 8h .. 8h+7 (two 4-pixel blocks)
      h8 bh;
      if constexpr (POOLED) {
        pooled_frag(b, ks * RPW + rr, bh);
      } else {
        bh = tr_pair(dz_b, (rr * 16) * 64, (rr * 16 + 4) * 64);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t % 3;
        const int o = ((rr + dy) * 18 + dx) * 64;
        const h8 ah = tr_pair(in_b, o, o + 4 * 64);
        acc[t] = mfma_b8(ah, bh, acc[t]);
        // one LDS-DMA piece after every tap (every second tap where there are two k-steps) until the wave's pieces are out
        constexpr int EVERY = RPW == 1 ? 1 : 2;
        const int slot = rr * 9 + t;
        if (slot % EVERY == 0 && slot / EVERY < NJ && have_in) {
          __builtin_amdgcn_sched_barrier(0);
          issue(slot / EVERY, Sin, b ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
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
        if constexpr (M16) {
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
        for (int i = 0; i < 16; ++i) scr[wave * 1024 + i * 64 + lane] = a[i];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2 * PW; ++k) {
          const int e = tid + 512 * k, pr = e >> 10, idx = e & 1023;
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < KS; ++w) sum += scr[(w * PW + pr) * 1024 + idx];
          const int reg = idx >> 6, ln = idx & 63;
          const int ci = M16 ? 16 * (reg >> 3) + 4 * (ln >> 4) + (reg & 3) : (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5);
          const int co = pr * 32 + (M16 ? 16 * ((reg >> 2) & 1) + (ln & 15) : (ln & 31));
          slab[(t * 32 + ci) * COW + co] = sum;
        }
      }
#pragma unroll
      for (int t = 0; t < (M16 ? 1 : 9); ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
      for (int t = 0; t < (M16 ? 9 : 1); ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) a4[t][k] = f32x4{0.f, 0.f, 0.f, 0.f};
      jb = jn;
    }
    b ^= 1;
  }
  }
}

struct WgFinish {
  const float* slab[kWgMaxJobs];
  float* dw[kWgMaxJobs];
  int ng[kWgMaxJobs];
};
// dW[tap][ci][co] (HWIO) = 2^-(e_in + e_dz) * sum over the job's groups, in order.  One thread per element; blockIdx.y = job.
__global__ __launch_bounds__(256) void wgrad_bf_finish(const WgFinish ft, int CI, int CO, int COW) {
  const int j = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 9 * CI * CO) return;
  const int co = e % CO, ci = (e / CO) % CI, tap = e / (CO * CI);
  const int ncoc = CO / COW, combo = (ci >> 5) * ncoc + co / COW;
  const int ng = ft.ng[j];
  const float* sl = ft.slab[j] + (size_t)combo * ng * (9 * 32 * COW) + (tap * 32 + (ci & 31)) * COW + (co % COW);
  float sum = 0.f;
  for (int g = 0; g < ng; ++g) sum += sl[(size_t)g * (9 * 32 * COW)];
  ft.dw[j][e] = sum;
}

template <int CI, int CO>
constexpr int wg_ngroups() { return 256 / ((CI / 32) * (CO / WGeo<CO, 0>::COW)); }

template <int CI, int CO, int HW, int POOLED>
int launch_wgrad(const uint16_t* const* in, const uint16_t* const* dz, const uint8_t* const* dz_idx, float* const* dw, const int* n,
                 int njobs, float* ws, size_t ws_floats, hipStream_t st) {
  using G = WGeo<CO, POOLED>;
  constexpr int NCOMBO = (CI / 32) * (CO / G::COW), NG = wg_ngroups<CI, CO>();
  constexpr int SPI = (HW / G::SR) * (HW / 16);
  constexpr int LDS = wg_lds_bytes<CO, POOLED>();
  static_assert(LDS <= 163840 && NG % 8 == 0, "geometry");
  auto kern = wgrad_bf_kernel<CI, CO, HW, POOLED>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { ugn_set_error("wgrad_bf: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const void* zeros = zero_block();
  if (!zeros) { ugn_set_error("wgrad_bf: cannot allocate the zero block"); return UGN_EINVAL; }
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
  }
  if (used > ws_floats) { ugn_set_error("wgrad_bf: workspace too small (%zu floats needed, %zu given)", used, ws_floats); return UGN_EINVAL; }
  // a group with an EMPTY share writes nothing, but may lie between two groups of a job's slab list: such a slab must add 0
  // (only possible when there are fewer strips than groups)
  if (total < NG) {
    hipError_t me = hipMemsetAsync(ws, 0, used * sizeof(float), st);
    if (me != hipSuccess) { ugn_set_error("wgrad_bf: memset: %s", hipGetErrorString(me)); return (int)me; }
  }
  hipLaunchKernelGGL(kern, dim3(NGE * NCOMBO), dim3(512), LDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("wgrad_bf");
  hipLaunchKernelGGL(wgrad_bf_finish, dim3((9 * CI * CO + 255) / 256, njobs), dim3(256), 0, st, ft, CI, CO, G::COW);
  UGN_CHECK_LAUNCH("wgrad_bf finish");
  return 0;
}

template <int CI, int CO>
size_t ws_floats_for(int njobs) {
  using G = WGeo<CO, 0>;
  constexpr int NCOMBO = (CI / 32) * (CO / G::COW), NG = wg_ngroups<CI, CO>();
  return (size_t)NCOMBO * (NG + njobs) * 9 * 32 * G::COW;
}

}  // namespace

extern "C" size_t ugn_bf_conv3x3_wgrad_ws(int hw, int cin, int cout) {
#define WS(CI_, CO_, HW_) if (cin == CI_ && cout == CO_ && hw == HW_) return ws_floats_for<CI_, CO_>(kWgMaxJobs) * sizeof(float);
  WS(32, 32, 64) WS(32, 64, 32) WS(64, 64, 32) WS(64, 128, 16) WS(128, 128, 16)
#undef WS
  return 0;
}

extern "C" int ugn_bf_conv3x3_wgrad_multi(const uint16_t* const* in, const uint16_t* const* dz, const uint8_t* const* dz_idx,
                                          float* const* dw, const int* n, int njobs, int hw, int cin, int cout, void* ws,
                                          size_t ws_bytes, void* stream) {
  UGN_REQUIRE(in && dz && dw && n && ws, "ugn_bf_conv3x3_wgrad_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kWgMaxJobs, "ugn_bf_conv3x3_wgrad_multi: njobs must be 1..%d (got %d)", kWgMaxJobs, njobs);
  const bool pooled = dz_idx != nullptr && dz_idx[0] != nullptr;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(in[j] && dz[j] && dw[j] && n[j] > 0, "ugn_bf_conv3x3_wgrad_multi: null pointer or n <= 0 in job %d", j);
    UGN_REQUIRE(pooled == (dz_idx != nullptr && dz_idx[j] != nullptr), "ugn_bf_conv3x3_wgrad_multi: dz_idx for all jobs or none");
  }
  hipStream_t st = (hipStream_t)stream;
  float* wsf = (float*)ws;
  const size_t wfl = ws_bytes / sizeof(float);
#define WG(CI_, CO_, HW_, P_)                                        \
  if (cin == CI_ && cout == CO_ && hw == HW_ && pooled == (P_ != 0)) \
    return launch_wgrad<CI_, CO_, HW_, P_>(in, dz, dz_idx, dw, n, njobs, wsf, wfl, st);
  WG(32, 32, 64, 1) WG(32, 64, 32, 0) WG(64, 64, 32, 1) WG(64, 128, 16, 0) WG(128, 128, 16, 0)
#undef WG
  ugn_set_error("ugn_bf_conv3x3_wgrad_multi: unsupported shape cin=%d cout=%d hw=%d pooled=%d", cin, cout, hw, (int)pooled);
  return UGN_EINVAL;
}
