// Weight gradient of the 3x3 layers on IEEE fp32 tensors, multiplied on v_mfma_f32_16x16x32_bf16 through the exact three-way bf16
// split of both operands (x3_common.h).  Replaces the implicit TF Conv2DBackpropFilter (+ MaxPoolGrad) of reference
// nets/mj_uwyhNets_ba.py:431-462, like wgrad3x3_wino.hip.
//
//   dW[tap][ci][co] = sum over images and pixels of  in[y + dy - 1][x + dx - 1][ci] * dz[y][x][co]
//
// Per tap a GEMM with M = ci, N = co and K = PIXELS.  Both operands lie in LDS pixel-major (three bf16 planes, 64 B per pixel, 32
// channels and plane), so an MFMA fragment -- 8 consecutive pixels of ONE channel per lane -- is two ds_read_b64_tr_b16 (gfx950's
// transposing LDS read: a 4-pixel x 16-channel block per 16 lanes, delivered channel-major), and a tap's shift is an address offset.
//   * A 512-thread workgroup owns 32 input x COW (32 or 64) output channels and all 9 taps.  Work items are SR-row x 16-column
//     strips (SR = 8 for COW 32, 4 for COW 64); the strips of all jobs of a launch form one list and the groups of a block
//     combination own equal contiguous shares of it.  wave = (K split ks: strip rows 2 ks, 2 ks + 1 = one K-32 step, input-channel
//     tile, 32-channel output block): 1 x 2 tiles x 9 taps = 72 accumulator registers, 108 MFMAs per strip.
//   * The strip's tiles get into LDS through registers, one strip ahead: fp32 global loads at the top of a strip, the three-way
//     split and the ds_write_b128 behind the fifth tap.  The pooled gradient of a MaxPool'ed layer is un-pooled on the way (value
//     where the argmax byte points, zeros elsewhere).  Two buffer sets, ONE barrier per strip.
//   * Bank conflicts: a ds_read_b64_tr_b16 pass is lanes 0-31 = two k groups; they are rows r, r + 1 of the same columns and the two
//     32-byte halves of a pixel's record are swapped on odd tile rows (as in wgrad3x3_mm.hip: 0.000 conflicts there).
//   * The two MaxPool'ed layers run on the SPARSE matrix pipe (wgrad_x3s_kernel below): half the matrix instructions, a quarter of the
//     gradient's split work.
//   * At the end of a share (or a job boundary inside it) the K-split waves are added through LDS in a fixed order and the partial
//     sums leave as a slab; wgrad_x3_finish adds the slabs of a job in a fixed order.  No atomics: bitwise reproducible.
#include "x3_common.h"

using namespace ugn_x3;


namespace {

typedef short s4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T) __attribute__((address_space(3))) T*

struct WgJob {
  const float* in;           // [n][hw][hw][ci]
  const float* dz;           // [n][hw][hw][co]   (pooled: [n][hw/2][hw/2][co])
  const uint8_t* dz_idx;     // pooled: argmax bytes
  float* slab;               // [combo][ng][9][32][COW]
  int g0, ng;                // the groups (of every combination) that touch this job
};
struct WgJobs {
  WgJob job[kMaxJobs];
  int start[kMaxJobs + 1];   // first strip of job j; [njobs..] = total
  int ngroups;               // groups per block combination (a multiple of 8)
};

template <int CO>
struct WGeo {
  static constexpr int COW = CO >= 64 ? 64 : 32;     // output channels of a workgroup
  static constexpr int SR = COW == 64 ? 4 : 8;       // pixel rows of a strip
  static constexpr int KS = SR / 2;                  // K-split waves (one K-32 step = two pixel rows each)
  static constexpr int KGZ = COW / 8;                // 8-channel groups of a gradient pixel
  static constexpr int IN_PIX = (SR + 2) * 18;
  static constexpr int IN_PLANE = IN_PIX * 64;
  static constexpr int IN_BYTES = 3 * IN_PLANE;
  static constexpr int IN_UNITS = IN_PIX * 4;        // (halo pixel, 8-channel group)
  static constexpr int DZ_BLOCK = SR * 16 * 64;      // one 32-channel block of the strip, one plane
  static constexpr int DZ_PLANE = (COW / 32) * DZ_BLOCK;
  static constexpr int DZ_BYTES = 3 * DZ_PLANE;
  static constexpr int SET = IN_BYTES + DZ_BYTES;    // 59,136 (COW 32) / 45,312 (COW 64)
  static constexpr int LDS = 2 * SET;
};

__device__ __forceinline__ uint4 tr_pair(const LDS_PTR(char) base, int off0, int off1) {
  const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(base + off0));
  const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(base + off1));
  const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  return make_uint4(ua.x, ua.y, ub.x, ub.y);
}

template <int CI, int CO, int HW, int POOLED, int NP>
__global__ __launch_bounds__(512, 2) void wgrad_x3_kernel(const WgJobs jt) {
  using G = WGeo<CO>;
  constexpr int COW = G::COW, SR = G::SR, KS = G::KS, KGZ = G::KGZ, SET = G::SET;
  constexpr int IN_PLANE = G::IN_PLANE, IN_BYTES = G::IN_BYTES, DZ_PLANE = G::DZ_PLANE, DZ_BLOCK = G::DZ_BLOCK;
  constexpr int NCOC = CO / COW, NCOMBO = (CI / 32) * NCOC;
  constexpr int SPX = HW / 16, SPI = (HW / SR) * SPX;       // strips per image row / per image
  static_assert(HW % SR == 0 && KS * (COW / 32) * 2 == 8, "eight waves: K split x input-channel tile x output block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const LDS_PTR(char) lds = (LDS_PTR(char))smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (combination, group): all combinations of a group sit on ONE XCD (blockIdx % 8), so the strip they all read comes
  // from HBM once per XCD and from that XCD's L2 afterwards
  const int bid = blockIdx.x, xcd = bid & 7, rest = bid >> 3;
  const int combo = rest % NCOMBO;
  const int grp = (rest / NCOMBO) * 8 + xcd;
  if (grp >= jt.ngroups) return;
  const int cic = combo / NCOC, coc = combo % NCOC;
  const int total = jt.start[kMaxJobs];
  const int ks = wave % KS, wrest = wave / KS, cit = wrest & 1, cob = wrest >> 1;
  const int s0 = (int)((long long)grp * total / jt.ngroups), s1 = (int)((long long)(grp + 1) * total / jt.ngroups);
  if (s0 >= s1) return;

  auto job_of = [&](int s) {
    int jb = 0;
#pragma unroll
    for (int j = 1; j < kMaxJobs; ++j) jb += s >= jt.start[j] ? 1 : 0;
    return jb;
  };

  // ---- staging units of this thread
  // input halo: unit u = tid + 512 k -> halo pixel u >> 2 (row hr, column hc), channels 8 (u & 3) .. + 7 of the 32-channel block
  // (slots beyond the IN_UNITS units repeat the last units -- same loads, same values to the same addresses: straight-line code that
  //  can be scheduled between the MFMAs of a tap)
  constexpr int NXU = (G::IN_UNITS + 511) / 512;
  int xu_lds[NXU], xu_g[NXU], xu_rc[NXU];
#pragma unroll
  for (int k = 0; k < NXU; ++k) {
    int u = tid + 512 * k;
    if (u >= G::IN_UNITS) u -= 512 * NXU - G::IN_UNITS;
    const int hp = u >> 2, kg = u & 3;
    const int hr = hp / 18, hc = hp - hr * 18;
    xu_lds[k] = hp * 64 + ((kg ^ ((hr & 1) << 1)) << 4);             // halves swapped on odd tile rows
    xu_g[k] = ((hr - 1) * HW + (hc - 1)) * CI + kg * 8;
    xu_rc[k] = (hr << 8) | hc;
  }
  // gradient tile: un-pooled: unit tid -> pixel tid / KGZ of the strip, channels 8 (tid % KGZ) .. of the COW block;
  // pooled: unit tid -> (pooled pixel, channel group, window position): the value lands at its position or a zero does
  const int zkg = POOLED ? (tid >> 2) % KGZ : tid % KGZ, zpx = POOLED ? (tid >> 2) / KGZ : tid / KGZ, zpos = tid & 3;
  const int zrow = POOLED ? 2 * (zpx >> 3) + (zpos >> 1) : zpx >> 4, zcol = POOLED ? 2 * (zpx & 7) + (zpos & 1) : zpx & 15;
  const int z_lds = (zkg >> 2) * DZ_BLOCK + (zrow * 16 + zcol) * 64 + (((zkg & 3) ^ ((zrow & 1) << 1)) << 4);
  const int z_g = POOLED ? ((zpx >> 3) * (HW / 2) + (zpx & 7)) * CO + zkg * 8 : (zrow * HW + zcol) * CO + zkg * 8;

  float4 xv[NXU][2], zv[2];
  uint2 zi = make_uint2(0u, 0u);
  auto stage_load = [&](int s) {
    const int jb = job_of(s), ls = s - jt.start[jb];
    const int img = ls / SPI, r = ls % SPI;
    const int sy0 = (r / SPX) * SR, sx0 = (r % SPX) * 16;
    const float* xb = jt.job[jb].in + ((size_t)img * HW * HW + (size_t)(sy0 * HW + sx0)) * CI + cic * 32;
#pragma unroll
    for (int k = 0; k < NXU; ++k) {
      const int y = sy0 + (xu_rc[k] >> 8) - 1, xx = sx0 + (xu_rc[k] & 255) - 1;
      const bool ok = (unsigned)y < (unsigned)HW && (unsigned)xx < (unsigned)HW;
      xv[k][0] = make_float4(0.f, 0.f, 0.f, 0.f);
      xv[k][1] = xv[k][0];
      if (ok) {
        xv[k][0] = *reinterpret_cast<const float4*>(xb + xu_g[k]);
        xv[k][1] = *reinterpret_cast<const float4*>(xb + xu_g[k] + 4);
      }
    }
    if constexpr (POOLED) {
      constexpr int HP = HW / 2;
      const size_t o = ((size_t)img * HP * HP + (size_t)((sy0 / 2) * HP + sx0 / 2)) * CO + coc * COW + z_g;
      zv[0] = *reinterpret_cast<const float4*>(jt.job[jb].dz + o);
      zv[1] = *reinterpret_cast<const float4*>(jt.job[jb].dz + o + 4);
      zi = *reinterpret_cast<const uint2*>(jt.job[jb].dz_idx + o);
    } else {
      const float* zb = jt.job[jb].dz + ((size_t)img * HW * HW + (size_t)(sy0 * HW + sx0)) * CO + coc * COW + z_g;
      zv[0] = *reinterpret_cast<const float4*>(zb);
      zv[1] = *reinterpret_cast<const float4*>(zb + 4);
    }
  };
  constexpr int NSU = NXU + 1;             // store steps of a strip: the input units, then the gradient unit
  auto stage_store_unit = [&](int b, int k) {
    if (k < NXU) {
      char* xs = smem + b * SET;
      uint4 p0, p1, p2;
      split8(xv[k < NXU ? k : 0][0], xv[k < NXU ? k : 0][1], p0, p1, p2);
      *reinterpret_cast<uint4*>(xs + xu_lds[k < NXU ? k : 0]) = p0;
      *reinterpret_cast<uint4*>(xs + IN_PLANE + xu_lds[k < NXU ? k : 0]) = p1;
      *reinterpret_cast<uint4*>(xs + 2 * IN_PLANE + xu_lds[k < NXU ? k : 0]) = p2;
      return;
    }
    char* zs = smem + b * SET + IN_BYTES + z_lds;
    uint4 q0, q1, q2;
    split8(zv[0], zv[1], q0, q1, q2);
    if constexpr (POOLED) {
      unsigned m[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const unsigned w = d < 2 ? zi.x : zi.y;
        const unsigned b0 = (w >> (16 * (d & 1))) & 0xffu, b1 = (w >> (16 * (d & 1) + 8)) & 0xffu;
        m[d] = (b0 == (unsigned)zpos ? 0x0000ffffu : 0u) | (b1 == (unsigned)zpos ? 0xffff0000u : 0u);
      }
      q0 = make_uint4(q0.x & m[0], q0.y & m[1], q0.z & m[2], q0.w & m[3]);
      q1 = make_uint4(q1.x & m[0], q1.y & m[1], q1.z & m[2], q1.w & m[3]);
      q2 = make_uint4(q2.x & m[0], q2.y & m[1], q2.z & m[2], q2.w & m[3]);
    }
    *reinterpret_cast<uint4*>(zs) = q0;
    *reinterpret_cast<uint4*>(zs + DZ_PLANE) = q1;
    *reinterpret_cast<uint4*>(zs + 2 * DZ_PLANE) = q2;
  };

  // ---- fragment addressing: lane (i = lane & 15, k group kg = lane >> 4) = 8 consecutive pixels of the wave's 32: row kr = kg & 1,
  // columns 8 (kg >> 1) .. + 7; within a 16-lane group lane (q, p) supplies pixel q of the 4-pixel block, channels 4 p .. + 3
  const int kg = lane >> 4, kr = kg & 1, kx = kg >> 1, q = (lane >> 2) & 3, p = lane & 3;
  const int in_l0 = (ks * 2 + kr) * (18 * 64) + (8 * kx + q) * 64 + 4 * p * 2;
  // the 32-byte half that holds channel tile `cit` on a tap row of parity (kr + dy) & 1
  const int in_l[2] = {in_l0 + 32 * (cit ^ kr), in_l0 + 32 * (cit ^ kr ^ 1)};
  const int dz_l0 = IN_BYTES + cob * DZ_BLOCK + ((ks * 2 + kr) * 16 + 8 * kx + q) * 64 + 4 * p * 2;
  const int dz_l[2] = {dz_l0 + 32 * (0 ^ kr), dz_l0 + 32 * (1 ^ kr)};      // [output-channel tile of the wave's block]

  f32x4 a4[9][2];                 // [tap][output tile]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) a4[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_load(s0);
#pragma unroll
  for (int k = 0; k < NSU; ++k) stage_store_unit(0, k);
  if (s0 + 1 < s1) stage_load(s0 + 1);
  int b = 0;
  int jb = job_of(s0);
  for (int s = s0; s < s1; ++s) {
    __syncthreads();              // strip s is complete in set b; nobody reads the other set any more
    const bool have_next = s + 1 < s1;
    const LDS_PTR(char) sb = lds + b * SET;
    uint4 zb[2][3];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) zb[c][pl] = tr_pair(sb + dz_l[c], pl * DZ_PLANE, pl * DZ_PLANE + 4 * 64);
    uint4 xa[2][3];
    auto load_a = [&](int set, int t) {
      const int dy = t / 3, dx = t % 3;
      const int o = (dy * 18 + dx) * 64;
      const LDS_PTR(char) im = sb + in_l[dy & 1];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) xa[set][pl] = tr_pair(im, pl * IN_PLANE + o, pl * IN_PLANE + o + 4 * 64);
    };
    load_a(0, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) load_a((t + 1) & 1, t + 1);
      // the fp32 values of the strip after next: their registers were split and written during the taps before
      if (t == NSU + 1 && s + 2 < s1) stage_load(s + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x4 acc = a4[t][c];
#pragma unroll
        for (int i = 0; i < NP; ++i) acc = mfma_bf(xa[t & 1][prod_w<NP>(i)], zb[c][prod_x<NP>(i)], acc);
        a4[t][c] = acc;
      }
      // the next strip's units are split and written behind the MFMAs of taps 1 .. NSU, one unit per tap, their vector instructions
      // interleaved with the matrix ones (with no strip after this one: stale values into the idle set)
      if (t >= 1 && t <= NSU) {
        stage_store_unit(b ^ 1, t - 1);
#pragma unroll
        for (int k = 0; k < 2 * NP; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int jn = have_next ? job_of(s + 1) : -1;
    if (jn != jb) {
      // ---- job (or share) finished: add the K-split waves through LDS in a fixed order, write the slab.  Scratch = the set just
      // multiplied (the next strip already sits in the other one).
      float* slab = jt.job[jb].slab + ((size_t)combo * jt.job[jb].ng + (grp - jt.job[jb].g0)) * (9 * 32 * COW);
      float* scr = reinterpret_cast<float*>(smem + b * SET);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) scr[wave * 512 + (c * 4 + r) * 64 + lane] = a4[t][c][r];
        __syncthreads();
        // element e of the tap: (unit = cit + 2 cob, output tile c, register r, lane ln); tile register r of lane ln = input channel
        // 16 cit + 4 (ln >> 4) + r (M), output channel 32 cob + 16 c + (ln & 15) (N)
#pragma unroll
        for (int k = 0; k < (32 * COW) / 512; ++k) {
          const int e = tid + 512 * k, unit = e >> 9, idx = e & 511;
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < KS; ++w) sum += scr[(unit * KS + w) * 512 + idx];
          const int c = idx >> 8, r = (idx >> 6) & 3, ln = idx & 63;
          const int ci = 16 * (unit & 1) + 4 * (ln >> 4) + r, co = 32 * (unit >> 1) + 16 * c + (ln & 15);
          slab[(t * 32 + ci) * COW + co] = sum;
        }
      }
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) a4[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      jb = jn;
    }
    b ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pooled layers on the SPARSE matrix pipe (v_smfmac_f32_16x16x64_bf16), as round 4 did for the f16x2 set (wgrad3x3_mm.hip).
// The un-pooled gradient of a MaxPool'ed layer has ONE non-zero per 2x2 window, so along a pixel row every pair (2j, 2j + 1) holds at
// most one: 2:4 structured along K = pixels.  v_smfmac takes the sparse operand compressed -- per lane (m, k block kb) the 8 kept
// values of its 16 k and a 2-bit position each -- and issues at the rate of the dense 16x16x32: HALF the matrix instructions.
//   M = output channels (dz^T, sparse A), N = input channels (x, dense B), K 64 = 4 strip rows x 16 pixels, k = 16 row + pixel.
//   * the compressed operand IS the pooled tensor: slot s of row y = the pooled gradient of window (y >> 1, s) where the window's argmax
//     lies in row y & 1 (else 0), position 2 (s & 1) + (argmax & 1).  The pooled tile is staged AS POOLED (three bf16 planes + the
//     argmax bytes): a quarter of the un-pooled tile's split work, no un-pooling at all.
//   * the dense operand of a tap = two 16x16x32 fragments per plane (rows kb >> 1 and 2 + (kb >> 1), pixels 8 (kb & 1) .. + 7).
// 16-row strips; wave = (row group rg: strip rows 4 rg .. + 3 = the K 64 of one v_smfmac, input-channel tile); 32 output channels per
// workgroup (2 tiles x 9 taps = 72 accumulator registers), 108 v_smfmac per strip and wave = the work of 216 dense MFMAs.
// Operand layouts as probed in round 4 (tools/experiments/smfmac_probe.hip): A lane (m = lane & 15, kb = lane >> 4) slots s = 0..7 =
// dense k 16 kb + 4 (s >> 1) + position; B lane (n = lane & 15, kb) elements 0-7 = k 8 kb .., 8-15 = k 32 + 8 kb ...
struct SGeo {
  static constexpr int COW = 32, SR = 16;
  static constexpr int IN_PIX = 18 * 18, IN_PLANE = IN_PIX * 64, IN_BYTES = 3 * IN_PLANE;     // 62,208
  static constexpr int IN_UNITS = IN_PIX * 4;
  static constexpr int PZ_PIX = 64, PZ_PLANE = PZ_PIX * 64, PZ_VAL = 3 * PZ_PLANE;             // 8 x 8 pooled pixels x 32 channels x 3 planes
  static constexpr int PZ_IDX = PZ_PIX * 32;
  static constexpr int SET = IN_BYTES + PZ_VAL + PZ_IDX;                                       // 76,544
  static constexpr int LDS = 2 * SET;                                                          // 153,088
};
typedef __bf16 sb8 __attribute__((ext_vector_type(8)));
typedef __bf16 sb16 __attribute__((ext_vector_type(16)));
struct u8x { uint4 lo, hi; };
__device__ __forceinline__ f32x4 smfmac_bf(uint4 a, const u8x& b, f32x4 c, int idx) {
  return __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(sb8, a), __builtin_bit_cast(sb16, b), c, idx, 0, 0);
}

template <int CI, int CO, int HW, int NP>
__global__ __launch_bounds__(512, 2) void wgrad_x3s_kernel(const WgJobs jt) {
  using G = SGeo;
  constexpr int COW = G::COW, SR = G::SR, SET = G::SET, IN_PLANE = G::IN_PLANE, IN_BYTES = G::IN_BYTES, PZ_PLANE = G::PZ_PLANE;
  constexpr int NCOC = CO / COW, NCOMBO = (CI / 32) * NCOC;
  constexpr int SPX = HW / 16, SPI = (HW / SR) * SPX, HP = HW / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const LDS_PTR(char) lds = (LDS_PTR(char))smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x, xcd = bid & 7, rest = bid >> 3;
  const int combo = rest % NCOMBO;
  const int grp = (rest / NCOMBO) * 8 + xcd;
  if (grp >= jt.ngroups) return;
  const int cic = combo / NCOC, coc = combo % NCOC;
  const int total = jt.start[kMaxJobs];
  const int rg = wave & 3, unit = wave >> 2;                 // row group, input-channel tile
  const int s0 = (int)((long long)grp * total / jt.ngroups), s1 = (int)((long long)(grp + 1) * total / jt.ngroups);
  if (s0 >= s1) return;
  auto job_of = [&](int s) {
    int jb = 0;
#pragma unroll
    for (int j = 1; j < kMaxJobs; ++j) jb += s >= jt.start[j] ? 1 : 0;
    return jb;
  };
  // ---- staging: the 18 x 18 input halo (three slots per thread; slots beyond the 1296 units repeat the last ones), the 8 x 8 pooled
  // gradient pixels (threads 256.. repeat 0..255) and their argmax bytes (16 B per thread, repeated four times)
  int xu_lds[3], xu_g[3], xu_rc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int u = tid + 512 * k;
    if (u >= G::IN_UNITS) u -= 1536 - G::IN_UNITS;
    const int hp = u >> 2, kg = u & 3;
    const int hr = hp / 18, hc = hp - hr * 18;
    xu_lds[k] = hp * 64 + ((kg ^ (((hc >> 3) & 1) << 1)) << 4);      // the 32-byte halves swapped for tile columns 8-15
    xu_g[k] = ((hr - 1) * HW + (hc - 1)) * CI + kg * 8;
    xu_rc[k] = (hr << 8) | hc;
  }
  const int pu = tid & 255, zpp = pu >> 2, zkg = pu & 3;
  const int z_lds = zpp * 64 + zkg * 16;
  const int z_g = ((zpp >> 3) * HP + (zpp & 7)) * CO + zkg * 8;
  const int ai = tid & 127, app = ai >> 1, ahalf = ai & 1;
  const int a_lds = G::PZ_VAL + app * 32 + ahalf * 16;
  const int a_g = ((app >> 3) * HP + (app & 7)) * CO + ahalf * 16;
  float4 xv[3][2], zv[2];
  uint4 av = make_uint4(0u, 0u, 0u, 0u);
  auto stage_load = [&](int s) {
    const int jb = job_of(s), ls = s - jt.start[jb];
    const int img = ls / SPI, r = ls % SPI;
    const int sy0 = (r / SPX) * SR, sx0 = (r % SPX) * 16;
    const float* xb = jt.job[jb].in + ((size_t)img * HW * HW + (size_t)(sy0 * HW + sx0)) * CI + cic * 32;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int y = sy0 + (xu_rc[k] >> 8) - 1, xx = sx0 + (xu_rc[k] & 255) - 1;
      const bool ok = (unsigned)y < (unsigned)HW && (unsigned)xx < (unsigned)HW;
      xv[k][0] = make_float4(0.f, 0.f, 0.f, 0.f);
      xv[k][1] = xv[k][0];
      if (ok) {
        xv[k][0] = *reinterpret_cast<const float4*>(xb + xu_g[k]);
        xv[k][1] = *reinterpret_cast<const float4*>(xb + xu_g[k] + 4);
      }
    }
    const size_t o = ((size_t)img * HP * HP + (size_t)((sy0 / 2) * HP + sx0 / 2)) * CO + coc * COW;
    zv[0] = *reinterpret_cast<const float4*>(jt.job[jb].dz + o + z_g);
    zv[1] = *reinterpret_cast<const float4*>(jt.job[jb].dz + o + z_g + 4);
    av = *reinterpret_cast<const uint4*>(jt.job[jb].dz_idx + o + a_g);
  };
  constexpr int NSU = 4;            // store steps of a strip: three input slots, then the pooled gradient + its argmax bytes
  auto stage_store_unit = [&](int b, int k) {
    char* base = smem + b * SET;
    uint4 p0, p1, p2;
    if (k < 3) {
      split8(xv[k < 3 ? k : 0][0], xv[k < 3 ? k : 0][1], p0, p1, p2);
      *reinterpret_cast<uint4*>(base + xu_lds[k < 3 ? k : 0]) = p0;
      *reinterpret_cast<uint4*>(base + IN_PLANE + xu_lds[k < 3 ? k : 0]) = p1;
      *reinterpret_cast<uint4*>(base + 2 * IN_PLANE + xu_lds[k < 3 ? k : 0]) = p2;
      return;
    }
    split8(zv[0], zv[1], p0, p1, p2);
    *reinterpret_cast<uint4*>(base + IN_BYTES + z_lds) = p0;
    *reinterpret_cast<uint4*>(base + IN_BYTES + PZ_PLANE + z_lds) = p1;
    *reinterpret_cast<uint4*>(base + IN_BYTES + 2 * PZ_PLANE + z_lds) = p2;
    *reinterpret_cast<uint4*>(base + IN_BYTES + a_lds) = av;
  };

  const int kb = lane >> 4, mi = lane & 15, q = (lane >> 2) & 3, p = lane & 3;
  const int hsel = kb & 1;            // the half that holds input-channel tile 0 at this lane's first column
  const int in_r = ((4 * rg + (kb >> 1)) * 18 + 8 * (kb & 1) + q) * 64 + 4 * p * 2;
  // [dx]: byte offset of tile 0's half for the SECOND read of a pair (the first: 32 hsel)
  const int sec0[3] = {32 * hsel, 32 * (hsel ^ (q + 1 >= 4 ? 1 : 0)), 32 * (hsel ^ (q + 2 >= 4 ? 1 : 0))};
  const int fst_off = in_r + ((32 * hsel) ^ (32 * unit));

  f32x4 a4[9][2];                 // [tap][output-channel tile]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) a4[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_load(s0);
#pragma unroll
  for (int k = 0; k < NSU; ++k) stage_store_unit(0, k);
  if (s0 + 1 < s1) stage_load(s0 + 1);
  int b = 0;
  int jb = job_of(s0);
  for (int s = s0; s < s1; ++s) {
    __syncthreads();              // strip s is complete in set b; nobody reads the other set any more
    const bool have_next = s + 1 < s1;
    const LDS_PTR(char) sb = lds + b * SET;
    // ---- the compressed operands of the strip: [output tile][plane] + the position word
    uint4 sa[2][3];
    int sidx[2];
    {
      const int y = 4 * rg + kb;
      const unsigned rowpar = (unsigned)(y & 1);
#pragma unroll
      for (int cot = 0; cot < 2; ++cot) {
        const LDS_PTR(char) pv = sb + IN_BYTES + (y >> 1) * 512 + q * 64 + (16 * cot + 4 * p) * 2;
        const unsigned char* pi8 = reinterpret_cast<const unsigned char*>(smem) + b * SET + IN_BYTES + G::PZ_VAL + ((y >> 1) * 8) * 32 + 16 * cot + mi;
        unsigned m[4], iw = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {            // dword d = slots 2d, 2d + 1 = pooled columns 2d, 2d + 1
          const unsigned i0 = pi8[(2 * d) * 32], i1 = pi8[(2 * d + 1) * 32];
          m[d] = ((i0 >> 1) == rowpar ? 0x0000ffffu : 0u) | ((i1 >> 1) == rowpar ? 0xffff0000u : 0u);
          iw |= ((i0 & 1u) | ((2u | (i1 & 1u)) << 2)) << (4 * d);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const uint4 v = tr_pair(pv, pl * PZ_PLANE, pl * PZ_PLANE + 4 * 64);
          sa[cot][pl] = make_uint4(v.x & m[0], v.y & m[1], v.z & m[2], v.w & m[3]);
        }
        sidx[cot] = (int)iw;
      }
    }
    // ---- dense fragments of a tap: per plane 16 k values = rows (kb >> 1), 2 + (kb >> 1) of the wave's four, pixels 8 (kb & 1) .. + 7
    u8x xb[2][3];
    auto load_b = [&](int set, int t) {
      const int o = ((t / 3) * 18 + (t % 3)) * 64;
      const LDS_PTR(char) fst = sb + fst_off;
      const LDS_PTR(char) sec = sb + in_r + (sec0[t % 3] ^ (32 * unit));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const s4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(fst + pl * IN_PLANE + o));
        const s4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(sec + pl * IN_PLANE + o + 4 * 64));
        const s4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(fst + pl * IN_PLANE + o + 2 * 18 * 64));
        const s4 r3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s4))(sec + pl * IN_PLANE + o + 2 * 18 * 64 + 4 * 64));
        const uint2 u0 = __builtin_bit_cast(uint2, r0), u1 = __builtin_bit_cast(uint2, r1), u2 = __builtin_bit_cast(uint2, r2),
                    u3 = __builtin_bit_cast(uint2, r3);
        xb[set][pl].lo = make_uint4(u0.x, u0.y, u1.x, u1.y);
        xb[set][pl].hi = make_uint4(u2.x, u2.y, u3.x, u3.y);
      }
    };
    load_b(0, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) load_b((t + 1) & 1, t + 1);
      if (t == NSU + 1 && s + 2 < s1) stage_load(s + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x4 acc = a4[t][c];
#pragma unroll
        for (int i = 0; i < NP; ++i) acc = smfmac_bf(sa[c][prod_w<NP>(i)], xb[t & 1][prod_x<NP>(i)], acc, sidx[c]);
        a4[t][c] = acc;
      }
      if (t >= 1 && t <= NSU) {
        stage_store_unit(b ^ 1, t - 1);
#pragma unroll
        for (int k = 0; k < 2 * NP; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int jn = have_next ? job_of(s + 1) : -1;
    if (jn != jb) {
      // ---- job (or share) finished: add the four row-group waves of a unit through LDS in a fixed order, write the slab
      float* slab = jt.job[jb].slab + ((size_t)combo * jt.job[jb].ng + (grp - jt.job[jb].g0)) * (9 * 32 * COW);
      float* scr = reinterpret_cast<float*>(smem + b * SET);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) scr[wave * 512 + (c * 4 + r) * 64 + lane] = a4[t][c][r];
        __syncthreads();
        // element e: (unit u = input-channel tile, output tile c, register r, lane ln); tile register r of lane ln = output channel
        // 16 c + 4 (ln >> 4) + r (M), input channel 16 u + (ln & 15) (N)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int e = tid + 512 * k, u = e >> 9, idx = e & 511;
          float sum = 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) sum += scr[(u * 4 + g) * 512 + idx];
          const int c = idx >> 8, r = (idx >> 6) & 3, ln = idx & 63;
          const int co = 16 * c + 4 * (ln >> 4) + r, ci = 16 * u + (ln & 15);
          slab[(t * 32 + ci) * COW + co] = sum;
        }
      }
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) a4[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      jb = jn;
    }
    b ^= 1;
  }
}

struct WgFinish {
  const float* slab[kMaxJobs];
  float* dw[kMaxJobs];
  int ng[kMaxJobs];
};
// dW[tap][ci][co] (HWIO) = sum over the job's groups in a FIXED order (bitwise reproducible): the groups are cut into kFinSeg
// contiguous segments, a thread adds one segment's slabs in group order, the segment sums are added in segment order through LDS.
// (Round 6.  One thread per element walking all of a job's up to 256 slabs was a chain of 256 dependent 4-byte reads on 36-576
//  workgroups: 14 us per launch, five launches per step; 32 elements x 8 segments per workgroup read the same bytes on 8x the
//  workgroups with 8x shorter chains.)
constexpr int kFinSeg = 8, kFinEl = 32;
__global__ __launch_bounds__(kFinSeg * kFinEl) void wgrad_x3_finish(const WgFinish ft, int CI, int CO, int COW) {
  __shared__ float part[kFinSeg][kFinEl];
  const int j = blockIdx.y;
  const int el = threadIdx.x % kFinEl, seg = threadIdx.x / kFinEl;
  const int e = blockIdx.x * kFinEl + el;
  const bool ok = e < 9 * CI * CO;
  float sum = 0.f;
  if (ok) {
    const int co = e % CO, ci = (e / CO) % CI, tap = e / (CO * CI);
    const int ncoc = CO / COW, combo = (ci >> 5) * ncoc + co / COW;
    const int ng = ft.ng[j];
    const size_t stride = (size_t)9 * 32 * COW;
    const float* sl = ft.slab[j] + (size_t)combo * ng * stride + (tap * 32 + (ci & 31)) * COW + (co % COW);
    const int g0 = seg * ng / kFinSeg, g1 = (seg + 1) * ng / kFinSeg;
#pragma unroll 4
    for (int g = g0; g < g1; ++g) sum += sl[(size_t)g * stride];
  }
  part[seg][el] = sum;
  __syncthreads();
  if (seg == 0 && ok) {
    float t = part[0][el];
#pragma unroll
    for (int k = 1; k < kFinSeg; ++k) t += part[k][el];
    ft.dw[j][e] = t;
  }
}

#ifndef UGN_X3_SPARSE
#define UGN_X3_SPARSE 1          /* the pooled layers' weight gradients on the sparse matrix pipe (wgrad_x3s_kernel) */
#endif
// geometry of a launch: dense (WGeo<CO>) or sparse (SGeo: 32 output channels per workgroup, 16-row strips)
template <int CO, bool SPARSE>
struct LGeo {
  static constexpr int COW = SPARSE ? SGeo::COW : WGeo<CO>::COW, SR = SPARSE ? SGeo::SR : WGeo<CO>::SR;
  static constexpr int LDS = SPARSE ? SGeo::LDS : WGeo<CO>::LDS;
};
template <int CI, int CO, bool SPARSE>
constexpr int wg_ngroups() { return 256 / ((CI / 32) * (CO / LGeo<CO, SPARSE>::COW)); }

template <int CI, int CO, bool SPARSE>
size_t ws_floats_for(int njobs) {
  using G = LGeo<CO, SPARSE>;
  constexpr int NCOMBO = (CI / 32) * (CO / G::COW), NG = wg_ngroups<CI, CO, SPARSE>();
  return (size_t)NCOMBO * (NG + njobs) * 9 * 32 * G::COW;
}

template <int CI, int CO, int HW, int POOLED, int NP>
int launch_wgrad_np(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx, float* const* dw, const int* n, int njobs,
                    float* ws, size_t ws_floats, hipStream_t st) {
  constexpr bool SPARSE = (POOLED != 0) & (UGN_X3_SPARSE != 0);
  using G = LGeo<CO, SPARSE>;
  constexpr int NCOMBO = (CI / 32) * (CO / G::COW), NG = wg_ngroups<CI, CO, SPARSE>();
  constexpr int SPI = (HW / G::SR) * (HW / 16);
  static_assert(G::LDS <= 163840 && NG % 8 == 0 && HW % G::SR == 0, "geometry");
  void (*kern)(const WgJobs);
  if constexpr (SPARSE) kern = wgrad_x3s_kernel<CI, CO, HW, NP>;
  else kern = wgrad_x3_kernel<CI, CO, HW, POOLED, NP>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) { ugn_set_error("wgrad_x3: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  WgJobs jt = {};
  WgFinish ft = {};
  int total = 0;
  for (int j = 0; j < kMaxJobs; ++j) {
    jt.start[j] = total;
    if (j < njobs) total += n[j] * SPI;
  }
  jt.start[kMaxJobs] = total;
  jt.ngroups = NG;
  const size_t slab_floats = (size_t)9 * 32 * G::COW;
  size_t used = 0;
  for (int j = 0; j < kMaxJobs; ++j) {
    const int jj = j < njobs ? j : njobs - 1;
    jt.job[j].in = in[jj]; jt.job[j].dz = dz[jj]; jt.job[j].dz_idx = POOLED ? dz_idx[jj] : nullptr;
    if (j >= njobs) { jt.job[j].slab = jt.job[jj].slab; jt.job[j].g0 = jt.job[jj].g0; jt.job[j].ng = jt.job[jj].ng; continue; }
    // groups whose share [g*T/NG, (g+1)*T/NG) meets the job's strips [a, bnd)
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
  if (used > ws_floats) { ugn_set_error("wgrad_x3: workspace too small (%zu floats needed, %zu given)", used, ws_floats); return UGN_EINVAL; }
  // a group with an EMPTY share writes nothing, but may lie between two groups of a job's slab list: such a slab must add 0
  // (only possible when there are fewer strips than groups)
  if (total < NG) {
    hipError_t me = hipMemsetAsync(ws, 0, used * sizeof(float), st);
    if (me != hipSuccess) { ugn_set_error("wgrad_x3: memset: %s", hipGetErrorString(me)); return (int)me; }
  }
  hipLaunchKernelGGL(kern, dim3(NG * NCOMBO), dim3(512), G::LDS, st, jt);
  UGN_CHECK_LAUNCH("wgrad_x3");
  hipLaunchKernelGGL(wgrad_x3_finish, dim3((9 * CI * CO + kFinEl - 1) / kFinEl, njobs), dim3(kFinSeg * kFinEl), 0, st, ft, CI, CO, G::COW);
  UGN_CHECK_LAUNCH("wgrad_x3 finish");
  return 0;
}

template <int CI, int CO, int HW, int POOLED>
int launch_wgrad(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx, float* const* dw, const int* n, int njobs,
                 float* ws, size_t ws_floats, int products, hipStream_t st) {
  if (products == 9) return launch_wgrad_np<CI, CO, HW, POOLED, 9>(in, dz, dz_idx, dw, n, njobs, ws, ws_floats, st);
  return launch_wgrad_np<CI, CO, HW, POOLED, kProducts>(in, dz, dz_idx, dw, n, njobs, ws, ws_floats, st);
}

}  // namespace

extern "C" size_t ugn_x3_conv3x3_wgrad_ws(int hw, int cin, int cout) {
  // (the larger of the dense and the sparse geometry of a shape: the caller does not say whether dz is pooled)
#define WS(CI_, CO_, HW_, P_)                          \
  if (cin == CI_ && cout == CO_ && hw == HW_) {        \
    const size_t a = ws_floats_for<CI_, CO_, false>(kMaxJobs), b = P_ ? ws_floats_for<CI_, CO_, UGN_X3_SPARSE != 0>(kMaxJobs) : 0; \
    return (a > b ? a : b) * sizeof(float);            \
  }
  WS(32, 32, 64, 1) WS(32, 64, 32, 0) WS(64, 64, 32, 1) WS(64, 128, 16, 0) WS(128, 128, 16, 0)
#undef WS
  return 0;
}

extern "C" int ugn_x3_conv3x3_wgrad_multi(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx, float* const* dw,
                                          const int* n, int njobs, int hw, int cin, int cout, void* ws, size_t ws_bytes, int products, void* stream) {
  UGN_REQUIRE(in && dz && dw && n && ws, "ugn_x3_conv3x3_wgrad_multi: null array");
  UGN_REQUIRE(products == 6 || products == 9, "ugn_x3_conv3x3_wgrad_multi: products must be 6 (default) or 9 (all partial products), got %d", products);
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_x3_conv3x3_wgrad_multi: njobs must be 1..%d (got %d)", kMaxJobs, njobs);
  const bool pooled = dz_idx != nullptr && dz_idx[0] != nullptr;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(in[j] && dz[j] && dw[j] && n[j] > 0, "ugn_x3_conv3x3_wgrad_multi: null pointer or n <= 0 in job %d", j);
    UGN_REQUIRE(pooled == (dz_idx != nullptr && dz_idx[j] != nullptr), "ugn_x3_conv3x3_wgrad_multi: dz_idx for all jobs or none");
  }
  hipStream_t st = (hipStream_t)stream;
  float* wsf = (float*)ws;
  const size_t wfl = ws_bytes / sizeof(float);
#define WG(CI_, CO_, HW_, P_)                                        \
  if (cin == CI_ && cout == CO_ && hw == HW_ && pooled == (P_ != 0)) \
    return launch_wgrad<CI_, CO_, HW_, P_>(in, dz, dz_idx, dw, n, njobs, wsf, wfl, products, st);
  WG(32, 32, 64, 1) WG(32, 64, 32, 0) WG(64, 64, 32, 1) WG(64, 128, 16, 0) WG(128, 128, 16, 0)
#undef WG
  ugn_set_error("ugn_x3_conv3x3_wgrad_multi: unsupported shape cin=%d cout=%d hw=%d pooled=%d", cin, cout, hw, (int)pooled);
  return UGN_EINVAL;
}
