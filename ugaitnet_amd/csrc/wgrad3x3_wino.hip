// Winograd F(2x2,3x3) weight gradient of the 3x3 convolutions on v_mfma_f32_16x16x4_f32 (exact fp32, 16/36 of the direct
// kernel's matrix FLOPs).  Replaces TF's Conv2DBackpropFilter behind reference nets/mj_uwyhNets_ba.py:431-462.
//
//   dW = G^T [ sum over tiles  (B^T d B) .* (A dY A^T) ] G        d: 4x4 input patch, dY: 2x2 output-gradient tile
//
// GEMM view per Winograd point: Z[pt][ci][co] += V[pt][tile][ci] * Q[pt][tile][co], K = tiles (4 per MFMA).
// A 512-thread workgroup (8 waves, 2 per SIMD; persistent) owns a (32 input channel) x (32|64 output channel) block of
// every point and walks 8x16-pixel regions (32 tiles).  A wave owns 16 ci x 32 co (16 points x 2 channel blocks = 128
// accumulator registers) and a 1/KSPLIT share of each region's tiles.  Both operands are transformed IN REGISTERS by the
// lane that feeds them to the MFMA: lane (ci, tile) reads its 4x4 patch from the fp32 halo tile and applies B^T.B;
// lane (tile, co) reads its 2x2 gradient tile and applies A.A^T -- for a pooled layer the tile is ONE pooled pixel plus its
// argmax, so Q is that value times a sign pattern.  Halo and gradient tiles of the next region stream in by LDS-DMA while
// the current one computes: one barrier per region.  G^T . G is applied lane-locally at the end and the partial sums leave
// as slabs [group][tap][32][co]; `wino_wgrad_finish` sums the slabs in a fixed order (bitwise reproducible, no atomics).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int RH = 8, RW = 16;                      // region: 8 x 16 output pixels = 4 x 8 tiles
constexpr int PW = RW + 2, PHH = RH + 2, NPIX = PHH * PW;   // 10 x 18 halo
constexpr int CS = 32;                               // halo pixel stride (floats): 32 channels, NO padding.  The 4 tiles of an
// MFMA step sit 2 pixels apart, i.e. on the same banks; instead of padding the pixel (which costs LDS-DMA pieces: an LDS-DMA
// instruction is the most expensive thing the loop issues) the 16-byte quads of a pixel are XOR-swizzled by 4 when bit 1 of
// the pixel index is set: neighbouring tiles land 16 banks apart, conflict-free, and every DMA lane carries payload.
constexpr int SPX = CS / 4;                          // 16-byte slots per halo pixel
constexpr int HSLOTS = NPIX * SPX, HPIECES = (HSLOTS + 63) / 64;   // 1440 slots -> 23 pieces of 1 KB
__device__ __forceinline__ constexpr int swz(int p) { return ((p >> 1) & 1) * 4; }   // quad XOR of pixel p
constexpr int SIN = HPIECES * 256;                   // floats per halo buffer

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Packed fp32 arithmetic for the operand transforms (v_pk_add_f32: two sums per full-rate instruction).  A lane owns ONE
// channel of its patch, so the pairs are neighbouring patch columns: the row pass B^T d combines whole pairs and packs, the
// column pass (t B) mixes the halves of its pairs and stays scalar -- the op_sel forms the compiler builds for it measure
// 2.5 % SLOWER than scalar code, and inline asm is not an option next to MFMAs (wino_common.h pk_add).  The gradient
// transform A dY A^T packs over the lane's two channel blocks.  Worth 0.4 % here (the forward / data-gradient kernels, whose
// lanes own channel pairs, gain 1-7 % from the same idea).
#ifndef UGN_WG_PK
#define UGN_WG_PK 1
#endif
// timing-only ablations (WRONG results): 1 no input-transform arithmetic, 2 no gradient-transform arithmetic, 4 no operand
// reads from LDS, 8 no MFMA, 16 no DMA, 32 no operand reads in a region's FIRST step, 64 none in the other steps
#ifndef UGN_WG_ABLATE
#define UGN_WG_ABLATE 0
#endif
typedef float wg_v2f __attribute__((ext_vector_type(2)));

// bf16 operands (template flag BF, shapes with 16 tiles per wave and region): the 4 k-steps of a region become the 4 k-slots
// of ONE v_mfma_f32_16x16x16_bf16 per (point, channel block) -- slot j = the lane's tile of step j on both operands.
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wg_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t wg_pk(float a, float b) {
  wg_bf16x2 r;
  r[0] = (__bf16)a;
  r[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ f32x4 wg_mfma_bf16(uint32_t a01, uint32_t a23, uint32_t b01, uint32_t b23, f32x4 c) {
  const uint2 a = make_uint2(a01, a23), b = make_uint2(b01, b23);
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(wg_bf16x4, a), __builtin_bit_cast(wg_bf16x4, b), c, 0, 0, 0);
}

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst_uniform)
               : "memory");
}

template <int COC, int DZ_UNPOOL>
struct DzCfg {
  // plain : [128 px][COC] floats, quads XOR-swizzled like the halo -> COC/4 16-byte slots per pixel
  // pooled: [32 pooled px][COC values + COC argmax bytes (+ pad)]  -> 20 (COC = 64) / 12 (COC = 32) slots per pooled pixel
  // (strides chosen so that the 4 tiles of an MFMA step fall into different bank halves)
  static constexpr int PX = DZ_UNPOOL ? 32 : 128;
  static constexpr int SPP = DZ_UNPOOL ? (COC == 64 ? 20 : 12) : COC / 4;
  static constexpr int STRIDE = SPP * 4;              // floats per (pooled) pixel
  static constexpr int SLOTS = PX * SPP, PIECES = (SLOTS + 63) / 64;
  static constexpr int SDZ = PIECES * 256;            // floats per buffer
};

// One launch serves up to kWgMaxJobs weight gradients of the same shape (a frame-level layer and its set-level twin of the
// global branch, for every modality).  The regions of all jobs form ONE list; each of the `groups` workgroup groups of a
// (ci, co) block combination owns a contiguous 1/groups share of it, so the work is balanced to one region whatever the
// jobs' sizes.  A group whose share crosses a job boundary writes its partial sums for the finished job (a slab) and starts
// the next job from zero.  Slabs of job j: [combo][ng_j] at job[j].slab, written by the groups g0_j .. g0_j + ng_j - 1.
constexpr int kWgMaxJobs = 6;
struct WgJob {
  const float* in;
  const float* dz;
  const uint8_t* dz_idx;
  float* slab;
  int g0, ng;
};
struct WgJobs {
  WgJob job[kWgMaxJobs];
  int rstart[kWgMaxJobs + 1];   // first region of job j in the list; entries from the job count onwards hold the total
  int groups;
};
__device__ __forceinline__ int wg_job_of(const WgJobs& jt, int r) {
  int jb = 0;
#pragma unroll
  for (int j = 1; j < kWgMaxJobs; ++j) jb += r >= jt.rstart[j] ? 1 : 0;
  return jb;
}

template <int CI, int CO, int HW, int COC, int DZ_UNPOOL, bool BF = false>
__global__ __launch_bounds__(512, 2) void wgrad_wino_kernel(const WgJobs jt, const float* __restrict__ zeros) {
  using D = DzCfg<COC, DZ_UNPOOL>;
  constexpr int COP = COC / 32;                       // 32-channel output pairs per workgroup (1 or 2)
  constexpr int KSPLIT = 8 / (2 * COP);               // waves sharing one output block (4 or 2)
  constexpr int TPW = 32 / KSPLIT, STEPS = TPW / 4;   // tiles per wave per region, MFMA k-steps
  constexpr int RPX = HW / RW, RPY = HW / RH, RPI = RPX * RPY;
  constexpr int NCO = CO / COC;
  constexpr int HPW = (HPIECES + 7) / 8, DPW = (D::PIECES + 7) / 8;   // DMA pieces per wave

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sIn0 = smem;
  float* sDz0 = smem + 2 * SIN;
  const unsigned sin_bytes = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem);
  const unsigned sdz_bytes = sin_bytes + 2u * SIN * 4u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kq = lane >> 4;
  const int cop = wave % COP, cib = (wave / COP) & 1, ks = wave / (2 * COP);
  const int groups = jt.groups, total = jt.rstart[kWgMaxJobs];
  const int combo = (int)blockIdx.x / groups, grp = (int)blockIdx.x % groups;
  const int cic = combo / NCO, coc = combo % NCO;
  // this group's share of the region list (groups <= total: never empty)
  const int r_begin = (int)((long)grp * total / groups), r_end = (int)((long)(grp + 1) * total / groups);

  // per-lane geometry of this wave's DMA pieces (region independent)
  int hgeo[HPW], dgeo[DPW];
#pragma unroll
  for (int j = 0; j < HPW; ++j) {
    int inst = wave * HPW + j;
    inst = inst < HPIECES ? inst : HPIECES - 1;
    int slot = inst * 64 + lane;
    slot = slot < HSLOTS ? slot : HSLOTS - 1;
    const int p = slot / SPX, c4 = (slot - p * SPX) ^ swz(p);   // the quad that belongs at this LDS position
    hgeo[j] = ((p / PW) << 16) | ((p % PW) << 8) | c4;
  }
#pragma unroll
  for (int j = 0; j < DPW; ++j) {
    int inst = wave * DPW + j;
    inst = inst < D::PIECES ? inst : D::PIECES - 1;
    int slot = inst * 64 + lane;
    slot = slot < D::SLOTS ? slot : D::SLOTS - 1;
    const int dp = slot / D::SPP, dq = slot % D::SPP;
    dgeo[j] = (dp << 8) | (DZ_UNPOOL ? dq : (dq ^ swz(dp)));
  }
  auto issue_dma = [&](int job, int gregion, int buf) {   // (ONE table entry is read per call: scalar loads)
    if constexpr ((UGN_WG_ABLATE & 16) != 0) return;
    const float* __restrict__ in = jt.job[job].in;
    const float* __restrict__ dz = jt.job[job].dz;
    const uint8_t* __restrict__ dz_idx = jt.job[job].dz_idx;
    const int region = gregion - jt.rstart[job];
    const int img = region / RPI, rrem = region % RPI;
    const int ry0 = (rrem / RPX) * RH, rx0 = (rrem % RPX) * RW;
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
      int inst = wave * HPW + j;
      inst = inst < HPIECES ? inst : HPIECES - 1;
      const int yy = hgeo[j] >> 16, xx = (hgeo[j] >> 8) & 0xff, c4 = hgeo[j] & 0xff;
      const int gy = ry0 - 1 + yy, gx = rx0 - 1 + xx;
      const bool ok = (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
      const float* src = ok ? in + (((size_t)img * HW + gy) * HW + gx) * CI + cic * 32 + c4 * 4 : zeros;
      dma16(src, sin_bytes + (unsigned)buf * SIN * 4u + (unsigned)inst * 1024u);
    }
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
      int inst = wave * DPW + j;
      inst = inst < D::PIECES ? inst : D::PIECES - 1;
      const int p = dgeo[j] >> 8, q = dgeo[j] & 0xff;
      const void* src;
      if constexpr (DZ_UNPOOL) {
        constexpr int HP = HW / 2;
        const size_t o = (((size_t)img * HP + ry0 / 2 + (p >> 3)) * HP + rx0 / 2 + (p & 7)) * CO + coc * COC;
        src = q < COC / 4 ? (const void*)(dz + o + q * 4)
                          : (q < COC / 4 + COC / 16 ? (const void*)(dz_idx + o + (q - COC / 4) * 16) : (const void*)zeros);
      } else {
        const size_t o = (((size_t)img * HW + ry0 + (p >> 4)) * HW + rx0 + (p & 15)) * CO + coc * COC;
        src = (const void*)(dz + o + q * 4);
      }
      dma16(src, sdz_bytes + (unsigned)buf * D::SDZ * 4u + (unsigned)inst * 1024u);
    }
  };

  f32x4 acc[16][2];
#pragma unroll
  for (int pt = 0; pt < 16; ++pt)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[pt][cb][r] = 0.f;

  int jb = wg_job_of(jt, r_begin);
  issue_dma(jb, r_begin, 0);
  int buf = 0;
  for (int region = r_begin; region < r_end; ++region, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the current buffers have landed
    __syncthreads();                                    // everyone's have; the other buffers have no readers left
    const int next = region + 1;
    const bool has_next = next < r_end;
    const int jn = has_next ? wg_job_of(jt, next) : jb;
    const bool flush = !has_next || jn != jb;            // the accumulators leave after this region (wave-uniform)
    if (!flush) issue_dma(jb, next, buf ^ 1);            // (a flush reuses the tile buffers: the next region is fetched after it)
    const float* sIn = sIn0 + buf * SIN;
    const float* sDz = sDz0 + buf * D::SDZ;
    // raw operands of one MFMA step: the lane's 4x4 input patch and its two 2x2 gradient tiles (or pooled pixel + argmax);
    // the reads of step st+1 are issued before the MFMAs of step st, so their latency hides behind the matrix pipe
    float d[16], yv[2][4];
    unsigned ypos[2];
    auto load_raw = [&](int st) {
      if ((UGN_WG_ABLATE & 4) != 0 || ((UGN_WG_ABLATE & 32) != 0 && st == 0) || ((UGN_WG_ABLATE & 64) != 0 && st != 0)) {
        for (int e = 0; e < 16; ++e) d[e] = (float)(st + e);
        for (int cb = 0; cb < 2; ++cb) { for (int e = 0; e < 4; ++e) yv[cb][e] = (float)(cb + e); ypos[cb] = (unsigned)st; }
        return;
      }
      // this lane's tile of the step: t = ks*TPW + st*4 + kq, row t>>3 (0..3), col t&7.  Split into a lane part (kq, ks) and
      // a compile-time step part so that no address arithmetic is left inside the loop: the tile origin p0 = 2*(18*tr + tc)
      // has swizzle bit (18*tr + tc) & 1 = tc & 1 = kq & 1 -- a lane constant.
      const int tr0 = (ks * TPW) >> 3, tc0 = ((ks * TPW) & 7) + kq;          // step 0 tile (kq < 4, TPW is 8 or 16)
      const int dtr = (st * 4) >> 3, dtc = (st * 4) & 7;                      // compile-time step advance (tc0 + dtc < 8)
      const int cq = cib * 16 + lj, sw = (kq & 1) * 4;
      const float* pl0 = sIn + ((2 * tr0) * PW + 2 * tc0) * CS + (((cq >> 2) ^ sw) << 2) + (cq & 3);   // lane constants
      const float* pl1 = sIn + ((2 * tr0) * PW + 2 * tc0) * CS + (((cq >> 2) ^ sw ^ 4) << 2) + (cq & 3);
      const int pst = ((2 * dtr) * PW + 2 * dtc) * CS;                        // compile-time
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int off = (e >> 2) * PW + (e & 3);
        d[e] = (swz(off) ? pl1 : pl0)[pst + off * CS];
      }
      const int tr = tr0 + dtr, tc = tc0 + dtc;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int co = cop * 32 + cb * 16 + lj;
        if constexpr (DZ_UNPOOL) {
          const float* pp = sDz + (tr * 8 + tc) * D::STRIDE;
          yv[cb][0] = pp[co];
          ypos[cb] = reinterpret_cast<const uint8_t*>(pp + COC)[co];
        } else {
          // pixels q0, q0+1, q0+16, q0+17 (q0 even): the same swizzle bit for all four (bit 1 of 16 and 17 is 0)
          const int q0 = (2 * tr) * RW + 2 * tc;          // swizzle bit of q0 = (16*tr + tc) & 1 = kq & 1 as well
          const float* pp = sDz + q0 * D::STRIDE + (((co >> 2) ^ sw) << 2) + (co & 3);
          yv[cb][0] = pp[0]; yv[cb][1] = pp[D::STRIDE]; yv[cb][2] = pp[RW * D::STRIDE]; yv[cb][3] = pp[(RW + 1) * D::STRIDE];
        }
      }
    };
    // (bf16 form: the same reads, split by operand)
    auto load_d = [&](int st) {
      const int tr0 = (ks * TPW) >> 3, tc0 = ((ks * TPW) & 7) + kq;
      const int dtr = (st * 4) >> 3, dtc = (st * 4) & 7;
      const int cq = cib * 16 + lj, sw = (kq & 1) * 4;
      const float* pl0 = sIn + ((2 * tr0) * PW + 2 * tc0) * CS + (((cq >> 2) ^ sw) << 2) + (cq & 3);
      const float* pl1 = sIn + ((2 * tr0) * PW + 2 * tc0) * CS + (((cq >> 2) ^ sw ^ 4) << 2) + (cq & 3);
      const int pst = ((2 * dtr) * PW + 2 * dtc) * CS;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int off = (e >> 2) * PW + (e & 3);
        d[e] = (swz(off) ? pl1 : pl0)[pst + off * CS];
      }
    };
    auto load_y = [&](int st, int cb) {
      const int tr0 = (ks * TPW) >> 3, tc0 = ((ks * TPW) & 7) + kq;
      const int tr = tr0 + ((st * 4) >> 3), tc = tc0 + ((st * 4) & 7);
      const int sw = (kq & 1) * 4;
      const int co = cop * 32 + cb * 16 + lj;
      if constexpr (DZ_UNPOOL) {
        const float* pp = sDz + (tr * 8 + tc) * D::STRIDE;
        yv[cb][0] = pp[co];
        ypos[cb] = reinterpret_cast<const uint8_t*>(pp + COC)[co];
      } else {
        const int q0 = (2 * tr) * RW + 2 * tc;
        const float* pp = sDz + q0 * D::STRIDE + (((co >> 2) ^ sw) << 2) + (co & 3);
        yv[cb][0] = pp[0]; yv[cb][1] = pp[D::STRIDE]; yv[cb][2] = pp[RW * D::STRIDE]; yv[cb][3] = pp[(RW + 1) * D::STRIDE];
      }
    };
    if constexpr (BF) {
      // (STEPS == 2, the 32-channel workgroup shape: 8 tiles per wave and region fill k-slots 0, 1; slots 2, 3 stay empty)
      // ---- A operand of all 4 steps: V = B^T d B of (tile of step j, input channel cib*16 + lj), packed as it is produced
      uint32_t Vlo[16], Vhi[16];
#pragma unroll
      for (int pt = 0; pt < 16; ++pt) Vhi[pt] = 0u;
      {
        float Ve[16];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          load_d(st);
          float tt[16], V[16];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            tt[0 + c] = d[0 + c] - d[8 + c];
            tt[4 + c] = d[4 + c] + d[8 + c];
            tt[8 + c] = d[8 + c] - d[4 + c];
            tt[12 + c] = d[4 + c] - d[12 + c];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            V[r * 4 + 0] = tt[r * 4 + 0] - tt[r * 4 + 2];
            V[r * 4 + 1] = tt[r * 4 + 1] + tt[r * 4 + 2];
            V[r * 4 + 2] = tt[r * 4 + 2] - tt[r * 4 + 1];
            V[r * 4 + 3] = tt[r * 4 + 1] - tt[r * 4 + 3];
          }
#pragma unroll
          for (int pt = 0; pt < 16; ++pt) {
            if (st == 1) Vlo[pt] = wg_pk(Ve[pt], V[pt]);
            else if (st == 3) Vhi[pt] = wg_pk(Ve[pt], V[pt]);
            else Ve[pt] = V[pt];
          }
        }
      }
      // ---- per channel block: B operand of all 4 steps (Q = A dY A^T), then the block's 16 MFMAs
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        uint32_t Qlo[16], Qhi[16];
        float Qe[16];
#pragma unroll
        for (int pt = 0; pt < 16; ++pt) Qhi[pt] = 0u;
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          load_y(st, cb);
          float Q[16];
          if constexpr (DZ_UNPOOL) {
            const float v = yv[cb][0];
            const bool ay = (ypos[cb] >> 1) != 0, ax = (ypos[cb] & 1) != 0;
            float ty[4];
            ty[0] = ay ? 0.f : v; ty[1] = v; ty[2] = ay ? -v : v; ty[3] = ay ? -v : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              Q[r * 4 + 0] = ax ? 0.f : ty[r];
              Q[r * 4 + 1] = ty[r];
              Q[r * 4 + 2] = ax ? -ty[r] : ty[r];
              Q[r * 4 + 3] = ax ? -ty[r] : 0.f;
            }
          } else {
            const float y00 = yv[cb][0], y01 = yv[cb][1], y10 = yv[cb][2], y11 = yv[cb][3];
            float q[4][2];   // (row 3 / column 3 keep the sign convention of the fp32 form: see below)
            q[0][0] = y00; q[0][1] = y01;
            q[1][0] = y00 + y10; q[1][1] = y01 + y11;
            q[2][0] = y00 - y10; q[2][1] = y01 - y11;
            q[3][0] = y10; q[3][1] = y11;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              Q[r * 4 + 0] = q[r][0];
              Q[r * 4 + 1] = q[r][0] + q[r][1];
              Q[r * 4 + 2] = q[r][0] - q[r][1];
              Q[r * 4 + 3] = q[r][1];
            }
          }
#pragma unroll
          for (int pt = 0; pt < 16; ++pt) {
            if (st == 1) Qlo[pt] = wg_pk(Qe[pt], Q[pt]);
            else if (st == 3) Qhi[pt] = wg_pk(Qe[pt], Q[pt]);
            else Qe[pt] = Q[pt];
          }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int pt = 0; pt < 16; ++pt) acc[pt][cb] = wg_mfma_bf16(Vlo[pt], Vhi[pt], Qlo[pt], Qhi[pt], acc[pt][cb]);
        __builtin_amdgcn_s_setprio(0);
      }
    } else {
    load_raw(0);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      // ---- A operand: V = B^T d B of (tile, input channel cib*16 + lj)
      float V[16], Q[2][16];
      if constexpr ((UGN_WG_ABLATE & 1) != 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) V[e] = d[e];
      } else if constexpr (UGN_WG_PK && !DZ_UNPOOL) {   // (the pooled shapes measure 0.5-1.5 % slower with it)
        wg_v2f tp[4][2];   // [row of t][column pair]
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const wg_v2f d0 = {d[0 + 2 * cp], d[1 + 2 * cp]}, d1 = {d[4 + 2 * cp], d[5 + 2 * cp]};
          const wg_v2f d2 = {d[8 + 2 * cp], d[9 + 2 * cp]}, d3 = {d[12 + 2 * cp], d[13 + 2 * cp]};
          tp[0][cp] = d0 - d2;
          tp[1][cp] = d1 + d2;
          tp[2][cp] = d2 - d1;
          tp[3][cp] = d1 - d3;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          V[r * 4 + 0] = tp[r][0].x - tp[r][1].x;
          V[r * 4 + 1] = tp[r][0].y + tp[r][1].x;
          V[r * 4 + 2] = tp[r][1].x - tp[r][0].y;
          V[r * 4 + 3] = tp[r][0].y - tp[r][1].y;
        }
      } else {
        float tt[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          tt[0 + c] = d[0 + c] - d[8 + c];
          tt[4 + c] = d[4 + c] + d[8 + c];
          tt[8 + c] = d[8 + c] - d[4 + c];
          tt[12 + c] = d[4 + c] - d[12 + c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          V[r * 4 + 0] = tt[r * 4 + 0] - tt[r * 4 + 2];
          V[r * 4 + 1] = tt[r * 4 + 1] + tt[r * 4 + 2];
          V[r * 4 + 2] = tt[r * 4 + 2] - tt[r * 4 + 1];
          V[r * 4 + 3] = tt[r * 4 + 1] - tt[r * 4 + 3];
        }
      }
      // ---- B operand: Q = A dY A^T of (tile, output channel cop*32 + cb*16 + lj)
      if constexpr ((UGN_WG_ABLATE & 2) != 0) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int e = 0; e < 16; ++e) Q[cb][e] = yv[cb][DZ_UNPOOL ? 0 : (e & 3)];
      } else if constexpr (!DZ_UNPOOL && UGN_WG_PK) {
        // (packed over the lane's two channel blocks; sign convention of rows / columns 3 as in the scalar form below)
        const wg_v2f y00 = {yv[0][0], yv[1][0]}, y01 = {yv[0][1], yv[1][1]}, y10 = {yv[0][2], yv[1][2]}, y11 = {yv[0][3], yv[1][3]};
        wg_v2f q[4][2];
        q[0][0] = y00; q[0][1] = y01;
        q[1][0] = y00 + y10; q[1][1] = y01 + y11;
        q[2][0] = y00 - y10; q[2][1] = y01 - y11;
        q[3][0] = y10; q[3][1] = y11;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const wg_v2f s = q[r][0] + q[r][1], t = q[r][0] - q[r][1];
          Q[0][r * 4 + 0] = q[r][0].x; Q[1][r * 4 + 0] = q[r][0].y;
          Q[0][r * 4 + 1] = s.x; Q[1][r * 4 + 1] = s.y;
          Q[0][r * 4 + 2] = t.x; Q[1][r * 4 + 2] = t.y;
          Q[0][r * 4 + 3] = q[r][1].x; Q[1][r * 4 + 3] = q[r][1].y;
        }
      } else
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        if constexpr (DZ_UNPOOL) {
          // dY has one non-zero, v at (pos>>1, pos&1): Q = v * a_y (x) a_x with a_0 = (1,1,1,0), a_1 = (0,1,-1,-1)
          const float v = yv[cb][0];
          const bool ay = (ypos[cb] >> 1) != 0, ax = (ypos[cb] & 1) != 0;
          float ty[4];
          ty[0] = ay ? 0.f : v; ty[1] = v; ty[2] = ay ? -v : v; ty[3] = ay ? -v : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Q[cb][r * 4 + 0] = ax ? 0.f : ty[r];
            Q[cb][r * 4 + 1] = ty[r];
            Q[cb][r * 4 + 2] = ax ? -ty[r] : ty[r];
            Q[cb][r * 4 + 3] = ax ? -ty[r] : 0.f;
          }
        } else {
          const float y00 = yv[cb][0], y01 = yv[cb][1], y10 = yv[cb][2], y11 = yv[cb][3];
          // A dY A^T has rows (y0, y0+y1, y0-y1, -y1) and the same pattern in the columns.  The two negations are NOT
          // materialised (an MFMA operand takes no sign modifier: 8 v_xor per step): points of row 3 / column 3 accumulate
          // with the opposite sign and the final G^T Z G subtracts them instead of adding (see the epilogue).
          float q[4][2];
          q[0][0] = y00; q[0][1] = y01;
          q[1][0] = y00 + y10; q[1][1] = y01 + y11;
          q[2][0] = y00 - y10; q[2][1] = y01 - y11;
          q[3][0] = y10; q[3][1] = y11;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Q[cb][r * 4 + 0] = q[r][0];
            Q[cb][r * 4 + 1] = q[r][0] + q[r][1];
            Q[cb][r * 4 + 2] = q[r][0] - q[r][1];
            Q[cb][r * 4 + 3] = q[r][1];
          }
        }
      }
      if (st + 1 < STEPS) load_raw(st + 1);
      __builtin_amdgcn_s_setprio(1);   // keep the matrix pipe while the SIMD's other wave transforms its operands
#pragma unroll
      for (int pt = 0; pt < 16; ++pt) {
        if constexpr ((UGN_WG_ABLATE & 8) != 0) {
          acc[pt][0][0] += V[pt] * Q[0][pt];
          acc[pt][1][0] += V[pt] * Q[1][pt];
        } else {
          acc[pt][0] = mfma16(V[pt], Q[0][pt], acc[pt][0]);
          acc[pt][1] = mfma16(V[pt], Q[1][pt], acc[pt][1]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    }   // !BF
    if (!flush) continue;

  // ---- job (or share) finished: combine the KSPLIT waves that share an output block (through LDS, fixed order), write the slab
  __syncthreads();
  float* sRed = smem;   // 16 points x 4 registers x 64 lanes = 16 KB per (wave, channel block)
#pragma unroll
  for (int src = 1; src < KSPLIT; ++src) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      __syncthreads();
      if (ks == src) {
        float* dst = sRed + (cib * COP + cop) * 4096;
#pragma unroll
        for (int pt = 0; pt < 16; ++pt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(pt * 4 + r) * 64 + lane] = acc[pt][cb][r];
      }
      __syncthreads();
      if (ks == 0) {
        const float* s2 = sRed + (cib * COP + cop) * 4096;
#pragma unroll
        for (int pt = 0; pt < 16; ++pt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[pt][cb][r] += s2[(pt * 4 + r) * 64 + lane];
      }
    }
  }
  if (ks == 0) {
    // G^T Z G is lane-local (a lane holds all 16 points of its (ci, co) elements): the slab carries the 9 taps
    float* dst = jt.job[jb].slab + ((size_t)combo * jt.job[jb].ng + (grp - jt.job[jb].g0)) * 9 * 32 * COC;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float z0 = acc[0 + c][cb][r], z1 = acc[4 + c][cb][r], z2 = acc[8 + c][cb][r], z3 = acc[12 + c][cb][r];
          t[0][c] = z0 + 0.5f * (z1 + z2);
          t[1][c] = 0.5f * (z1 - z2);
          t[2][c] = DZ_UNPOOL ? 0.5f * (z1 + z2) + z3 : 0.5f * (z1 + z2) - z3;   // plain dz: row 3 was accumulated negated
        }
        float* o = dst + ((size_t)cib * 16 + 4 * kq + r) * COC + cop * 32 + cb * 16 + lj;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          o[(size_t)(a * 3 + 0) * 32 * COC] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
          o[(size_t)(a * 3 + 1) * 32 * COC] = 0.5f * (t[a][1] - t[a][2]);
          o[(size_t)(a * 3 + 2) * 32 * COC] = DZ_UNPOOL ? 0.5f * (t[a][1] + t[a][2]) + t[a][3] : 0.5f * (t[a][1] + t[a][2]) - t[a][3];
        }
      }
  }
    if (!has_next) break;
    // ---- the share continues with the next job: zero sums, refill the tile buffer the combine overwrote
#pragma unroll
    for (int pt = 0; pt < 16; ++pt)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[pt][cb][r] = 0.f;
    jb = jn;
    __syncthreads();                     // the combine's last LDS reads are done before the DMA writes there
    issue_dma(jb, next, buf ^ 1);
  }
}

// dW[tap][cic*32 + ci][coc*COC + co] = sum_g slab[combo][g][tap][ci][co], in float4 units over co.
// 256 threads = 16 elements x 16 slab lanes; lane gl sums groups gl, gl+16, ..., then the 16 lanes are added in order.
// (blockIdx.z = job: both weight gradients of a pair launch are finished by one launch)
struct WgFinish {
  const float4* slab[kWgMaxJobs];
  float* dw[kWgMaxJobs];
  int ng[kWgMaxJobs];
};
template <int COC>
__global__ __launch_bounds__(256) void wino_wgrad_finish(const WgFinish ft, int CI, int CO) {
  __shared__ float4 sR[16][16];
  const float4* __restrict__ slab = ft.slab[blockIdx.z];
  float* __restrict__ dw = ft.dw[blockIdx.z];
  const int groups = ft.ng[blockIdx.z];
  constexpr int E4 = 9 * 32 * COC / 4;
  const int le = threadIdx.x & 15, lg = threadIdx.x >> 4;
  const int combo = blockIdx.y, e = blockIdx.x * 16 + le;   // E4 is a multiple of 16
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = lg; g < groups; g += 16) {
    const float4 v = slab[((size_t)combo * groups + g) * E4 + e];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  sR[lg][le] = s;
  __syncthreads();
  if (lg != 0) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    const float4 v = sR[k][le];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int nco = CO / COC, cic = combo / nco, coc = combo % nco;
  const int co4 = e % (COC / 4), ci = (e / (COC / 4)) % 32, tap = e / (COC / 4 * 32);
  *reinterpret_cast<float4*>(dw + ((size_t)tap * CI + cic * 32 + ci) * CO + coc * COC + co4 * 4) = s;
}

inline const float* zero_block_w() {
  static float* z = nullptr;
  if (!z) {
    float* p = nullptr;
    if (hipMalloc((void**)&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) return nullptr;
    z = p;
  }
  return z;
}

constexpr int kWgs = 256;

struct WgHostJob {
  const float* in;
  const float* dz;
  const uint8_t* dz_idx;
  float* dw;
  int n;
};

template <int CI, int CO, int HW, int COC, int DZ_UNPOOL, bool BF = false>
int launch_wgrad_wino(const WgHostJob* hj, int njobs, float* ws, size_t ws_floats, hipStream_t st) {
  using D = DzCfg<COC, DZ_UNPOOL>;
  constexpr int LDS = (2 * SIN + 2 * D::SDZ) * 4 > 8 * 16384 ? (2 * SIN + 2 * D::SDZ) * 4 : 8 * 16384;
  auto kern = wgrad_wino_kernel<CI, CO, HW, COC, DZ_UNPOOL, BF>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { ugn_set_error("wgrad wino: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const float* zeros = zero_block_w();
  if (!zeros) { ugn_set_error("wgrad wino: cannot allocate the zero block"); return UGN_EINVAL; }
  constexpr int NCOMBO = (CI / 32) * (CO / COC);
  WgJobs jt;
  long total = 0;
  for (int j = 0; j < kWgMaxJobs; ++j) {
    jt.rstart[j] = (int)total;
    if (j < njobs) total += (long)hj[j].n * (HW / RH) * (HW / RW);
  }
  jt.rstart[kWgMaxJobs] = (int)total;
  if (total > 0x3fffffff) { ugn_set_error("wgrad wino: too many regions (%ld)", total); return UGN_EINVAL; }
  const int groups = (long)(kWgs / NCOMBO) > total ? (int)total : kWgs / NCOMBO;
  jt.groups = groups;
  // which groups touch which job: group g owns regions [g * total / groups, (g + 1) * total / groups)
  const size_t slab_floats = (size_t)9 * 32 * COC;
  WgFinish ft = {};
  float* slab = ws;
  size_t need = 0;
  for (int j = 0; j < kWgMaxJobs; ++j) {
    const int jj = j < njobs ? j : njobs - 1;
    jt.job[j] = {hj[jj].in, hj[jj].dz, hj[jj].dz_idx, nullptr, 0, 0};
    if (j >= njobs) continue;
    int g0 = -1, g1 = -1;
    for (int g = 0; g < groups; ++g) {
      const long b0 = (long)g * total / groups, b1 = (long)(g + 1) * total / groups;
      if (b0 < jt.rstart[j + 1] && b1 > jt.rstart[j]) { if (g0 < 0) g0 = g; g1 = g; }
    }
    const int ng = g1 - g0 + 1;
    need += (size_t)NCOMBO * ng * slab_floats;
    if (ws_floats < need) {
      ugn_set_error("wgrad wino: workspace too small (%zu < %zu floats)", ws_floats, need);
      return UGN_EINVAL;
    }
    jt.job[j].slab = slab;
    jt.job[j].g0 = g0;
    jt.job[j].ng = ng;
    ft.slab[j] = (const float4*)slab;
    ft.dw[j] = hj[j].dw;
    ft.ng[j] = ng;
    slab += (size_t)NCOMBO * ng * slab_floats;
  }
  hipLaunchKernelGGL(kern, dim3(NCOMBO * groups), dim3(512), LDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("wgrad wino");
  hipLaunchKernelGGL(wino_wgrad_finish<COC>, dim3(9 * 32 * COC / 4 / 16, NCOMBO, njobs), dim3(256), 0, st, ft, CI, CO);
  UGN_CHECK_LAUNCH("wgrad wino finish");
  return 0;
}

bool wg_cfg(int hw, int cin, int cout, int* coc) {
  if ((hw == 64 && cin == 32 && cout == 32) || (hw == 32 && cin == 32 && cout == 64) || (hw == 32 && cin == 64 && cout == 64) ||
      (hw == 16 && cin == 64 && cout == 128) || (hw == 16 && cin == 128 && cout == 128)) {
    *coc = cout >= 64 ? 64 : 32;
    return true;
  }
  return false;
}

int dispatch_wgrad(const WgHostJob* hj, int njobs, int hw, int cin, int cout, void* ws, size_t ws_bytes, bool bf, hipStream_t st) {
  const int unpool = hj[0].dz_idx != nullptr;
  if (bf) { ugn_set_error("the bf16-operand Winograd kernels (fp32 tensors, 'bf16w') were retired in round 5: conv_precision='bf16' is the configs[4] path"); return UGN_EINVAL; }
#define WGW(CI_, CO_, HW_, COC_, U_)                          \
  if (cin == CI_ && cout == CO_ && hw == HW_ && unpool == U_) \
    return launch_wgrad_wino<CI_, CO_, HW_, COC_, U_>(hj, njobs, (float*)ws, ws_bytes / sizeof(float), st);
  WGW(32, 32, 64, 32, 1)    // a2
  WGW(32, 64, 32, 64, 0)    // a3, b1
  WGW(64, 64, 32, 64, 1)    // a4, b2
  WGW(64, 128, 16, 64, 0)   // a5, b3
  WGW(128, 128, 16, 64, 0)  // a6, b4
#undef WGW
  ugn_set_error("ugn_conv3x3_wgrad_wino: unsupported shape cin=%d cout=%d hw=%d unpool=%d", cin, cout, hw, unpool);
  return UGN_EINVAL;
}

}  // namespace

extern "C" size_t ugn_conv3x3_wgrad_wino_ws(int n, int hw, int cin, int cout) {
  // upper bound for one job of n images, and for a pair whose image counts sum to n
  int coc;
  if (!wg_cfg(hw, cin, cout, &coc) || n <= 0) return 0;
  // every group writes one slab per job it touches: at most 256 + (jobs - 1) per block combination
  return (size_t)(kWgs + 8 * (kWgMaxJobs - 1)) * 9 * 32 * coc * sizeof(float);
}

static int wgrad_one(const float* in, const float* dz, const uint8_t* dz_idx, float* dw, int n, int hw, int cin, int cout,
                     void* ws, size_t ws_bytes, bool bf, void* stream) {
  UGN_REQUIRE(in && dz && dw && ws && n > 0, "ugn_conv3x3_wgrad_wino: null pointer or n <= 0");
  const WgHostJob job = {in, dz, dz_idx, dw, n};
  return dispatch_wgrad(&job, 1, hw, cin, cout, ws, ws_bytes, bf, (hipStream_t)stream);
}
extern "C" int ugn_conv3x3_wgrad_wino(const float* in, const float* dz, const uint8_t* dz_idx, float* dw, int n, int hw,
                                      int cin, int cout, void* ws, size_t ws_bytes, void* stream) {
  return wgrad_one(in, dz, dz_idx, dw, n, hw, cin, cout, ws, ws_bytes, false, stream);
}

static int wgrad_multi(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx, float* const* dw,
                       const int* n, int njobs, int hw, int cin, int cout, void* ws, size_t ws_bytes, bool bf, void* stream) {
  UGN_REQUIRE(in && dz && dw && n && ws, "ugn_conv3x3_wgrad_wino_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kWgMaxJobs, "ugn_conv3x3_wgrad_wino_multi: njobs must be 1..%d (got %d)", kWgMaxJobs, njobs);
  WgHostJob jobs[kWgMaxJobs];
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(in[j] && dz[j] && dw[j] && n[j] > 0, "ugn_conv3x3_wgrad_wino_multi: null pointer or n <= 0 in job %d", j);
    jobs[j] = {in[j], dz[j], dz_idx ? dz_idx[j] : nullptr, dw[j], n[j]};
    UGN_REQUIRE((jobs[0].dz_idx != nullptr) == (jobs[j].dz_idx != nullptr), "ugn_conv3x3_wgrad_wino_multi: dz_idx for all jobs or none");
  }
  return dispatch_wgrad(jobs, njobs, hw, cin, cout, ws, ws_bytes, bf, (hipStream_t)stream);
}
static int wgrad_pair(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx, float* const* dw,
                      const int* n, int hw, int cin, int cout, void* ws, size_t ws_bytes, bool bf, void* stream) {
  return wgrad_multi(in, dz, dz_idx, dw, n, 2, hw, cin, cout, ws, ws_bytes, bf, stream);
}
extern "C" int ugn_conv3x3_wgrad_wino_multi(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx,
                                            float* const* dw, const int* n, int njobs, int hw, int cin, int cout, void* ws,
                                            size_t ws_bytes, int bf16, void* stream) {
  return wgrad_multi(in, dz, dz_idx, dw, n, njobs, hw, cin, cout, ws, ws_bytes, bf16 != 0, stream);
}
extern "C" int ugn_conv3x3_wgrad_wino_pair(const float* const* in, const float* const* dz, const uint8_t* const* dz_idx,
                                           float* const* dw, const int* n, int hw, int cin, int cout, void* ws,
                                           size_t ws_bytes, void* stream) {
  return wgrad_pair(in, dz, dz_idx, dw, n, hw, cin, cout, ws, ws_bytes, false, stream);
}
