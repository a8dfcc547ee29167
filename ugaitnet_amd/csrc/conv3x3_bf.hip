// Direct 3x3 convolution (forward / data gradient) of bf16 tensors on v_mfma_f32_32x32x16_bf16, fp32 accumulate: BASELINE.json
// configs[4] ("3-modality bf16 path: MFMA bf16 conv tiles + fp32 accumulate"), SURVEY 8(d) "C5": bf16 activations, gradients and
// saved tensors in HBM, fp32 master weights and Adam.  Same operator contract and the same structure as conv3x3_mm.hip (the
// fp32-class H2 kernels; read its header first): 512-thread persistent workgroups own 16x16-pixel regions and all N output
// channels, wave = one 32-pixel MFMA block in pool order, LDS-DMA for input tiles (waves 4..7) and filter stages (waves 0..3).
// What differs: ONE plane and ONE MFMA per (tap, k-step, block) instead of two and three; a pixel's record is 2 bytes per
// channel, so a K chunk is 64 channels (128-byte records, 8 + 1 slots per pixel, 168 per row) where the layer has them and
// the input is not pooled, else 32 channels (64-byte records, 4 + 1 slots per pixel, 104 per row: also conflict-free for
// ds_read_b128 -- pixel stride 5 is odd, row stride = 8 mod 16); no block exponents (bf16 has fp32's range).
// Reference lines replaced: nets/mj_uwyhNets_ba.py:431-462 (Conv2D + LeakyReLU (+ MaxPool) and their input gradients).
#include <stdlib.h>
#include "mm_common.h"

using namespace ugn_mm;

#ifndef UGN_BF_XCD
#define UGN_BF_XCD 1
#endif
#ifndef UGN_BF_RESW
#define UGN_BF_RESW 1      /* 0: the staged filter ring of rounds 1-5 everywhere (A/B) */
#endif

namespace {

enum { EPI_LRELU = 0, EPI_LRELU_POOL = 1, EPI_DGRAD = 2, EPI_DGRAD_ACT = 3 };
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <int KC, int NC, int IN_POOLED>
struct BGeo {
  static constexpr int RS = (KC % 64 == 0 && !IN_POOLED) ? 8 : 4;      // 16-byte slots of a pixel record (one K chunk)
  static constexpr int CH = RS * 8;                                    // channels per chunk
  static constexpr int KSTEPS = CH / 16;
  static constexpr int PS = RS + 1;                                    // slots per pixel in LDS (1 pad)
  static constexpr int HROW = RS == 8 ? 168 : 104;                     // slots per halo row (18 * PS + pad; = 8 mod 16)
  static constexpr int HPIECES = RS == 8 ? 48 : 30;                    // ceil(18 * HROW / 64)
  static constexpr int HALO_BYTES = HPIECES * 1024;
  static constexpr int W_OFF = 2 * HALO_BYTES;
  static constexpr int NCHUNK = KC / CH;
  static constexpr int NB = NC / 32;
  static constexpr int TPS = NB == 4 ? 1 : 3;
  static constexpr int NSTG = 9 / TPS;
  static constexpr int WPIECES = TPS * KSTEPS * NB;                    // [tap][k-step][block][1 KB]
  static constexpr int WSTAGE = WPIECES * 1024;
  static constexpr int SPP = RS + RS / 2;                              // staging slots per pooled pixel: values + argmax bytes
  static constexpr int STG_PIECES = (100 * SPP + 63) / 64;
  static constexpr int STG_BYTES = STG_PIECES * 1024;
  static constexpr int HPW = (HPIECES + 3) / 4;                        // halo pieces per fetching wave
  // RESW (round 6): the job's WHOLE packed filter (all chunks, all taps: 9 * KC * NC * 2 bytes) stays in LDS for as long as the
  // workgroup works on that job, where it fits beside the two halo buffers.  The layers this covers (32 -> 32, 32 <-> 64 and the
  // pooled 64 -> 64 data gradient: 18-72 KB of filter) multiply only 6-24 MFMAs per stage and wave, so the per-stage filter DMA with
  // its wait + barrier was an exposed memory latency three times per tile: the matrix pipe was busy 7 % of a 32 -> 32 item.
  static constexpr int WALL = NCHUNK * NSTG * WSTAGE;                  // the whole filter
  static constexpr bool RESW = UGN_BF_RESW && (W_OFF + WALL + (IN_POOLED ? STG_BYTES : 0)) <= 147456;
  static constexpr int WBYTES = RESW ? WALL : 2 * WSTAGE;
  static constexpr int LDS = W_OFF + WBYTES + (IN_POOLED ? STG_BYTES : 0);
};

constexpr int kPackJobs = 64;
struct BPackTable {
  const float* w[kPackJobs];
  uint16_t* pk[kPackJobs];
  int cin[kPackJobs], cout[kPackJobs], dgrad[kPackJobs], ch[kPackJobs];
};
// element e of job j -> bf16 at [chunk][tap][k-step][block][h][col][8]   (forward g = w; data gradient g[tap][k = cout][n = cin] = w[8 - tap])
__global__ void bf_pack_kernel(BPackTable t) {
  const int j = blockIdx.y;
  const int cin = t.cin[j], cout = t.cout[j], dgrad = t.dgrad[j], CH = t.ch[j];
  const int kc = dgrad ? cout : cin, nc = dgrad ? cin : cout;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 9 * kc * nc) return;
  const int tap = e / (kc * nc), rem = e - tap * (kc * nc);
  const int k = dgrad ? rem % kc : rem / nc, n = dgrad ? rem / kc : rem % nc;
  const float v = dgrad ? t.w[j][((size_t)(8 - tap) * cin + n) * cout + k] : t.w[j][((size_t)tap * cin + k) * cout + n];
  const int nb_all = nc / 32, ksteps = CH / 16;
  const int chunk = k / CH, kk = k - chunk * CH, s = kk >> 4, h = (kk >> 3) & 1, ee = kk & 7;
  const int nb = mm_block_of(n, nc), col = mm_col_of(n, nc);
  const size_t base = (((size_t)chunk * 9 + tap) * ksteps + s) * nb_all + nb;
  t.pk[j][base * 512 + (h * 32 + col) * 8 + ee] = __builtin_bit_cast(unsigned short, (__bf16)v);
}

template <typename G, int KC, int HW>
__device__ __forceinline__ void dma_halo_piece(const char* __restrict__ img_base, const void* __restrict__ zeros, int ry0, int rx0,
                                               int chunk, int piece, int lane, unsigned lds_byte_base) {
  const int g = piece * 64 + lane;
  const int row = G::RS == 8 ? (g * 6242) >> 20 : (g * 10083) >> 20;        // g / 168 (g < 3072), g / 104 (g < 1920)
  const int rem = g - row * G::HROW;
  const int px = G::RS == 8 ? (rem * 57) >> 9 : (rem * 205) >> 10;          // rem / 9, rem / 5
  const int c = rem - px * G::PS;
  const int gy = ry0 - 1 + row, gx = rx0 - 1 + px;
  const bool ok = rem < 18 * G::PS && c < G::RS && row < 18 && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
  const unsigned off = (unsigned)(gy * HW + gx) * (unsigned)(KC * 2) + (unsigned)(chunk * G::RS * 16 + c * 16);
  dma16(ok ? (const void*)(img_base + off) : zeros, lds_byte_base + (unsigned)piece * 1024u);
}

// pooled staging tile (RS = 4: 32 channels): 10 x 10 pooled pixels x [values 4 slots | argmax bytes 2 slots]
template <typename G, int KC, int HW>
__device__ __forceinline__ void dma_pooled_piece(const char* __restrict__ dz_img, const char* __restrict__ idx_img,
                                                 const void* __restrict__ zeros, int ry0, int rx0, int chunk, int piece, int lane,
                                                 unsigned lds_byte_base) {
  constexpr int HP = HW / 2, SPP = G::SPP;
  static_assert(SPP == 6, "pooled inputs use 32-channel chunks");
  const int g = piece * 64 + lane;
  const int pp = (g * 683) >> 12;                    // g / 6 for g < 1024
  const int part = g - pp * 6;
  const int prow = (pp * 205) >> 11, pcol = pp - prow * 10;
  const int pr = ry0 / 2 - 1 + prow, pc = rx0 / 2 - 1 + pcol;
  const bool ok = pp < 100 && (unsigned)pr < (unsigned)HP && (unsigned)pc < (unsigned)HP;
  const unsigned o = (unsigned)(pr * HP + pc);
  const char* vsrc = dz_img + o * (unsigned)(KC * 2) + (unsigned)(chunk * 64 + part * 16);
  const char* isrc = idx_img + o * (unsigned)KC + (unsigned)(chunk * 32 + (part - 4) * 16);
  dma16(!ok ? zeros : (part < 4 ? (const void*)vsrc : (const void*)isrc), lds_byte_base + (unsigned)piece * 1024u);
}

template <typename G>
__device__ __forceinline__ void scatter_pooled(const char* stg, char* halo, int u) {
  if (u >= 400) return;
  const int pp = u >> 2, cg = u & 3;
  const int prow = (pp * 205) >> 11, pcol = pp - prow * 10;
  const uint4 va = *reinterpret_cast<const uint4*>(stg + pp * 96 + cg * 16);
  const uint2 ix = *reinterpret_cast<const uint2*>(stg + pp * 96 + 64 + cg * 8);
  const unsigned vv[4] = {va.x, va.y, va.z, va.w};
#pragma unroll
  for (int pos = 0; pos < 4; ++pos) {
    const int hy = 2 * prow - 1 + (pos >> 1), hx = 2 * pcol - 1 + (pos & 1);
    if ((unsigned)hy >= 18u || (unsigned)hx >= 18u) continue;
    unsigned m[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned w = d < 2 ? ix.x : ix.y;
      const unsigned b0 = (w >> (16 * (d & 1))) & 0xffu, b1 = (w >> (16 * (d & 1) + 8)) & 0xffu;
      m[d] = (b0 == (unsigned)pos ? 0x0000ffffu : 0u) | (b1 == (unsigned)pos ? 0xffff0000u : 0u);
    }
    *reinterpret_cast<uint4*>(halo + (hy * G::HROW + hx * G::PS) * 16 + cg * 16) =
        make_uint4(vv[0] & m[0], vv[1] & m[1], vv[2] & m[2], vv[3] & m[3]);
  }
}

__device__ __forceinline__ f32x16 mfma_b(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned bf_pack(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}

template <int KC, int NC, int HW, int IN_POOLED, int EPI>
__global__ __launch_bounds__(512, 2) void conv_bf_kernel(const MmJobs jt, const void* __restrict__ zeros) {
  using G = BGeo<KC, NC, IN_POOLED>;
  constexpr int NB = G::NB, TPS = G::TPS, NSTG = G::NSTG, WSTAGE = G::WSTAGE, WPIECES = G::WPIECES, NCHUNK = G::NCHUNK;
  constexpr int KSTEPS = G::KSTEPS, W_OFF = G::W_OFF, HALO_BYTES = G::HALO_BYTES;
  constexpr bool RESW = G::RESW;
  constexpr int STG_OFF = W_OFF + G::WBYTES;                 // the pooled staging tile sits behind the filter area
  constexpr int RPX = HW / 16, RPI = RPX * RPX;
  static_assert(!IN_POOLED || NSTG >= 2, "the pooled scatter runs in the last stage of a chunk");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, win = r >> 2, q = r & 3;
  const int a_lane = ((2 * wave + (q >> 1)) * G::HROW + (2 * win + (q & 1)) * G::PS) * 16 + h * 16;
  const int b_lane = W_OFF + lane * 16;

  // (XCD-aware first item, as conv3x3_mm.hip: an XCD's workgroups take a contiguous range of a round's items, so the regions of an
  //  image -- overlapping halos -- are in flight on ONE XCD and the overlap comes from its L2)
  int item = (UGN_BF_XCD && (gridDim.x & 7) == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int nitems = jt.start[kMaxJobs];
  if (item >= nitems) return;
  int jb = mm_job_of(jt, item), lit = item - jt.start[jb];
  auto img_in = [&](const MmJob& J, int img) {
    constexpr size_t IMG = IN_POOLED ? (size_t)(HW / 2) * (HW / 2) * KC : (size_t)HW * HW * KC;
    return reinterpret_cast<const char*>(J.in) + (size_t)img * IMG * 2;
  };
  auto img_idx = [&](const MmJob& J, int img) { return reinterpret_cast<const char*>(J.in_idx) + (size_t)img * (HW / 2) * (HW / 2) * KC; };
  const bool is_hw = wave >= 4;      // DMA roles as in conv3x3_mm.hip: waves 4..7 input tiles, waves 0..3 filter stages
  const int rw = wave & 3;
  constexpr int HSTG = NSTG == 3 ? 2 : 6;
  constexpr int HPER = (G::HPW + HSTG - 1) / HSTG;
  auto stage_in = [&](const MmJob& J, int lit_, int chunk, unsigned halo_dst, int first) {
    const int img = lit_ / RPI, rrem = lit_ % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    if constexpr (IN_POOLED) {
      const char* vb = img_in(J, img);
      const char* ib = img_idx(J, img);
#pragma unroll
      for (int j = 0; j < (G::STG_PIECES + 3) / 4; ++j) {
        const int p = rw + 4 * j;
        if (p < G::STG_PIECES) dma_pooled_piece<G, KC, HW>(vb, ib, zeros, ry0, rx0, chunk, p, lane, sbase + STG_OFF);
      }
    } else {
      const char* vb = img_in(J, img);
#pragma unroll
      for (int j = 0; j < HPER; ++j) {
        const int p = rw + 4 * (first + j);
        if (first + j < G::HPW && p < G::HPIECES) dma_halo_piece<G, KC, HW>(vb, zeros, ry0, rx0, chunk, p, lane, halo_dst);
      }
    }
  };
  auto stage_w = [&](const uint16_t* wpk, int st, unsigned dst) {
    const char* src = reinterpret_cast<const char*>(wpk) + (size_t)st * WSTAGE;
#pragma unroll
    for (int j = 0; j < (WPIECES + 3) / 4; ++j) {
      const int p = rw + 4 * j;
      if (p < WPIECES) dma16(src + p * 1024 + lane * 16, dst + (unsigned)p * 1024u);
    }
  };

  // RESW: every piece of the job's packed filter, by all eight waves (the packed order IS [chunk][stage]: one linear copy)
  auto load_filter = [&](const uint16_t* wpk) {
    constexpr int NP = G::WALL / 1024;
    const char* src = reinterpret_cast<const char*>(wpk);
#pragma unroll
    for (int j = 0; j < (NP + 7) / 8; ++j) {
      const int p = wave + 8 * j;
      if (p < NP) dma16(src + p * 1024 + lane * 16, sbase + W_OFF + (unsigned)p * 1024u);
    }
  };
  if constexpr (RESW) load_filter(jt.job[jb].wpk);
  if (is_hw) {
#pragma unroll
    for (int k = 0; k < HSTG; ++k) stage_in(jt.job[jb], lit, 0, sbase, k * HPER);
  } else if constexpr (!RESW) {
    stage_w(jt.job[jb].wpk, 0, sbase + W_OFF);
  }
  if constexpr (IN_POOLED) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    scatter_pooled<G>(smem + STG_OFF, smem, tid);
  }
  int hbuf = 0, wbuf = 0;
  bool first_item = true;

  for (; item < nitems; item += gridDim.x) {
    const int next_item = item + gridDim.x;
    const bool more = next_item < nitems;
    const int jn = more ? mm_job_of(jt, next_item) : jb, nlit = more ? next_item - jt.start[jn] : lit;
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    const int img = lit / RPI, rrem = lit % RPI;
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    constexpr bool ACTPF = EPI == EPI_DGRAD_ACT && NB >= 2;
    unsigned actv[ACTPF ? NB / 2 : 1][16];

#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool last_chunk = chunk + 1 == NCHUNK;
      const bool next_tile = !last_chunk || more;
      const bool to_next = last_chunk && more;
      const int n_chunk = last_chunk ? 0 : chunk + 1;
      const int nx_job = to_next ? jn : jb, n_lit = to_next ? nlit : lit;
      const int a_addr = a_lane + hbuf * HALO_BYTES;
#pragma unroll
      for (int sg = 0; sg < NSTG; ++sg) {
        if constexpr (RESW) {
          // resident filter: ONE barrier per chunk (its halo tile has landed; nobody reads the other buffer any more) + for a pooled
          // input the one before the scatter of the next tile's staging records
          if (sg == 0 || (IN_POOLED && sg == NSTG - 1)) {
            if (!(sg == 0 && chunk == 0 && !first_item)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
          }
        } else {
          if (!(sg == 0 && chunk == 0 && !first_item)) {
            if (!is_hw || sg == 0 || (IN_POOLED && sg == NSTG - 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __syncthreads();
        }
        if constexpr (ACTPF) {
          if (sg == NSTG - 1 && last_chunk) {
            const char* act = reinterpret_cast<const char*>(jt.job[jb].act) + (size_t)img * HW * HW * NC * 2;
#pragma unroll
            for (int m = 0; m < NB / 2; ++m)
#pragma unroll
              for (int rr = 0; rr < 16; ++rr) {
                const int g = rr >> 2, i = rr & 3;
                const unsigned pix = (unsigned)((ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * (2 * g + h) + (i & 1));
                actv[m][rr] = *reinterpret_cast<const unsigned*>(act + pix * (unsigned)(NC * 2) + (unsigned)(64 * m + 2 * (lane & 31)) * 2u);
              }
          }
        }
        if (!is_hw && !RESW) {
          if (sg + 1 < NSTG) {
            stage_w(jt.job[jb].wpk, chunk * NSTG + sg + 1, sbase + W_OFF + (unsigned)(wbuf ^ 1) * WSTAGE);
          } else if (next_tile) {
            stage_w(jt.job[nx_job].wpk, n_chunk * NSTG, sbase + W_OFF + (unsigned)(wbuf ^ 1) * WSTAGE);
          }
        } else if (is_hw && next_tile) {
          if (IN_POOLED ? sg == 0 : sg < HSTG) stage_in(jt.job[nx_job], n_lit, n_chunk, sbase + (unsigned)(hbuf ^ 1) * HALO_BYTES, sg * HPER);
        }
        if constexpr (IN_POOLED) {
          if (sg == NSTG - 1 && next_tile) scatter_pooled<G>(smem + STG_OFF, smem + (hbuf ^ 1) * HALO_BYTES, tid);
        }
        const int b_addr = b_lane + (RESW ? (chunk * NSTG + sg) * WSTAGE : wbuf * WSTAGE);
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
          const int tap = sg * TPS + t, dy = tap / 3, dx = tap % 3;
#pragma unroll
          for (int s = 0; s < KSTEPS; ++s) {
            const uint4 a = *reinterpret_cast<const uint4*>(smem + a_addr + (dy * G::HROW + dx * G::PS) * 16 + s * 32);
            uint4 b[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const uint4*>(smem + b_addr + ((t * KSTEPS + s) * NB + nb) * 1024);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_b(a, b[nb], acc[nb]);
          }
        }
        wbuf ^= 1;
      }
      hbuf ^= 1;
    }
    if constexpr (RESW) {
      if (more && jn != jb) {        // the next item belongs to another job (another modality's filter): everybody has left the matrix loop
        __syncthreads();
        load_filter(jt.job[jn].wpk);
      }
    }
    if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first_item = false;

    // ---- epilogue (lane <-> pixels / channels as in conv3x3_mm.hip)
    const MmJob& J = jt.job[jb];
    constexpr bool POOL = EPI == EPI_LRELU_POOL;
    constexpr int HO = POOL ? HW / 2 : HW;
    char* out = reinterpret_cast<char*>(J.out) + (size_t)img * HO * HO * NC * 2;
    const int c = lane & 31;
    if constexpr (NB >= 2) {
#pragma unroll
      for (int m = 0; m < NB / 2; ++m) {
        const unsigned chb = (unsigned)(64 * m + 2 * c) * 2u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int wx = 2 * g + h;
          if constexpr (POOL) {
            float best[2];
            unsigned bi[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              best[e] = acc[2 * m + e][4 * g];
              bi[e] = 0;
#pragma unroll
              for (int i = 1; i < 4; ++i) {
                const float v = acc[2 * m + e][4 * g + i];
                if (v > best[e]) { best[e] = v; bi[e] = i; }     // strict >: the FIRST maximum wins (TF MaxPoolGrad)
              }
              best[e] = ugn_lrelu(best[e]);
            }
            const unsigned pix = (unsigned)((ry0 / 2 + wave) * HO + rx0 / 2 + wx);
            *reinterpret_cast<unsigned*>(out + pix * (unsigned)(NC * 2) + chb) = bf_pack(best[0], best[1]);
            uint8_t* oi = J.out_idx + (size_t)img * HO * HO * NC;
            *reinterpret_cast<uint16_t*>(oi + pix * (unsigned)NC + (unsigned)(64 * m + 2 * c)) = (uint16_t)(bi[0] | (bi[1] << 8));
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const unsigned pix = (unsigned)((ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * wx + (i & 1));
              float v0 = acc[2 * m][4 * g + i], v1 = acc[2 * m + 1][4 * g + i];
              if constexpr (EPI == EPI_LRELU) {
                v0 = ugn_lrelu(v0);
                v1 = ugn_lrelu(v1);
              } else if constexpr (EPI == EPI_DGRAD_ACT) {
                const unsigned ab = actv[m][4 * g + i];
                v0 *= (short)(ab & 0xffffu) > 0 ? 1.f : UGN_LRELU_ALPHA;      // (a positive bf16 has a positive bit pattern)
                v1 *= (short)(ab >> 16) > 0 ? 1.f : UGN_LRELU_ALPHA;
              }
              *reinterpret_cast<unsigned*>(out + pix * (unsigned)(NC * 2) + chb) = bf_pack(v0, v1);
            }
          }
        }
      }
    } else {
      // one block: lane c owns channel c; the even lane of a pair stores both channels (4 bytes)
      auto store1 = [&](unsigned pix, float v) {
        const unsigned own = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v);
        const unsigned oth = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        if (!(c & 1)) *reinterpret_cast<unsigned*>(out + pix * (unsigned)(NC * 2) + (unsigned)(c * 2)) = own | (oth << 16);
      };
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int wx = 2 * g + h;
        if constexpr (POOL) {
          float best = acc[0][4 * g];
          unsigned bi = 0;
#pragma unroll
          for (int i = 1; i < 4; ++i) {
            const float v = acc[0][4 * g + i];
            if (v > best) { best = v; bi = i; }
          }
          const unsigned pix = (unsigned)((ry0 / 2 + wave) * HO + rx0 / 2 + wx);
          store1(pix, ugn_lrelu(best));
          uint8_t* oi = J.out_idx + (size_t)img * HO * HO * NC;
          oi[pix * (unsigned)NC + (unsigned)c] = (uint8_t)bi;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned pix = (unsigned)((ry0 + 2 * wave + (i >> 1)) * HW + rx0 + 2 * wx + (i & 1));
            float v = acc[0][4 * g + i];
            if constexpr (EPI == EPI_LRELU) {
              v = ugn_lrelu(v);
            } else if constexpr (EPI == EPI_DGRAD_ACT) {
              const char* act = reinterpret_cast<const char*>(J.act) + (size_t)img * HW * HW * NC * 2;
              v *= *reinterpret_cast<const short*>(act + pix * (unsigned)(NC * 2) + (unsigned)(c * 2)) > 0 ? 1.f : UGN_LRELU_ALPHA;
            }
            store1(pix, v);
          }
        }
      }
    }
    jb = jn;
    lit = nlit;
  }
}

template <int KC, int NC, int HW, int IN_POOLED, int EPI>
int launch_bf(const MmJob* jobs, const int* n, int njobs, hipStream_t st) {
  using G = BGeo<KC, NC, IN_POOLED>;
  auto kern = conv_bf_kernel<KC, NC, HW, IN_POOLED, EPI>;
  static_assert(G::LDS <= 163840, "LDS budget");
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) { ugn_set_error("conv_bf: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const void* zeros = zero_block();
  if (!zeros) { ugn_set_error("conv_bf: cannot allocate the zero block"); return UGN_EINVAL; }
  MmJobs jt;
  const int nitems = make_mm_table(jt, jobs, n, njobs, (HW / 16) * (HW / 16));
  // (two persistent workgroups per CU where the LDS allows it -- the 32 -> 32 forward, 72 KB: the second one's MFMAs cover
  //  the first one's tile wait and epilogue)
  const int WGS = (G::LDS <= 81920 ? 2 : 1) * persistent_wgs();
  const int grid = nitems < WGS ? nitems : WGS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("conv_bf");
  return 0;
}

}  // namespace

/* channels per K chunk the kernels use for a (layer, direction): what ugn_bf_pack_multi must lay the filter out for */
static int bf_chunk(int kc, bool pooled_in) { return (kc % 64 == 0 && !pooled_in) ? 64 : 32; }

extern "C" int ugn_bf_pack_multi(const float* const* w_hwio_host, uint16_t* const* wpk_host, const int* cin_host, const int* cout_host,
                                 const int* dgrad_host, const int* pooled_host, int njobs, void* stream) {
  UGN_REQUIRE(w_hwio_host && wpk_host && cin_host && cout_host && dgrad_host && pooled_host, "ugn_bf_pack_multi: null pointer");
  UGN_REQUIRE(njobs >= 1 && njobs <= kPackJobs, "ugn_bf_pack_multi: njobs must be 1..%d (got %d)", kPackJobs, njobs);
  BPackTable t = {};
  int maxe = 0;
  for (int j = 0; j < njobs; ++j) {
    const int ci = cin_host[j], co = cout_host[j];
    UGN_REQUIRE(w_hwio_host[j] && wpk_host[j] && ci > 0 && co > 0 && ci % 32 == 0 && co % 32 == 0, "ugn_bf_pack_multi: bad job %d", j);
    t.w[j] = w_hwio_host[j]; t.pk[j] = wpk_host[j]; t.cin[j] = ci; t.cout[j] = co; t.dgrad[j] = dgrad_host[j] ? 1 : 0;
    // forward: K = cin, never a pooled input; data gradient: K = cout, pooled input for the MaxPool'ed layers
    t.ch[j] = bf_chunk(dgrad_host[j] ? co : ci, dgrad_host[j] && pooled_host[j]);
    if (9 * ci * co > maxe) maxe = 9 * ci * co;
  }
  hipLaunchKernelGGL(bf_pack_kernel, dim3((maxe + 255) / 256, njobs), dim3(256), 0, (hipStream_t)stream, t);
  UGN_CHECK_LAUNCH("bf_pack");
  return 0;
}

extern "C" int ugn_bf_conv3x3_fwd_multi(const uint16_t* const* in, const uint16_t* const* wpk, uint16_t* const* out,
                                        uint8_t* const* out_idx, const int* n, int njobs, int hw, int cin, int cout, int pool,
                                        void* stream) {
  UGN_REQUIRE(in && wpk && out && n, "ugn_bf_conv3x3_fwd_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_bf_conv3x3_fwd_multi: njobs must be 1..%d (got %d)", kMaxJobs, njobs);
  MmJob jobs[kMaxJobs];
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(in[j] && wpk[j] && out[j] && n[j] > 0, "ugn_bf_conv3x3_fwd_multi: null pointer or n <= 0 in job %d", j);
    UGN_REQUIRE(!pool || (out_idx && out_idx[j]), "ugn_bf_conv3x3_fwd_multi: pool needs out_idx");
    jobs[j] = {in[j], nullptr, nullptr, wpk[j], nullptr, out[j], pool ? out_idx[j] : nullptr, nullptr, nullptr};
  }
  hipStream_t st = (hipStream_t)stream;
#define BFW(KC_, NC_, HW_, P_)                                             \
  if (cin == KC_ && cout == NC_ && hw == HW_ && (pool != 0) == (P_ != 0))  \
    return launch_bf<KC_, NC_, HW_, 0, P_ ? EPI_LRELU_POOL : EPI_LRELU>(jobs, n, njobs, st);
  BFW(32, 32, 64, 1) BFW(32, 64, 32, 0) BFW(64, 64, 32, 1) BFW(64, 128, 16, 0) BFW(128, 128, 16, 0)
#undef BFW
  ugn_set_error("ugn_bf_conv3x3_fwd_multi: unsupported shape cin=%d cout=%d hw=%d pool=%d", cin, cout, hw, pool);
  return UGN_EINVAL;
}

extern "C" int ugn_bf_conv3x3_dgrad_multi(const uint16_t* const* dz, const uint8_t* const* dz_idx, const uint16_t* const* wpk,
                                          const uint16_t* const* act, uint16_t* const* out, const int* n, int njobs, int hw, int cin,
                                          int cout, void* stream) {
  UGN_REQUIRE(dz && wpk && out && n, "ugn_bf_conv3x3_dgrad_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_bf_conv3x3_dgrad_multi: njobs must be 1..%d (got %d)", kMaxJobs, njobs);
  MmJob jobs[kMaxJobs];
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(dz[j] && wpk[j] && out[j] && n[j] > 0, "ugn_bf_conv3x3_dgrad_multi: null pointer or n <= 0 in job %d", j);
    jobs[j] = {dz[j], dz_idx ? dz_idx[j] : nullptr, nullptr, wpk[j], nullptr, out[j], nullptr, nullptr, act ? act[j] : nullptr};
    UGN_REQUIRE((jobs[0].in_idx != nullptr) == (jobs[j].in_idx != nullptr), "ugn_bf_conv3x3_dgrad_multi: dz_idx for all jobs or none");
    UGN_REQUIRE((jobs[0].act != nullptr) == (jobs[j].act != nullptr), "ugn_bf_conv3x3_dgrad_multi: act for all jobs or none");
  }
  const int unpool = jobs[0].in_idx != nullptr;
  const bool has_act = jobs[0].act != nullptr;
  hipStream_t st = (hipStream_t)stream;
#define BFD(CI_, CO_, HW_, U_)                                                      \
  if (cin == CI_ && cout == CO_ && hw == HW_ && unpool == U_)                       \
    return has_act ? launch_bf<CO_, CI_, HW_, U_, EPI_DGRAD_ACT>(jobs, n, njobs, st) \
                   : launch_bf<CO_, CI_, HW_, U_, EPI_DGRAD>(jobs, n, njobs, st);
  BFD(32, 32, 64, 1) BFD(32, 64, 32, 0) BFD(64, 64, 32, 1) BFD(64, 128, 16, 0) BFD(128, 128, 16, 0)
#undef BFD
  ugn_set_error("ugn_bf_conv3x3_dgrad_multi: unsupported shape cin=%d cout=%d hw=%d unpool=%d", cin, cout, hw, unpool);
  return UGN_EINVAL;
}
