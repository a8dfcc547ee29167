// Winograd F(2x2,3x3) forward / data gradient for the layers with 32 OUTPUT channels of the GEMM (a2 forward, a2 and a3
// data gradient; reference nets/mj_uwyhNets_ba.py:431-450): the "tall" variant of wino_kernel (conv3x3_wino.hip).
//
// With only 32 output channels the narrow variant transforms every input patch twice (once per 16-channel wave) and
// stages 108 bytes per MFMA.  Here a wave owns 16 tiles x ALL 32 channels (2 blocks, 128 accumulator registers), so a
// patch is transformed exactly once, and the 8 waves of a workgroup own 8 tile groups = a 32 x 16 pixel region:
//   * halo tile [34 x 18 pixels][16 channels + 4 pad] = 48 KB per 16-channel STAGE, double-buffered, LDS-DMA;
//   * filter slices [8-channel group][16 points][4 kq][16 lj][2 blocks][2 k] = 16 KB in a ring of 4: for 32 input channels
//     the whole transformed filter (64 KB) is RESIDENT in LDS for the life of the workgroup, otherwise the slice of group
//     g+1 streams in while group g computes;
//   * 163,840 B of LDS in all (the CU's 160 KiB), one workgroup per CU, 2 waves per SIMD.
// Per MFMA: 1 transform VALU instead of 2, 48 (resident filters) / 80 staged bytes instead of 108.  Everything else --
// in-register input transform in the shadow of the previous group's MFMAs, bank-exact tile layout (even columns first,
// a wave's two tile rows 32 banks apart), lane-local output transform + LeakyReLU + MaxPool + argmax, pooled-gradient
// input tile with the MaxPool scatter at patch-read time, two jobs per launch -- is as in wino_kernel.
#include <stdlib.h>
#include "wino_common.h"

#ifndef UGN_TALL_B64
#define UGN_TALL_B64 1
#endif

namespace ugn_wino {
namespace {

constexpr int TPW = 18, TPH = 34, TNPIX = TPH * TPW;     // 32 x 16 output region, 34 x 18 halo
constexpr int TCS = 20;                                  // halo pixel stride (floats): 16 channels + 4 pad = 5 slots
constexpr int THSLOTS = TNPIX * 5, THPIECES = 48;        // 3060 slots in 48 pieces of 1 KB
constexpr int TSIN = THPIECES * 256;                     // floats per halo buffer (49,152 B)
constexpr int TSUG = 16 * 4 * 16 * 4;                    // floats per filter slice of one 8-channel group (16 KB)
constexpr int NUB = 4;                                   // filter slice ring
constexpr int TLDS = (2 * TSIN + NUB * TSUG) * 4;        // 163,840 B
// pooled-gradient input: [18 x (10 + 2 pad) pooled pixels][16 values + 16 argmax bytes] = 20 floats per pixel
constexpr int TUPW = 12, TUPH = 18, TUCS = 20, TUSLOTS = TUPH * TUPW * 5, TUPIECES = 17;

__device__ __forceinline__ int tall_halo_geometry(int inst, int lane) {
  inst = inst < THPIECES ? inst : THPIECES - 1;
  int slot = inst * 64 + lane;
  slot = slot < THSLOTS ? slot : THSLOTS - 1;
  const int p = slot / 5, c4 = slot - p * 5;
  const int yy = p / TPW, xp = p - yy * TPW;
  const int xx = xp < 9 ? 2 * xp : 2 * (xp - 9) + 1;   // even columns first (colpos)
  return (yy << 16) | (xx << 8) | c4;
}

template <int KC, int HW>
__device__ __forceinline__ void tall_dma_halo(const float* __restrict__ in, const float* __restrict__ zeros, int img, int ry0,
                                              int rx0, int stage, int inst, int geom, unsigned lds_byte_base) {
  inst = inst < THPIECES ? inst : THPIECES - 1;
  const int yy = geom >> 16, xx = (geom >> 8) & 0xff, c4 = geom & 0xff;
  const int gy = ry0 - 1 + yy, gx = rx0 - 1 + xx;
  const bool ok = c4 < 4 && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
  const float* src = ok ? in + (((size_t)img * HW + gy) * HW + gx) * KC + stage * 16 + c4 * 4 : zeros;
  dma16(src, lds_byte_base + (unsigned)inst * 1024u);
}

// slot s = pooled pixel position s/5 (row-major, 12 positions per row of which 10 are pixels), part s%5 (0..3 values, 4 argmax)
template <int KC, int HW>
__device__ __forceinline__ void tall_dma_pooled(const float* __restrict__ dz, const uint8_t* __restrict__ idx,
                                                const float* __restrict__ zeros, int img, int ry0, int rx0, int stage, int inst,
                                                int lane, unsigned lds_byte_base) {
  constexpr int HP = HW / 2;
  inst = inst < TUPIECES ? inst : TUPIECES - 1;
  int slot = inst * 64 + lane;
  slot = slot < TUSLOTS ? slot : TUSLOTS - 1;
  const int p = slot / 5, c = slot - p * 5;
  const int prow = p / TUPW, ppos = p - prow * TUPW;
  const int pr = ry0 / 2 - 1 + prow, pc = rx0 / 2 - 1 + ppos;
  const bool ok = pr >= 0 && pr < HP && pc >= 0 && pc < HP && ppos < 10;
  const size_t o = (((size_t)img * HP + pr) * HP + pc) * KC + stage * 16;
  const void* src = !ok ? (const void*)zeros : (c < 4 ? (const void*)(dz + o + c * 4) : (const void*)(idx + o));
  dma16(src, lds_byte_base + (unsigned)inst * 1024u);
}

template <bool B64 = false>
__device__ __forceinline__ void tall_read_plain(float2 (&dn)[16], const float* base) {
  if constexpr (B64) {      // unfused ds_read_b64 (wino_common.h lds_read_b64); patch_wait() before the first use
    const unsigned a0 = lds_addr(base);
#define UGN_RD(e_) dn[e_] = lds_read_b64<(((e_) >> 2) * TPW + colpos((e_) & 3)) * TCS * 4>(a0);
    UGN_RD(0) UGN_RD(1) UGN_RD(2) UGN_RD(3) UGN_RD(4) UGN_RD(5) UGN_RD(6) UGN_RD(7)
    UGN_RD(8) UGN_RD(9) UGN_RD(10) UGN_RD(11) UGN_RD(12) UGN_RD(13) UGN_RD(14) UGN_RD(15)
#undef UGN_RD
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) dn[e] = *reinterpret_cast<const float2*>(base + ((e >> 2) * TPW + colpos(e & 3)) * TCS);
  }
}

// pooled row PROW of the 3x3 pooled pixels under the patch (0 -> patch row 0, 1 -> rows 1 and 2, 2 -> row 3)
template <int PROW>
__device__ __forceinline__ void tall_read_pooled_row(float2 (&dn)[16], const float* base, const uint8_t* ibytes) {
  float2 pv[3];
  unsigned iw[3];
#pragma unroll
  for (int qc = 0; qc < 3; ++qc) {
    pv[qc] = *reinterpret_cast<const float2*>(base + (PROW * TUPW + qc) * TUCS);
    iw[qc] = *reinterpret_cast<const uint16_t*>(ibytes + (PROW * TUPW + qc) * TUCS * 4);
  }
#pragma unroll
  for (int r = (PROW == 0 ? 0 : (PROW == 1 ? 1 : 3)); r <= (PROW == 0 ? 0 : (PROW == 1 ? 2 : 3)); ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int qc = (c + 1) >> 1;
      const unsigned pos = (((r + 1) & 1) << 1) | ((c + 1) & 1);
      dn[r * 4 + c].x = (iw[qc] & 0xffu) == pos ? pv[qc].x : 0.f;
      dn[r * 4 + c].y = (iw[qc] >> 8) == pos ? pv[qc].y : 0.f;
    }
}

// 16 KB filter slice: 2 pieces of 1 KB per wave (bf16 elements: 8 KB, 1 piece)
template <bool BF>
__device__ __forceinline__ void tall_dma_u(const float* __restrict__ us, unsigned lds_byte_base, int tid, int wave) {
#pragma unroll
  for (int q = 0; q < (BF ? 1 : 2); ++q) dma16(us + (q * 512 + tid) * 4, lds_byte_base + (unsigned)(q * 512 + wave * 64) * 16u);
}

// KC: GEMM K channels (32 or 64); 32 output channels; HW: image size
template <int KC, int HW, int IN_UNPOOL, int EPI, int EFLAGS, bool BF = false>
__global__ __launch_bounds__(512, 2) void wino_tall_kernel(const WinoJobs jt, const float* __restrict__ zeros) {
  // unfused patch reads: -3.5 % on the 64 -> 32 data gradient (a3 | b1); the 32 -> 32 forward spills with them (+24 %)
  constexpr bool B64 = UGN_TALL_B64 && !IN_UNPOOL && !BF && (KC == 64 || UGN_TALL_B64 == 2);
  constexpr int PK = (UGN_PK && !BF) ? 1 : 0;   // packed transform arithmetic (wino_common.h pk_add)
  constexpr int PKE = PK;
  constexpr int NST = KC / 16;                 // 16-channel stages per item
  constexpr int NGI = 2 * NST;                 // 8-channel groups per item
  constexpr int NCF = 32;
  constexpr int RPX = HW / 16, RPY = HW / 32, RPI = RPX * RPY;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sU0 = smem + 2 * TSIN;
  const unsigned sin_bytes = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem);
  const unsigned su_bytes = sin_bytes + 2u * TSIN * 4u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lj = lane & 15, kq = lane >> 4;
  const int a_tr = lj >> 3, a_tc = lj & 7;
  // wave = tile group: tile rows {trow0, trow0 + 2} of the region's 16 (two rows 32 banks apart, see conv3x3_wino.hip)
  const int trow0 = (wave & 1) + 4 * (wave >> 1);
  const int a_trow = trow0 + 2 * a_tr;
  const int pbase = IN_UNPOOL ? (a_trow * TUPW + a_tc) * TUCS + 2 * kq : (2 * a_trow * TPW + a_tc) * TCS + 2 * kq;
  const int ibase = ((a_trow * TUPW + a_tc) * TUCS + 16) * 4 + 2 * kq;   // BYTE offset of the lane's argmax pair (pooled tile)
  const int ubase = (kq * 16 + lj) * 4;

  int item = blockIdx.x;
  const int nitems = jt.start[kMaxJobs];
  if (item >= nitems) return;
  int hgeo[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) hgeo[j] = tall_halo_geometry(wave * 6 + j, lane);
  // item -> job (wave-uniform: the table is read with scalar loads, once per item, at the item boundary)
  int jb = wino_job_of(jt, item), lit = item - jt.start[jb];
  auto u_slice = [&](const float* upk, int gi) { return upk + (size_t)gi * (BF ? TSUG / 2 : TSUG); };
  auto dma_stage = [&](const float* in, const uint8_t* in_idx, int region, int stage, int j, unsigned lds) {   // piece j (0..5 plain, 0..2 pooled) of this wave
    const int img = region / RPI, rrem = region % RPI;
    const int ry0 = (rrem / RPX) * 32, rx0 = (rrem % RPX) * 16;
    if constexpr (IN_UNPOOL)
      tall_dma_pooled<KC, HW>(in, in_idx, zeros, img, ry0, rx0, stage, wave * 3 + j, lane, lds);
    else
      tall_dma_halo<KC, HW>(in, zeros, img, ry0, rx0, stage, wave * 6 + j, hgeo[j], lds);
  };
  // The transformed filter of a 32-channel layer fits the ring: its NUB slices stay RESIDENT while the workgroup's items
  // belong to one job.  res[s] = the job whose slice s sits in ring slot s (an item walks the slots 0..NUB-1 in order): a
  // slice is fetched only when the group about to need it belongs to another job, i.e. at a job boundary.
  constexpr bool u_resident = NGI == NUB;
  int res[NUB];
  // ---- prologue: halo(item, stage 0) -> sIn[0]; filter slice(s) -> ring
#pragma unroll
  for (int j = 0; j < (IN_UNPOOL ? 3 : 6); ++j) dma_stage(jt.job[jb].in, jt.job[jb].in_idx, lit, 0, j, sin_bytes);
  if (u_resident) {
#pragma unroll
    for (int gi = 0; gi < NUB; ++gi) {
      tall_dma_u<BF>(u_slice(jt.job[jb].upk, gi), su_bytes + (unsigned)gi * TSUG * 4u, tid, wave);
      res[gi] = jb;
    }
  } else {
    tall_dma_u<BF>(u_slice(jt.job[jb].upk, 0), su_bytes, tid, wave);
  }
  bool upend = false;   // a filter slice was fetched during the previous group (resident mode): publish it
  int ibuf = 0, ubuf = 0;
  float V[16][2];      // transformed patch (2 channels) of the group about to be multiplied
  uint32_t Vp[16];     // BF: the same as a bf16 pair, rounded as it is produced
  auto setV = [&](int pt, float x, float y) {
    if constexpr (BF) Vp[pt] = pk_bf16(x, y);
    else { V[pt][0] = x; V[pt][1] = y; }
  };
  bool first = true;

  for (; item < nitems; item += gridDim.x) {
    const int next_item = item + gridDim.x;
    // the next item's job: its first halo stage and filter slice are fetched during this item's last stage
    const bool more = next_item < nitems;
    const int jn = more ? wino_job_of(jt, next_item) : jb, nlit = more ? next_item - jt.start[jn] : lit;
    f32x4 acc[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int pt = 0; pt < 16; ++pt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[cb][pt][r] = 0.f;

#pragma unroll(NST > 2 ? 1 : NST)   // (four stages unrolled hoist the DMA addresses of all of them and spill)
    for (int st = 0; st < NST; ++st) {
      const bool last_stage = st + 1 == NST;
      const bool has_next = !last_stage || next_item < nitems;
      // no next stage (last stage of the last item): re-fetch the current one into the free buffers (branch-free MFMA stream)
      const bool to_next = last_stage && has_next;      // the prefetches of this stage belong to the next item
      const int n_stage = last_stage ? (has_next ? 0 : st) : st + 1;
      const int n_lit = to_next ? nlit : lit;
      const int nx_job = to_next ? jn : jb;             // (ONE table entry is read: selecting between two pointers instead
      const float* nx_in = jt.job[nx_job].in;            //  makes the compiler form both sets of DMA addresses)
      const uint8_t* nx_idx = jt.job[nx_job].in_idx;
      const float* nx_upk = jt.job[nx_job].upk;
      const float* sIn = smem + ibuf * TSIN;
      const float* sInNext = smem + (ibuf ^ 1) * TSIN;
#pragma unroll
      for (int G = 0; G < 2; ++G) {
        // Publish what the previous group's DMA brought.  With resident filters the first group of a stage consumes nothing
        // new (its tile was published a group ago, and the buffer its DMA refills has had no reader since the last barrier).
        if (G == 1 || !u_resident || first || upend) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          upend = false;
        }
        const float* sU = sU0 + ubuf * TSUG;
        {   // filter slice of the next group -> next ring slot, while this group computes (resident: only if another job's)
          const int nslot = (ubuf + 1) & (NUB - 1);
          const int want = G == 0 ? jb : nx_job;
          bool fetch = true;
          if constexpr (u_resident) {   // (ubuf is the compile-time 2 * st + G here: an item walks the ring exactly once)
            fetch = res[(2 * st + G + 1) & (NUB - 1)] != want;
            res[(2 * st + G + 1) & (NUB - 1)] = want;
            upend = fetch;
          }
          if (fetch)
            tall_dma_u<BF>(G == 0 ? u_slice(jt.job[jb].upk, 2 * st + 1) : u_slice(nx_upk, 2 * n_stage), su_bytes + (unsigned)nslot * TSUG * 4u, tid, wave);
        }
        if (first) {
          first = false;
          float2 d[16], t[16];
          if constexpr (IN_UNPOOL) {
            tall_read_pooled_row<0>(d, sIn + pbase, reinterpret_cast<const uint8_t*>(sIn) + ibase);
            tall_read_pooled_row<1>(d, sIn + pbase, reinterpret_cast<const uint8_t*>(sIn) + ibase);
            tall_read_pooled_row<2>(d, sIn + pbase, reinterpret_cast<const uint8_t*>(sIn) + ibase);
          } else {
            tall_read_plain<B64>(d, sIn + pbase);
            if constexpr (B64) patch_wait(d);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            t[0 + c] = make_float2(d[0 + c].x - d[8 + c].x, d[0 + c].y - d[8 + c].y);
            t[4 + c] = make_float2(d[4 + c].x + d[8 + c].x, d[4 + c].y + d[8 + c].y);
            t[8 + c] = make_float2(d[8 + c].x - d[4 + c].x, d[8 + c].y - d[4 + c].y);
            t[12 + c] = make_float2(d[4 + c].x - d[12 + c].x, d[4 + c].y - d[12 + c].y);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            setV(r * 4 + 0, t[r * 4 + 0].x - t[r * 4 + 2].x, t[r * 4 + 0].y - t[r * 4 + 2].y);
            setV(r * 4 + 1, t[r * 4 + 1].x + t[r * 4 + 2].x, t[r * 4 + 1].y + t[r * 4 + 2].y);
            setV(r * 4 + 2, t[r * 4 + 2].x - t[r * 4 + 1].x, t[r * 4 + 2].y - t[r * 4 + 1].y);
            setV(r * 4 + 3, t[r * 4 + 1].x - t[r * 4 + 3].x, t[r * 4 + 1].y - t[r * 4 + 3].y);
          }
        }
        // the NEXT group's patch: second half of this stage (G == 0) or the next stage's tile, which landed a group ago
        const float* sNx = (G == 0 ? sIn + 8 : sInNext) + pbase;
        const uint8_t* sNi = reinterpret_cast<const uint8_t*>(G == 0 ? sIn : sInNext) + (G == 0 ? 8 : 0) + ibase;
        float2 dn[16];   // the row pass runs in place
        auto rowpass = [&](int c) {
          const float2 d0 = dn[0 + c], d1 = dn[4 + c], d2 = dn[8 + c], d3 = dn[12 + c];
          dn[0 + c] = pk_sub<PK>(d0, d2);
          dn[4 + c] = pk_add<PK>(d1, d2);
          dn[8 + c] = pk_sub<PK>(d2, d1);
          dn[12 + c] = pk_sub<PK>(d1, d3);
        };
        auto colpass = [&](int r) {
          auto put = [&](int pt, float2 v) { setV(pt, v.x, v.y); };
          put(r * 4 + 0, pk_sub<PK>(dn[r * 4 + 0], dn[r * 4 + 2]));
          put(r * 4 + 1, pk_add<PK>(dn[r * 4 + 1], dn[r * 4 + 2]));
          put(r * 4 + 2, pk_sub<PK>(dn[r * 4 + 2], dn[r * 4 + 1]));
          put(r * 4 + 3, pk_sub<PK>(dn[r * 4 + 1], dn[r * 4 + 3]));
        };
        // points in PAIRS, their k-steps and channel blocks interleaved: no MFMA waits on its predecessor
        float4 u[2][2];   // {cb0 s0, cb0 s1, cb1 s0, cb1 s1}
        uint2 ub[2][2];   // BF: the same 4 elements as bf16
        if constexpr (BF) {
          ub[0][0] = *reinterpret_cast<const uint2*>(sU + ubase / 2);
          ub[0][1] = *reinterpret_cast<const uint2*>(sU + ubase / 2 + 128);
        } else {
          u[0][0] = *reinterpret_cast<const float4*>(sU + ubase);
          u[0][1] = *reinterpret_cast<const float4*>(sU + ubase + 256);
        }
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
          const int cu = pp & 1, nu = cu ^ 1;
          // (no priority turns here: they cost the a2 forward 9 %, where the same change gains 3-4 % in wino_kernel's forwards)
          if (pp < 7) {
            if constexpr (BF) {
              ub[nu][0] = *reinterpret_cast<const uint2*>(sU + ubase / 2 + (2 * pp + 2) * 128);
              ub[nu][1] = *reinterpret_cast<const uint2*>(sU + ubase / 2 + (2 * pp + 3) * 128);
            } else {
              u[nu][0] = *reinterpret_cast<const float4*>(sU + ubase + (2 * pp + 2) * 256);
              u[nu][1] = *reinterpret_cast<const float4*>(sU + ubase + (2 * pp + 3) * 256);
            }
          }
          if constexpr (BF) {   // one bf16 MFMA per channel block: the lane's 2 channels in k-slots 0, 1 (2, 3 empty)
            const uint32_t a0 = Vp[2 * pp], a1 = Vp[2 * pp + 1];
            // B is the register pair as loaded: k-slots 0, 1 = block 0's filters, 2, 3 = block 1's; A selects the block
            acc[0][2 * pp] = mfma_bf16(a0, 0u, ub[cu][0].x, ub[cu][0].y, acc[0][2 * pp]);
            acc[0][2 * pp + 1] = mfma_bf16(a1, 0u, ub[cu][1].x, ub[cu][1].y, acc[0][2 * pp + 1]);
            acc[1][2 * pp] = mfma_bf16(0u, a0, ub[cu][0].x, ub[cu][0].y, acc[1][2 * pp]);
            acc[1][2 * pp + 1] = mfma_bf16(0u, a1, ub[cu][1].x, ub[cu][1].y, acc[1][2 * pp + 1]);
          } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) {
                acc[cb][2 * pp] = mfma16(V[2 * pp][s], u[cu][0][cb * 2 + s], acc[cb][2 * pp]);
                acc[cb][2 * pp + 1] = mfma16(V[2 * pp + 1][s], u[cu][1][cb * 2 + s], acc[cb][2 * pp + 1]);
              }
          }
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int pt = 2 * pp + half;
            if constexpr (IN_UNPOOL) {
              if (pt == 0) tall_read_pooled_row<0>(dn, sNx, sNi);
              if (pt == 1) tall_read_pooled_row<1>(dn, sNx, sNi);
              if (pt == 2) tall_read_pooled_row<2>(dn, sNx, sNi);
              if (G == 0 && pt >= 1 && pt <= 3) dma_stage(nx_in, nx_idx, n_lit, n_stage, pt - 1, sin_bytes + (unsigned)(ibuf ^ 1) * TSIN * 4u);
              if (pt == 3) { rowpass(0); rowpass(1); }
              if (pt == 4) { rowpass(2); rowpass(3); }
            } else {
              if (pt == 0) tall_read_plain<B64>(dn, sNx);
              if (G == 0 && pt >= 1 && pt < 7) dma_stage(nx_in, nx_idx, n_lit, n_stage, pt - 1, sin_bytes + (unsigned)(ibuf ^ 1) * TSIN * 4u);
              if (B64 && pt == 1) patch_wait(dn);
              if (pt == 1) { rowpass(0); rowpass(1); }
              if (pt == 2) { rowpass(2); rowpass(3); }
            }
            if (pt == 6) colpass(0);
            if (pt == 8) colpass(1);
            if (pt == 12) colpass(2);
            if (pt == 15) colpass(3);
          }
        }
        ubuf = (ubuf + 1) & (NUB - 1);
      }
      ibuf ^= 1;
    }

    // ---- output transform + epilogue: lane holds tiles 4*kq + r (r = 0..3) x channels {lj, 16 + lj}, all 16 points
    const int region = lit;
    const int img = region / RPI, rrem = region % RPI;
    const int ry0 = (rrem / RPX) * 32, rx0 = (rrem % RPX) * 16;
    float* out = jt.job[jb].out;
    uint8_t* out_idx = jt.job[jb].out_idx;
    const float* act = jt.job[jb].act;
    const float* addend = jt.job[jb].addend;
    float* raw_out = jt.job[jb].raw_out;
    const float* sm_m = jt.job[jb].smax_m;
    const float* sm_g = jt.job[jb].smax_g;
    {   // advance every tensor to this image (pooled outputs are a quarter of the size; set-level tensors are per clip)
      constexpr size_t IMG = (size_t)HW * HW * NCF, OIMG = EPI == EPI_LRELU_POOL ? IMG / 4 : IMG;
      out += (size_t)img * OIMG;
      if constexpr (EPI == EPI_LRELU_POOL) out_idx += (size_t)img * OIMG;
      if constexpr (EPI == EPI_DGRAD) {
        if constexpr (EFLAGS & 1) act += (size_t)img * IMG;
        if constexpr (EFLAGS & 2) addend += (size_t)img * IMG;
        if constexpr (EFLAGS & 4) raw_out += (size_t)img * IMG;
        if constexpr (EFLAGS & 8) {
          const int clip = img / jt.job[jb].frames;
          sm_m += (size_t)clip * IMG;
          sm_g += (size_t)clip * IMG;
        }
      }
    }
    const int co = UGN_EPI_PAIR ? 2 * lj : lj;      // the lane's first channel (wino_common.h pair_lj / pair_cb)
    float y[2][4][4];      // [block][tile r][output (a,b) row-major]
    unsigned o[4][4];      // element offsets inside the image: the image base is wave-uniform and rides in SGPRs
    wino_out_transform<2, PKE>(acc, y);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ti = 4 * kq + r, tr = ti >> 3, tc = ti & 7;
      const int oy = ry0 + 2 * (trow0 + 2 * tr), ox = rx0 + 2 * tc;
      if constexpr (EPI == EPI_LRELU_POOL) {
        constexpr int HP = HW / 2;
        o[r][0] = (unsigned)(((oy / 2) * HP + ox / 2) * NCF + co);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[r][q] = (unsigned)(((oy + (q >> 1)) * HW + ox + (q & 1)) * NCF + co);
      }
    }
    wino_epilogue<2, EPI, EFLAGS>(y, o, out, out_idx, act, addend, raw_out, sm_m, sm_g);
    jb = jn; lit = nlit;
  }
}

template <int KC, int HW, int IN_UNPOOL, int EPI, int EFLAGS, bool BF = false>
int launch_tall_t(const WinoJob* jobs, const int* n, int njobs, hipStream_t st) {
  auto kern = wino_tall_kernel<KC, HW, IN_UNPOOL, EPI, EFLAGS, BF>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TLDS);
    if (e != hipSuccess) { ugn_set_error("wino tall: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const float* zeros = zero_block();
  if (!zeros) { ugn_set_error("wino tall: cannot allocate the zero block"); return UGN_EINVAL; }
  constexpr int per_img = (HW / 16) * (HW / 32);
  WinoJobs jt;
  const int nitems = make_job_table(jt, jobs, n, njobs, per_img);
  const int grid = nitems < kGrid ? nitems : kGrid;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), TLDS, st, jt, zeros);
  UGN_CHECK_LAUNCH("wino tall");
  return 0;
}

template <int KC, int HW, int IN_UNPOOL>
int launch_tall_dgrad(const WinoJob* jobs, const int* n, int njobs, bool bf, hipStream_t st) {
  const int flags = (jobs[0].act ? 1 : 0) | (jobs[0].addend ? 2 : 0) | (jobs[0].raw_out ? 4 : 0) | (jobs[0].smax_m ? 8 : 0);
  if (bf) { ugn_set_error("the bf16-operand Winograd kernels (fp32 tensors, 'bf16w') were retired in round 5: conv_precision='bf16' is the configs[4] path"); return UGN_EINVAL; }
  if constexpr (KC == 64 && HW == 32 && !IN_UNPOOL)   // a3: act + routed set-max gradient
    if (flags == 9) return launch_tall_t<KC, HW, IN_UNPOOL, EPI_DGRAD, 9>(jobs, n, njobs, st);
#define UGN_TDG(F_) \
  case F_:          \
    return launch_tall_t<KC, HW, IN_UNPOOL, EPI_DGRAD, F_>(jobs, n, njobs, st);
  switch (flags) {
    UGN_TDG(0) UGN_TDG(1) UGN_TDG(3) UGN_TDG(5) UGN_TDG(7)
    default: break;
  }
#undef UGN_TDG
  ugn_set_error("ugn_conv3x3_dgrad_wino: unsupported epilogue combination %d (addend/raw_out need act)", flags);
  return UGN_EINVAL;
}

}  // namespace

// kind 0: forward (kc = cin, flag = pool); kind 1: data gradient (kc = cout, flag = pooled dz).  32 GEMM output channels.
bool tall_supported(int kind, int hw, int kc, int flag) {
  if (kind == 0) return hw == 64 && kc == 32 && flag;                                    // a2 forward
  return (hw == 64 && kc == 32 && flag) || (hw == 32 && kc == 64 && !flag);              // a2, a3 / b1 data gradient
}

int launch_tall(int kind, const WinoJob* jobs, const int* n, int njobs, int hw, int kc, int flag, bool bf, hipStream_t st) {
  if (bf) { ugn_set_error("the bf16-operand Winograd kernels (fp32 tensors, 'bf16w') were retired in round 5: conv_precision='bf16' is the configs[4] path"); return UGN_EINVAL; }
  if (kind == 0 && hw == 64 && kc == 32 && flag)
    return launch_tall_t<32, 64, 0, EPI_LRELU_POOL, 0>(jobs, n, njobs, st);
  if (kind == 1 && hw == 64 && kc == 32 && flag) return launch_tall_dgrad<32, 64, 1>(jobs, n, njobs, bf, st);
  if (kind == 1 && hw == 32 && kc == 64 && !flag) return launch_tall_dgrad<64, 32, 0>(jobs, n, njobs, bf, st);
  ugn_set_error("wino tall: unsupported shape kind=%d hw=%d kc=%d flag=%d", kind, hw, kc, flag);
  return UGN_EINVAL;
}

}  // namespace ugn_wino
