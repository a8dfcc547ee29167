// Winograd F(2x2,3x3) forward / data-gradient convolution on v_mfma_f32_16x16x4_f32 (exact fp32, 2.25x fewer matrix
// FLOPs than the direct implicit GEMM in conv3x3.hip).  Same operator contract as conv3x3_kernel: replaces the implicit
// TF Conv2D / Conv2DBackpropInput + LeakyRelu(Grad) + MaxPool(Grad) of reference nets/mj_uwyhNets_ba.py:431-462.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// Mapping.  A 512-thread workgroup (8 waves, two per SIMD; persistent: 256 workgroups stride over the items) owns a 16x16
// output region (64 Winograd tiles).  The GEMM of each of the 16 Winograd points is [tiles] x [out channels] x [K channels] on
// 16x16x4 MFMAs.  Two variants (chosen from the GEMM dimensions, wino_common.h): NARROW -- the workgroup owns 32 output
// channels, a wave 16 tiles x 16 channels (64 accumulator registers), 16-channel K groups; WIDE -- 64 output channels, a wave
// 16 tiles x 32 channels (128 accumulators), 8-channel K groups, so one transformed patch feeds two channel blocks.  The
// layers with 32 output channels use the TALL kernel of conv3x3_wino_tall.hip.
//   * The transformed filters U = G g G^T are precomputed once per step (wino_pack_multi) in the order the kernel consumes
//     them; a 32 KB slice per K group is double-buffered in LDS by LDS-DMA (global_load_lds_dwordx4) -> one barrier per group
//     (64 MFMAs per wave).  Single-chunk narrow layers keep both slices resident.
//   * The input transform B^T d B is computed IN REGISTERS by the lane that feeds it to the MFMA (lane = tile x channel
//     pair), straight from the fp32 halo tile in LDS and in the shadow of the previous group's MFMAs: transformed
//     activations never touch LDS or HBM.
//   * The halo tile of the next chunk / next item streams in by LDS-DMA (per-lane source addresses; lanes outside the image
//     read a zero block) early in the first group of a chunk, so global latency is never exposed.
//   * The output transform A^T M A is lane-local (a lane holds all 16 points of its 4 tiles), and a 2x2 Winograd tile IS a
//     pooling window, so LeakyReLU + MaxPool + argmax need no cross-lane traffic either.
//   * One launch can carry two jobs of the same shape (the frame-level layer and its set-level twin of the global branch).
#include <stdlib.h>
#include "wino_common.h"

using namespace ugn_wino;

namespace {

#ifndef UGN_STAMPS
#define UGN_STAMPS 0     // diagnostic build only: s_memtime stamps around every group barrier of workgroup 0 (tools/stamps.py)
#endif
#if UGN_STAMPS
__device__ unsigned long long ugn_stamp_buf[8 * 2048];
#endif

constexpr int TW = 16, PW = TW + 2, PH = 18, NPIX = PH * PW;   // 16x16 output region, 18x18 halo
constexpr int CS = 36;                                         // halo pixel stride (floats) per 32-channel chunk
constexpr int HSLOTS = NPIX * 9;                                // 16-byte slots of a halo tile (8 data + 1 pad per pixel)
// An LDS-DMA piece (64 lanes x 16 B) starts at a PIXEL boundary: 7 pixels = 63 slots, and lane 63 writes the first slot of
// the next pixel (the same data the next piece puts there).  A lane's place inside its pixel group is then a constant of the
// lane, and the per-piece address arithmetic is one multiply-shift for the halo row instead of divisions by 9 and 18.
constexpr int HPX = 7, HPIECES = (NPIX + HPX - 1) / HPX;        // 47 pieces (8 waves x 6 = 48 issue slots)
constexpr int SIN = ((HPIECES - 1) * HPX * 9 + 64) * 4;         // one halo buffer (floats) = 47,392 B incl. the last piece's overhang
constexpr int SU = 16 * 4 * 32 * 4;                             // one filter slice: [16 pts][4 kq][32 out][4 ch] = 32 KB
constexpr int LDS_BYTES = (2 * SIN + 2 * SU) * 4 + 8 * 256;     // 160,320 B + a 256-B dump per wave = 162,368 (of 163,840)
// pooled-resolution input (data gradient of a MaxPool'ed layer): the LDS tile holds the 10x10 POOLED pixels under the halo,
// 40 floats per pixel = 32 gradient values + 32 argmax bytes; the scatter through the argmax happens when a lane reads
// its 4x4 patch (3x3 pooled pixels), so the 4x larger un-pooled gradient is never materialised anywhere.
constexpr int UPW = 12, UCS = 44, UPIX = 10 * UPW, USLOTS = UPIX * 11;   // 10 rows x (10 + 2 pad) positions
constexpr int UPX = 5, UPIECES = UPIX / UPX;                    // pooled pieces: 5 positions = 55 slots (+ 9 of the next): 24 pieces
static_assert(((UPIECES - 1) * UPX * 11 + 64) * 4 <= SIN, "pooled tile fits the halo buffer");
// LDS bank layout.  A lane of an MFMA is (tile lj, channel slot kq) and reads its patch with ds_read_b64; the hardware
// serves 32 lanes per pass, i.e. all 16 tiles x 2 channel slots.  Two choices make every pass touch each bank exactly twice
// (ds_read_b64: 64 banks, 32 lanes per pass) -- the b64 optimum: (1) the halo tile stores the EVEN columns of a row first,
// then the odd ones, so the 8 tile columns fall into 8 different bank quads (pixel stride 36 floats; pooled tile: 44);
// (2) a wave's two tile rows are chosen 32 banks apart (see trow0 in the kernel); (3) lane kq owns channels
// {2kq, 2kq+1, 8+2kq, 9+2kq} of a 16-channel group, so the two slots of a pass are the two halves of one quad.

// U[pt][n][k] = (G g G^T)[pt] for filter g(k -> n).  Layout [nsp][chunk][G][pt][kq][32 n][4 s] with k = 32*chunk + 16*G +
// 8*(s>>1) + 2*kq + (s&1): a 32 KB slice per 16-channel group, and the lane (n, kq) of an MFMA reads its 4 steps with one
// ds_read_b128 at consecutive 16-byte slots (conflict-free).
// Layers with >= 64 output channels use the WIDE kernel variant (a wave owns 16 tiles x 32 channels) and its own layout.
// fwd : g[dy][dx] = w[dy][dx][k = cin][n = cout]            (kc = cin, nc = cout)
// dgrad: g[dy][dx] = w[2-dy][2-dx][n = cin][k = cout]        (kc = cout, nc = cin)
__device__ __forceinline__ void wino_pack_one(const float* __restrict__ w, float* __restrict__ u, int cin, int cout, int mode,
                                              int e) {
  const int dgrad = mode & 1;   // mode: 0 fwd, 1 dgrad of a full-resolution dz, 3 dgrad of a pooled dz (same layout as 1)
  const bool bf = (mode & 4) != 0;   // + 4: bf16 elements in the same element order (the buffer is then half used)
  auto put = [&](float* base, int off, float v) {
    if (bf) reinterpret_cast<__bf16*>(u)[(base - u) + off] = (__bf16)v;
    else base[off] = v;
  };
  const int kc = dgrad ? cout : cin, nc = dgrad ? cin : cout;
  if (e >= kc * nc) return;
  // consecutive threads walk the contiguous axis of the HWIO filter: cout = n in the forward, cout = k in the data gradient
  const int k = dgrad ? e % kc : e / nc, n = dgrad ? e / kc : e % nc;
  float g[3][3];
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx)
      g[dy][dx] = dgrad ? w[(((2 - dy) * 3 + (2 - dx)) * cin + n) * cout + k] : w[((dy * 3 + dx) * cin + k) * cout + n];
  float t[4][3];   // G g
  for (int c = 0; c < 3; ++c) {
    t[0][c] = g[0][c];
    t[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
    t[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
    t[3][c] = g[2][c];
  }
  const int nchunk = kc >> 5, chunk = k >> 5, kq = (k >> 1) & 3;
  float* dst;
  if (wino_tall(kc, nc)) {
    // tall (conv3x3_wino_tall.hip): 8-channel groups gi = k >> 3, [gi][pt][kq][16 lj][cb][s], n = 2*lj + cb: a lane's two
    // accumulator blocks are ADJACENT channels, so the epilogue moves 8 bytes per lane and 128-byte lines per 16 lanes
    dst = u + (size_t)(k >> 3) * 4096 + (kq * 16 + pair_lj(n)) * 4 + pair_cb(n) * 2 + (k & 1);
    for (int r = 0; r < 4; ++r) {
      const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]),
                  u3 = t[r][2];
      put(dst, (r * 4 + 0) * 256, u0);
      put(dst, (r * 4 + 1) * 256, u1);
      put(dst, (r * 4 + 2) * 256, u2);
      put(dst, (r * 4 + 3) * 256, u3);
    }
    return;
  }
  if (wino_wide_ex(kc, nc, bf, (mode & 3) == 3)) {
    // wide: a workgroup owns 64 output channels, a wave 2 blocks of 16 (cb); 8-channel groups (k = 32*chunk + 8*G + 2*kq + s):
    // [nsp64][chunk][G 0..3][pt][kq][32 = 16*cp + lj][cb][s]
    // (channel 32*ch + 2*lj + cb of the workgroup's 64: adjacent channels in a lane's two blocks, see the tall layout)
    const int nsp = n >> 6, nl = ((n >> 5) & 1) * 16 + pair_lj(n), cb = pair_cb(n), G = (k >> 3) & 3;
    dst = u + (((size_t)nsp * nchunk + chunk) * 4 + G) * SU + (kq * 32 + nl) * 4 + cb * 2 + (k & 1);
  } else {
    const int nsp = n >> 5, nl = n & 31, G = (k >> 4) & 1, st = ((k >> 3) & 1) * 2 + (k & 1);
    dst = u + (((size_t)nsp * nchunk + chunk) * 2 + G) * SU + (kq * 32 + nl) * 4 + st;
  }
  for (int r = 0; r < 4; ++r) {
    const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]),
                u3 = t[r][2];
    put(dst, (r * 4 + 0) * 512, u0);
    put(dst, (r * 4 + 1) * 512, u1);
    put(dst, (r * 4 + 2) * 512, u2);
    put(dst, (r * 4 + 3) * 512, u3);
  }
}

__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int cin, int cout, int dgrad) {
  wino_pack_one(w, u, cin, cout, dgrad, blockIdx.x * blockDim.x + threadIdx.x);
}

constexpr int kPackJobs = 64;   // (3 modality branches x 9 layers x 2 directions = 54)
struct WinoPackTable {   // up to kPackJobs (layer, direction) jobs in one launch; job = blockIdx.y
  const float* w[kPackJobs];
  float* u[kPackJobs];
  int cin[kPackJobs], cout[kPackJobs], dgrad[kPackJobs];
};

__global__ void wino_pack_multi_kernel(WinoPackTable t) {
  const int j = blockIdx.y;
  wino_pack_one(t.w[j], t.u[j], t.cin[j], t.cout[j], t.dgrad[j], blockIdx.x * blockDim.x + threadIdx.x);
}

// Lane constants of the halo DMA: lane l of a piece fills slot 9 * (l / 9) + l % 9 of the piece's 7-pixel group, i.e.
// quarter-row c4 = l % 9 of pixel j = l / 9 (lane 63: first slot of the following pixel).  The pad slot (c4 = 8) is never
// read: it carries the pixel's last quarter-row again, so no lane needs a channel test.  Packed: j | (float offset << 8).
__device__ __forceinline__ int halo_lane_geometry(int lane) {
  const int j = (lane * 57) >> 9, c4 = lane - 9 * j;      // lane / 9, lane % 9
  return j | ((c4 < 8 ? c4 : 7) * 4) << 8;
}
// pooled tile: slot 11 * (l / 11) + l % 11 of the piece's 5-position group -- part c = l % 11: 0..7 gradient values,
// 8..9 argmax bytes, 10 pad (never read: carries part 9 again).  Packed: j | (byte offset inside the pixel << 8) | is_idx << 16.
__device__ __forceinline__ int pooled_lane_geometry(int lane) {
  const int j = (lane * 47) >> 9, c = lane - 11 * j;      // lane / 11, lane % 11
  const int cc = c < 10 ? c : 9;
  return j | (cc < 8 ? cc * 16 : (cc - 8) * 16) << 8 | (cc >= 8 ? 1 << 16 : 0);
}

// The compiler may not hoist what is derived from the result out of a loop: the per-lane slot geometry of a DMA piece is
// RECOMPUTED where the piece is issued (~10 VALU) instead of living in registers across the MFMA loop -- hoisted, the 64-bit
// address parts of three pieces spill to scratch, and a scratch reload in the loop waits for vmcnt(0), i.e. for every DMA.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// One piece of the halo tile of (image, region origin, chunk): pixels 7 * inst .. 7 * inst + 6 (row-major over the 18 x 18
// halo, columns in colpos order) -> 63 + 1 slots at byte inst * 1008.  `lq` = halo_lane_geometry(lane).
template <int KC, int HW>
__device__ __forceinline__ void dma_halo_piece(const float* __restrict__ in, const float* __restrict__ zeros, int img, int ry0,
                                               int rx0, int chunk, int inst, int lq, unsigned lds_byte_base) {
  inst = inst < HPIECES ? inst : HPIECES - 1;       // 8 waves x 6 pieces = 48 >= 47: the surplus repeats the last piece
  const int p = inst * HPX + (lq & 0xff);           // (the last piece's p runs to 329: halo "row 18", in the buffer's overhang)
  const int yy = (p * 3641) >> 16, xp = p - yy * PW;          // p / 18 exactly for p < 400
  const int xx = 2 * xp - (xp >= 9 ? 17 : 0);       // even columns first (see colpos)
  const int gy = ry0 - 1 + yy, gx = rx0 - 1 + xx;
  const bool ok = (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
  const float* base = in + (size_t)img * HW * HW * KC + chunk * 32;        // wave-uniform
  const float* src = ok ? base + (unsigned)((gy * HW + gx) * KC + (lq >> 8)) : zeros;
  dma16(src, lds_byte_base + (unsigned)inst * (HPX * 9 * 16));
}

// One piece of the POOLED input tile: positions 5 * inst .. 5 * inst + 4 of the 10 x 12 grid (positions 10, 11 of a row are
// padding), 11 slots each (8 of values, 2 of argmax bytes, 1 pad).  `lq` = pooled_lane_geometry(lane).
template <int KC, int HW>
__device__ __forceinline__ void dma_pooled_piece(const float* __restrict__ dz, const uint8_t* __restrict__ idx,
                                                 const float* __restrict__ zeros, int img, int ry0, int rx0, int chunk,
                                                 int inst, int lq, unsigned lds_byte_base) {
  constexpr int HP = HW / 2;
  inst = inst < UPIECES ? inst : UPIECES - 1;       // (8 waves x 3 pieces = 24 = UPIECES)
  const int p = inst * UPX + (lq & 0xff);           // (the last piece's lanes 55.. see "row 10": harmless, inside the buffer)
  const int prow = (p * 2731) >> 15, ppos = p - prow * UPW;   // p / 12 exactly for p < 200
  const int pr = ry0 / 2 - 1 + prow, pc = rx0 / 2 - 1 + ppos;
  const bool ok = (unsigned)pr < (unsigned)HP && (unsigned)pc < (unsigned)HP && ppos < 10;
  const unsigned o = (unsigned)((pr * HP + pc) * KC);                       // element offset of the pooled pixel
  const size_t pix = (size_t)img * HP * HP * KC + chunk * 32;               // wave-uniform
  const bool is_idx = (lq >> 16) != 0;
  const char* vsrc = reinterpret_cast<const char*>(dz + pix) + (size_t)(o * 4u + (unsigned)((lq >> 8) & 0xff));
  const char* isrc = reinterpret_cast<const char*>(idx + pix) + (size_t)(o + (unsigned)((lq >> 8) & 0xff));
  const void* src = !ok ? (const void*)zeros : (is_idx ? (const void*)isrc : (const void*)vsrc);
  dma16(src, lds_byte_base + (unsigned)inst * (UPX * 11 * 16));
}

// The 4x4 patch of one channel pair (h = 0: channels 2kq, 2kq+1 of the group; h = 1: 8 + those).  Plain input: 16
// ds_read_b64 from the halo tile.  Pooled input: the 3x3 pooled pixels under the patch and their argmax byte pairs; patch
// element (r,c) is pooled pixel ((r+1)>>1, (c+1)>>1) if its argmax byte equals the element's position
// ((r+1)&1)*2 + ((c+1)&1) inside the 2x2 window, else 0 (MaxPool backward).  `ibytes`: the lane's argmax pair for h = 0.
// (hipcc fuses pairs of these reads into ds_read2_b64, which banks mod 32 and is 2-way conflicted on this layout; keeping
// them apart costs address registers that the 256-register budget of the multi-chunk variants does not have.)
template <int IN_UNPOOL, bool B64 = false>
__device__ __forceinline__ void read_pair(float2 (&dn)[16], const float* base, const uint8_t* ibytes, int h) {
#if UGN_ABLATE & 16
  for (int e = 0; e < 16; ++e) dn[e] = make_float2((float)h + e, 1.f);
  return;
#endif
  if constexpr (!IN_UNPOOL && B64) {
    const unsigned a0 = lds_addr(base) + (unsigned)(32 * h);      // (h: the second channel pair sits 8 floats further)
#define UGN_RD(e_) dn[e_] = lds_read_b64<(((e_) >> 2) * PW + colpos((e_) & 3)) * CS * 4>(a0);
    UGN_RD(0) UGN_RD(1) UGN_RD(2) UGN_RD(3) UGN_RD(4) UGN_RD(5) UGN_RD(6) UGN_RD(7)
    UGN_RD(8) UGN_RD(9) UGN_RD(10) UGN_RD(11) UGN_RD(12) UGN_RD(13) UGN_RD(14) UGN_RD(15)
#undef UGN_RD
  } else if constexpr (!IN_UNPOOL) {
#pragma unroll
    for (int e = 0; e < 16; ++e) dn[e] = *reinterpret_cast<const float2*>(base + ((e >> 2) * PW + colpos(e & 3)) * CS + 8 * h);
  } else {
    float2 pv[9];
    unsigned iw[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      pv[q] = *reinterpret_cast<const float2*>(base + ((q / 3) * UPW + (q % 3)) * UCS + 8 * h);
      iw[q] = *reinterpret_cast<const uint16_t*>(ibytes + ((q / 3) * UPW + (q % 3)) * UCS * 4 + 8 * h);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = e >> 2, c = e & 3;
      const int q = ((r + 1) >> 1) * 3 + ((c + 1) >> 1);
      const unsigned pos = (((r + 1) & 1) << 1) | ((c + 1) & 1);
      dn[e].x = (iw[q] & 0xffu) == pos ? pv[q].x : 0.f;
      dn[e].y = (iw[q] >> 8) == pos ? pv[q].y : 0.f;
    }
  }
}

// Pooled input, one pooled ROW of the 3x3 pixels under the patch at a time (prow 0 -> patch row 0, 1 -> rows 1 and 2,
// 2 -> row 3): 6 registers of raw data live instead of 27, which is what lets the wide variant fit 256 registers.
template <int PROW>
__device__ __forceinline__ void read_pair_pooled_row(float2 (&dn)[16], const float* base, const uint8_t* ibytes) {
  float2 pv[3];
  unsigned iw[3];
#pragma unroll
  for (int qc = 0; qc < 3; ++qc) {
    pv[qc] = *reinterpret_cast<const float2*>(base + (PROW * UPW + qc) * UCS);
    iw[qc] = *reinterpret_cast<const uint16_t*>(ibytes + (PROW * UPW + qc) * UCS * 4);
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

// The same in two halves for the wide variant: the three value pairs of pooled row PROW are read with UNFUSED ds_read_b64
// (lds_read_b64: the compiler would fuse them into ds_read2_b64, 2-way bank conflicted on this layout) one point ahead of
// their use; pooled_row_select() waits for them and scatters them through the argmax bytes.
template <int PROW>
__device__ __forceinline__ void pooled_row_issue(float2 (&pv)[3], const float* base) {
  const unsigned a0 = lds_addr(base);
  pv[0] = lds_read_b64<(PROW * UPW + 0) * UCS * 4>(a0);
  pv[1] = lds_read_b64<(PROW * UPW + 1) * UCS * 4>(a0);
  pv[2] = lds_read_b64<(PROW * UPW + 2) * UCS * 4>(a0);
}
template <int PROW>
__device__ __forceinline__ void pooled_row_select(float2 (&dn)[16], float2 (&pv)[3], const uint8_t* ibytes) {
  unsigned iw[3];
#pragma unroll
  for (int qc = 0; qc < 3; ++qc) iw[qc] = *reinterpret_cast<const uint16_t*>(ibytes + (PROW * UPW + qc) * UCS * 4);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(iw[0]), "+v"(iw[1]), "+v"(iw[2]));
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

// 32 KB filter slice (16 KB of bf16 elements): linear in both spaces, 4 (2) pieces of 1 KB per wave (8 waves)
#ifndef UGN_POOLED_B64
#define UGN_POOLED_B64 1
#endif
#ifndef UGN_B64_DGRAD
#define UGN_B64_DGRAD 1
#endif
#ifndef UGN_ALT_PRIO
#define UGN_ALT_PRIO 1
#endif
#ifndef UGN_PREF_POOLED
#define UGN_PREF_POOLED 1     // touch the epilogue's act lines during the last group in the pooled data gradients too (-2.4 %)
#endif
// timing-only ablations (WRONG results): 1 no filter DMA, 2 no halo DMA, 4 no transform arithmetic, 8 no MFMA, 16 no patch reads
// from LDS, 32 no filter reads from LDS, 64 no group barrier
#ifndef UGN_ABLATE
#define UGN_ABLATE 0
#endif
template <bool BF>
__device__ __forceinline__ void dma_u_slice(const float* __restrict__ us, unsigned lds_byte_base, int tid, int wave) {
#if UGN_ABLATE & 1
  return;
#endif
#pragma unroll
  for (int q = 0; q < (BF ? 2 : 4); ++q) dma16(us + (q * 512 + tid) * 4, lds_byte_base + (unsigned)(q * 512 + wave * 64) * 16u);
}

// KC: GEMM K channels, NCF: output channels of the layer (a workgroup owns 32 of them), HW: image size
template <int KC, int NCF, int HW, int IN_UNPOOL, int EPI, int EFLAGS, bool BF = false>
__global__ __launch_bounds__(512, 2) void wino_kernel(const WinoJobs jt, const float* __restrict__ zeros) {
  // WIDE (NCF >= 64): the workgroup owns 64 output channels, a wave 16 tiles x 2 channel blocks, and a group is 8 input
  // channels (2 k-steps): per MFMA half the transform work, patch reads and halo traffic of the narrow variant.
  constexpr bool WIDE = wino_wide_ex(KC, NCF, BF, IN_UNPOOL != 0);
  constexpr bool ALT_PRIO = UGN_ALT_PRIO == 2 || (UGN_ALT_PRIO && EPI != EPI_DGRAD);
  // unfused ds_read_b64 patch reads (wino_common.h lds_read_b64): -1.5...4 % where the extra register pressure does not spill
  // into the loop: the forward kernels and the 128 -> 64 data gradient
  constexpr bool B64 = UGN_B64ASM && !IN_UNPOOL && !BF && (UGN_B64_DGRAD || EPI != EPI_DGRAD || (KC == 128 && NCF == 64));
  // packed transform arithmetic (wino_common.h pk_add): -1...7 % except where the aligned register pairs it needs push the
  // kernel into spilling inside the loop (the 128 -> 128 forward kernel; the bf16 variants stay as they were)
  constexpr int PK = (UGN_PK && !BF && !(KC == 128 && NCF == 128 && EPI != EPI_DGRAD)) ? 1 : 0;
  constexpr int PKE = PK;
  constexpr bool PB64 = UGN_POOLED_B64 && IN_UNPOOL && WIDE && !BF;   // unfused reads of the pooled tile (pooled_row_issue)
  constexpr int NB = WIDE ? 2 : 1;          // 16-channel output blocks per wave
  constexpr int NG = WIDE ? 4 : 2;          // channel groups per 32-channel chunk
  constexpr int GW = 32 / NG;               // input channels per group
  constexpr int NH = GW / 8;                // channel pairs per lane and group
  constexpr int NCHUNK = KC / 32, NSPLIT = NCF / (32 * NB);
  constexpr int RPX = HW / 16, RPI = RPX * RPX;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sU0 = smem + 2 * SIN;
  const unsigned sin_bytes = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem);
  const unsigned su_bytes = sin_bytes + 2u * SIN * 4u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: DMA targets stay in SGPRs
  const int tg = wave & 3, ch = wave >> 2;                     // tile group (region rows 4tg..4tg+3), 16-channel half
  const int lj = lane & 15, kq = lane >> 4;
  // A-side role of the lane: tile (tr, tc) of the wave's 2 x 8 tiles; B-side: output channel lj (+16), both: channels 4*kq..
  const int a_tr = lj >> 3, a_tc = lj & 7;
  // patch origin in the halo tile (plain) / first of the 3x3 pooled pixels in the pooled tile
  // The wave's two tile rows are 2 tile rows apart: that offsets their LDS images by 32 banks (of 64),
  // so with the 8 tile columns the 16 tile origins cover all 16 bank quads and a ds_read_b64 pass is conflict-free.
  const int trow0 = (tg & 1) + 4 * (tg >> 1);
  constexpr int TRSTEP = 2;
  const int a_trow = trow0 + TRSTEP * a_tr;                                    // tile row (0..7) of the lane's A-side tile
  const int pbase = IN_UNPOOL ? (a_trow * UPW + a_tc) * UCS + 2 * kq : (2 * a_trow * PW + a_tc) * CS + 2 * kq;
  const int ibase = ((a_trow * UPW + a_tc) * UCS + 32) * 4 + 2 * kq;           // BYTE offset of the argmax pair (pooled tile)
  const int ubase = (kq * 32 + ch * 16 + lj) * 4;

  const int hlq = IN_UNPOOL ? pooled_lane_geometry(lane) : halo_lane_geometry(lane);   // the lane's place in a DMA piece
  int item = blockIdx.x;
  const int nitems = jt.start[kMaxJobs];
  if (item >= nitems) return;
  // (what depends on the piece is recomputed where the piece is issued, see opaque(): ~15 VALU per piece, but no live
  //  registers across the MFMA loop in a kernel that sits at the 256-register limit)
  // item -> job (wave-uniform: the table is read with scalar loads, once per item, at the item boundary)
  int jb = wino_job_of(jt, item), lit = item - jt.start[jb];   // job and job-local number of the current item
  auto u_slice = [&](const float* upk, int lit_, int chunk, int G) {
    return upk + (((size_t)(lit_ % NSPLIT) * NCHUNK + chunk) * NG + G) * (BF ? SU / 2 : SU);
  };
  // ---- prologue: halo(item, chunk 0) -> sIn[0]; U(item, 0, 0) -> sU[0]
  {
    const int region = lit / NSPLIT, img = region / RPI, rrem = region % RPI;
    const float* in = jt.job[jb].in;
    const uint8_t* in_idx = jt.job[jb].in_idx;
    if constexpr (IN_UNPOOL) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        dma_pooled_piece<KC, HW>(in, in_idx, zeros, img, (rrem / RPX) * 16, (rrem % RPX) * 16, 0, wave * 3 + j, hlq, sin_bytes);
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        dma_halo_piece<KC, HW>(in, zeros, img, (rrem / RPX) * 16, (rrem % RPX) * 16, 0, wave * 6 + j, hlq, sin_bytes);
    }
    dma_u_slice<BF>(u_slice(jt.job[jb].upk, lit, 0, 0), su_bytes, tid, wave);
  }
  int ibuf = 0, ubuf = 0;
  float V[16][2 * NH]; // transformed patch (4 / 2 channels) of the group about to be multiplied
  uint32_t Vp[16][NH]; // BF: the same, rounded to bf16 as it is produced (channel pairs packed: half the registers)
  auto setV = [&](int pt, int h, float x, float y) {
    if constexpr (BF) Vp[pt][h] = pk_bf16(x, y);
    else { V[pt][2 * h] = x; V[pt][2 * h + 1] = y; }
  };
  bool first = true;   // only the very first group of the workgroup transforms its patch un-pipelined
#if UGN_STAMPS
  int stamp_n = 0;
#endif

  // Single-chunk narrow layers: the two filter slices of (job, nsp) stay RESIDENT in the two slots; every item of this
  // workgroup has the same nsp when the grid is a multiple of NSPLIT (each job's items are one).  res0 / res1 = the job whose
  // slice 0 / 1 sits in slot 0 / 1: a slice is fetched only when the group about to need it belongs to another job.
  const bool u_resident = NCHUNK == 1 && NG == 2 && gridDim.x % NSPLIT == 0;
  int res0 = jb, res1 = -1;
  for (; item < nitems; item += gridDim.x) {
    const int next_item = item + gridDim.x;
    // the next item's job: its first halo tile and filter slice are fetched during this item's last chunk
    const bool more = next_item < nitems;
    const int jn = more ? wino_job_of(jt, next_item) : jb, nlit = more ? next_item - jt.start[jn] : lit;
    f32x4 acc[NB][16];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int pt = 0; pt < 16; ++pt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[cb][pt][r] = 0.f;

#pragma unroll((WIDE || NCHUNK > 2) ? 1 : NCHUNK)   // (fully unrolled, the wide variants hoist DMA addresses and spill)
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool last_chunk = chunk + 1 == NCHUNK;
      const bool has_next = !last_chunk || next_item < nitems;
      // no next stage (last chunk of the last item): the prefetches re-fetch the current stage into the free buffers
      // instead of branching around the DMA (keeps the MFMA stream one basic block)
      const bool to_next = last_chunk && has_next;      // the prefetches of this chunk belong to the next item
      const int n_chunk = last_chunk ? (has_next ? 0 : chunk) : chunk + 1;
      const int n_lit = to_next ? nlit : lit;
      const int n_region = n_lit / NSPLIT;
      const int nx_job = to_next ? jn : jb;             // (ONE table entry is read: selecting between two pointers instead
      const float* in = jt.job[nx_job].in;               //  makes the compiler form both sets of DMA addresses)
      const uint8_t* in_idx = jt.job[nx_job].in_idx;
      const float* nx_upk = jt.job[nx_job].upk;
      const int n_img = n_region / RPI, n_rrem = n_region % RPI;
      const int n_ry0 = (n_rrem / RPX) * 16, n_rx0 = (n_rrem % RPX) * 16;
      const float* sIn = smem + ibuf * SIN;
      const float* sInNext = smem + (ibuf ^ 1) * SIN;
#pragma unroll
      for (int G = 0; G < NG; ++G) {
#if UGN_STAMPS
        if (blockIdx.x == 0 && lane == 0 && stamp_n < 2046) ugn_stamp_buf[wave * 2048 + stamp_n++] = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every DMA issued so far has landed (all are >= 1 group old)
#if !(UGN_ABLATE & 64)
        __syncthreads();                                    // ... and is visible; the other buffers have no readers left
#endif
#if UGN_STAMPS
        if (blockIdx.x == 0 && lane == 0 && stamp_n < 2046) ugn_stamp_buf[wave * 2048 + stamp_n++] = __builtin_amdgcn_s_memtime();
#endif
        const float* sU = sU0 + ubuf * SU;
        // filter slice of the next group -> the other buffer, while this group computes
        // (single-chunk layers: the two slices of the workgroup's 32 output channels stay resident after the first item)
        if (G + 1 < NG) {
          if (!u_resident || res1 != jb) {
            dma_u_slice<BF>(u_slice(jt.job[jb].upk, lit, chunk, G + 1), su_bytes + (unsigned)(ubuf ^ 1) * SU * 4u, tid, wave);
            res1 = jb;
          }
        } else {
          if (!u_resident || res0 != nx_job) {
            dma_u_slice<BF>(u_slice(nx_upk, n_lit, n_chunk, 0), su_bytes + (unsigned)(ubuf ^ 1) * SU * 4u, tid, wave);
            res0 = nx_job;
          }
        }
        if (first) {
          first = false;
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            float2 d[16], t[16];
            read_pair<IN_UNPOOL, B64>(d, sIn + pbase, reinterpret_cast<const uint8_t*>(sIn) + ibase, h);
            if constexpr (B64) patch_wait(d);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              t[0 + c] = make_float2(d[0 + c].x - d[8 + c].x, d[0 + c].y - d[8 + c].y);
              t[4 + c] = make_float2(d[4 + c].x + d[8 + c].x, d[4 + c].y + d[8 + c].y);
              t[8 + c] = make_float2(d[8 + c].x - d[4 + c].x, d[8 + c].y - d[4 + c].y);
              t[12 + c] = make_float2(d[4 + c].x - d[12 + c].x, d[4 + c].y - d[12 + c].y);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              setV(r * 4 + 0, h, t[r * 4 + 0].x - t[r * 4 + 2].x, t[r * 4 + 0].y - t[r * 4 + 2].y);
              setV(r * 4 + 1, h, t[r * 4 + 1].x + t[r * 4 + 2].x, t[r * 4 + 1].y + t[r * 4 + 2].y);
              setV(r * 4 + 2, h, t[r * 4 + 2].x - t[r * 4 + 1].x, t[r * 4 + 2].y - t[r * 4 + 1].y);
              setV(r * 4 + 3, h, t[r * 4 + 1].x - t[r * 4 + 3].x, t[r * 4 + 1].y - t[r * 4 + 3].y);
            }
          }
        }
        // the NEXT group's patch comes from this chunk (G == 0) or from the next chunk's halo, which landed a group ago
        constexpr bool tnext = true;   // (at the very end this transforms a re-fetched tile that nobody consumes)
        const float* sNx = (G + 1 < NG ? sIn + GW * (G + 1) : sInNext) + pbase;
        const uint8_t* sNi = reinterpret_cast<const uint8_t*>(G + 1 < NG ? sIn : sInNext) + (G + 1 < NG ? GW * (G + 1) : 0) + ibase;
        // The next group's transformed patch is built in the shadow of this group's MFMAs and written straight into the
        // V registers of points that have already been multiplied (V[4r..4r+3] are dead once point 4r+3 is done), so only
        // one V set plus the row-pass temporaries are live: the kernel must fit 256 arch VGPRs beside 128 accumulators.
        // (WIDE has a single channel pair per group: its row pass runs in place on dn, which frees 32 registers)
        float2 dn[16], tn0s[16], tn1[16];
        float2 pvr[3];   // (PB64) raw value pairs of the pooled row in flight
        float2 (&tn0)[16] = *(WIDE ? &dn : &tn0s);
        auto rowpass = [&](float2 (&tn)[16], int c) {
          if (UGN_ABLATE & 4) { tn[0 + c] = dn[0 + c]; tn[4 + c] = dn[4 + c]; tn[8 + c] = dn[8 + c]; tn[12 + c] = dn[12 + c]; return; }
          const float2 d0 = dn[0 + c], d1 = dn[4 + c], d2 = dn[8 + c], d3 = dn[12 + c];
          tn[0 + c] = pk_sub<PK>(d0, d2);
          tn[4 + c] = pk_add<PK>(d1, d2);
          tn[8 + c] = pk_sub<PK>(d2, d1);
          tn[12 + c] = pk_sub<PK>(d1, d3);
        };
        auto colpass = [&](int r) {
          if (UGN_ABLATE & 4) {
            for (int j = 0; j < 4; ++j) { setV(r * 4 + j, 0, tn0[r * 4 + j].x, tn0[r * 4 + j].y); if constexpr (NH == 2) setV(r * 4 + j, 1, tn1[r * 4 + j].x, tn1[r * 4 + j].y); }
            return;
          }
          auto put = [&](int pt, int h, float2 v) { setV(pt, h, v.x, v.y); };
          put(r * 4 + 0, 0, pk_sub<PK>(tn0[r * 4 + 0], tn0[r * 4 + 2]));
          put(r * 4 + 1, 0, pk_add<PK>(tn0[r * 4 + 1], tn0[r * 4 + 2]));
          put(r * 4 + 2, 0, pk_sub<PK>(tn0[r * 4 + 2], tn0[r * 4 + 1]));
          put(r * 4 + 3, 0, pk_sub<PK>(tn0[r * 4 + 1], tn0[r * 4 + 3]));
          if constexpr (NH == 2) {
            put(r * 4 + 0, 1, pk_sub<PK>(tn1[r * 4 + 0], tn1[r * 4 + 2]));
            put(r * 4 + 1, 1, pk_add<PK>(tn1[r * 4 + 1], tn1[r * 4 + 2]));
            put(r * 4 + 2, 1, pk_sub<PK>(tn1[r * 4 + 2], tn1[r * 4 + 1]));
            put(r * 4 + 3, 1, pk_sub<PK>(tn1[r * 4 + 1], tn1[r * 4 + 3]));
          }
        };
        // points are multiplied in PAIRS with their 4 k-steps interleaved (pt0 s0, pt1 s0, pt0 s1, ...): consecutive
        // MFMAs on one accumulator would each wait out the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32
        float4 u[2][2];
        uint2 ub[2][2];   // BF: the same 4 elements as bf16 (8 bytes per point)
        if constexpr (BF) {
          ub[0][0] = *reinterpret_cast<const uint2*>(sU + ubase / 2);
          ub[0][1] = *reinterpret_cast<const uint2*>(sU + ubase / 2 + 256);
        } else {
          u[0][0] = *reinterpret_cast<const float4*>(sU + ubase);
          u[0][1] = *reinterpret_cast<const float4*>(sU + ubase + 512);
          if (UGN_ABLATE & 32) { u[1][0] = u[0][0]; u[1][1] = u[0][1]; }
        }
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
          const int cu = pp & 1, nu = cu ^ 1;
          // The two waves of a SIMD are not served alike: with equal priority the older one (waves 0-3) ends a group ~1300
          // cycles before the younger and waits at the barrier (tools/stamps.py: 4600 / 5900 cycles inside an a6 forward
          // group).  Taking turns pair by pair brings them to 5150 / 5650: -3...4 % on the forward kernels; the data
          // gradients lose 0-4 % with it (their epilogue operands), so they keep equal priorities.
          if constexpr (ALT_PRIO) {
            if ((wave >= 4) == ((pp & 1) != 0)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
          }
          if (pp < 7 && !(UGN_ABLATE & 32)) {
            if constexpr (BF) {
              ub[nu][0] = *reinterpret_cast<const uint2*>(sU + ubase / 2 + (2 * pp + 2) * 256);
              ub[nu][1] = *reinterpret_cast<const uint2*>(sU + ubase / 2 + (2 * pp + 3) * 256);
            } else {
              u[nu][0] = *reinterpret_cast<const float4*>(sU + ubase + (2 * pp + 2) * 512);
              u[nu][1] = *reinterpret_cast<const float4*>(sU + ubase + (2 * pp + 3) * 512);
            }
          }
          if constexpr (BF && WIDE) {   // ub = {cb0 (s0, s1), cb1 (s0, s1)}: one bf16 MFMA per block, k-slots 2, 3 empty
            const uint32_t a0 = Vp[2 * pp][0], a1 = Vp[2 * pp + 1][0];
            // B is the register pair as loaded: k-slots 0, 1 = block 0's filters, 2, 3 = block 1's; A selects the block
            acc[0][2 * pp] = mfma_bf16(a0, 0u, ub[cu][0].x, ub[cu][0].y, acc[0][2 * pp]);
            acc[0][2 * pp + 1] = mfma_bf16(a1, 0u, ub[cu][1].x, ub[cu][1].y, acc[0][2 * pp + 1]);
            acc[1][2 * pp] = mfma_bf16(0u, a0, ub[cu][0].x, ub[cu][0].y, acc[1][2 * pp]);
            acc[1][2 * pp + 1] = mfma_bf16(0u, a1, ub[cu][1].x, ub[cu][1].y, acc[1][2 * pp + 1]);
          } else if constexpr (BF) {    // the lane's 4 channels in one bf16 MFMA
            acc[0][2 * pp] = mfma_bf16(Vp[2 * pp][0], Vp[2 * pp][NH - 1], ub[cu][0].x, ub[cu][0].y, acc[0][2 * pp]);
            acc[0][2 * pp + 1] = mfma_bf16(Vp[2 * pp + 1][0], Vp[2 * pp + 1][NH - 1], ub[cu][1].x, ub[cu][1].y, acc[0][2 * pp + 1]);
          } else if constexpr (WIDE) {   // u float4 = {cb0 s0, cb0 s1, cb1 s0, cb1 s1}
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) {
                acc[cb][2 * pp] = mfma16(V[2 * pp][st], u[cu][0][cb * 2 + st], acc[cb][2 * pp]);
                acc[cb][2 * pp + 1] = mfma16(V[2 * pp + 1][st], u[cu][1][cb * 2 + st], acc[cb][2 * pp + 1]);
              }
          } else {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
              acc[0][2 * pp] = mfma16(V[2 * pp][st], u[cu][0][st], acc[0][2 * pp]);
              acc[0][2 * pp + 1] = mfma16(V[2 * pp + 1][st], u[cu][1][st], acc[0][2 * pp + 1]);
            }
          }
#pragma unroll
          for (int half = 0; half < 2; ++half) {
          const int pt = 2 * pp + half;
          constexpr bool ROWWISE = IN_UNPOOL != 0;   // pooled tile: the patch arrives one pooled row per point
          if constexpr (ROWWISE && PB64) {
            if (pt == 0) pooled_row_issue<0>(pvr, sNx);
            if (pt == 1) { pooled_row_select<0>(dn, pvr, sNi); pooled_row_issue<1>(pvr, sNx); }
            if (pt == 2) { pooled_row_select<1>(dn, pvr, sNi); pooled_row_issue<2>(pvr, sNx); }
            if (pt == 3) pooled_row_select<2>(dn, pvr, sNi);
          } else if constexpr (ROWWISE) {
            if (pt == 0) read_pair_pooled_row<0>(dn, sNx, sNi);
            if (pt == 1) read_pair_pooled_row<1>(dn, sNx, sNi);
            if (pt == 2) read_pair_pooled_row<2>(dn, sNx, sNi);
            if constexpr (NH == 2) {   // second channel pair (+8 channels = +8 floats / +8 argmax bytes)
              if (pt == 5) read_pair_pooled_row<0>(dn, sNx + 8, sNi + 8);
              if (pt == 6) read_pair_pooled_row<1>(dn, sNx + 8, sNi + 8);
              if (pt == 7) read_pair_pooled_row<2>(dn, sNx + 8, sNi + 8);
            }
          } else {
            if (tnext && (pt == 0 || (NH == 2 && pt == 3))) read_pair<IN_UNPOOL, B64>(dn, sNx, sNi, pt == 0 ? 0 : 1);   // channels 2kq.. / 8+2kq..
          }
          // halo of the next stage: all pieces of this wave during the FIRST group of the chunk, so that they are
          // a full group old at the next barrier and the next chunk's first transform can be pipelined as well
          if constexpr (IN_UNPOOL) {
            if (!(UGN_ABLATE & 2) && G == 0 && pt >= 1 && pt <= 3)
              dma_pooled_piece<KC, HW>(in, in_idx, zeros, n_img, n_ry0, n_rx0, n_chunk, wave * 3 + (pt - 1), opaque(hlq),
                                       sin_bytes + (unsigned)(ibuf ^ 1) * SIN * 4u);
          } else {
            if (!(UGN_ABLATE & 2) && G == 0 && pt >= 1 && pt < 7)   // early in the group: the pieces must have landed by the group's end
              dma_halo_piece<KC, HW>(in, zeros, n_img, n_ry0, n_rx0, n_chunk, wave * 6 + (pt - 1), opaque(hlq),
                                     sin_bytes + (unsigned)(ibuf ^ 1) * SIN * 4u);
          }
          if (tnext) {
            constexpr int RP0 = ROWWISE ? 3 : 1;
            if constexpr (B64) {
              if (pt == RP0 || (NH == 2 && pt == 4)) patch_wait(dn);
            }
            if (pt == RP0) { rowpass(tn0, 0); rowpass(tn0, 1); }
            if (pt == RP0 + 1) { rowpass(tn0, 2); rowpass(tn0, 3); }   // pair 0 done before pair 1 is loaded at point 3
            if (NH == 2 && pt == (ROWWISE ? 8 : 4)) { rowpass(tn1, 0); rowpass(tn1, 1); }
            if (NH == 2 && pt == (ROWWISE ? 9 : 5)) { rowpass(tn1, 2); rowpass(tn1, 3); }
            // The data-gradient epilogue reads act / addend at the item's output pixels; those loads are a dependent round
            // trip to HBM per item.  Touch one dword of every line they will read while the last group still computes
            // (LDS-DMA into a per-wave dump: no registers, nothing waits for it): the epilogue then hits L2.
            if constexpr (EPI == EPI_DGRAD && (EFLAGS & 3) != 0 && (!IN_UNPOOL || UGN_PREF_POOLED)) {
              if (pt == 9 && G == NG - 1 && last_chunk) {
                const int p_region = lit / NSPLIT, p_nsp = lit % NSPLIT;
                const int p_img = p_region / RPI, p_rrem = p_region % RPI;
                const int t = lane >> 2, q = lane & 3;
                const int oy = (p_rrem / RPX) * 16 + 2 * (trow0 + TRSTEP * (t >> 3)) + (q >> 1);
                const int ox = (p_rrem % RPX) * 16 + 2 * (t & 7) + (q & 1);
                const size_t o = (((size_t)p_img * HW + oy) * HW + ox) * NCF + p_nsp * (32 * NB) + ch * (16 * NB);
                const unsigned dump = sin_bytes + (unsigned)((2 * SIN + 2 * SU) * 4) + (unsigned)wave * 256u;
                if constexpr (EFLAGS & 1) dma4(jt.job[jb].act + o, dump);
                if constexpr (EFLAGS & 2) dma4(jt.job[jb].addend + o, dump);
              }
            }
            constexpr bool LATE = ROWWISE && NH == 2;   // both row passes end at point 9
            if (pt == (LATE ? 10 : 6)) colpass(0);
            if (pt == (LATE ? 11 : 8)) colpass(1);
            if (pt == 12) colpass(2);
            if (pt == 15) colpass(3);
          }
          }
        }
        if constexpr (ALT_PRIO) __builtin_amdgcn_s_setprio(0);
        ubuf ^= 1;
      }
      ibuf ^= 1;
    }

    // ---- output transform + epilogue: lane holds tiles 4*kq + r (r = 0..3) x channels {lj, 16 + lj}, all 16 points
    const int region = lit / NSPLIT, nsp = lit % NSPLIT;
    const int img = region / RPI, rrem = region % RPI;
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
    const int ry0 = (rrem / RPX) * 16, rx0 = (rrem % RPX) * 16;
    // the lane's first channel: block cb adds 16 channels, or 1 with the paired mapping (wino_common.h pair_lj / pair_cb)
    const int co = nsp * (32 * NB) + ch * (16 * NB) + ((NB == 2 && UGN_EPI_PAIR) ? 2 * lj : lj);
    float y[NB][4][4];     // [block][tile r][output (a,b) row-major]
    unsigned o[4][4];      // element offsets inside the image: the image base is wave-uniform and rides in SGPRs
    wino_out_transform<NB, PKE>(acc, y);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ti = 4 * kq + r, tr = ti >> 3, tc = ti & 7;
      const int oy = ry0 + 2 * (trow0 + TRSTEP * tr), ox = rx0 + 2 * tc;
      if constexpr (EPI == EPI_LRELU_POOL) {
        constexpr int HP = HW / 2;
        o[r][0] = (unsigned)(((oy / 2) * HP + ox / 2) * NCF + co);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[r][q] = (unsigned)(((oy + (q >> 1)) * HW + ox + (q & 1)) * NCF + co);
      }
    }
    wino_epilogue<NB, EPI, EFLAGS>(y, o, out, out_idx, act, addend, raw_out, sm_m, sm_g);
    jb = jn; lit = nlit;
  }
}

template <int KC, int NCF, int HW, int IN_UNPOOL, int EPI, int EFLAGS, bool BF = false>
int launch_wino(const WinoJob* jobs, const int* n, int njobs, hipStream_t st) {
  auto kern = wino_kernel<KC, NCF, HW, IN_UNPOOL, EPI, EFLAGS, BF>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) { ugn_set_error("wino: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_done = true;
  }
  const float* zeros = zero_block();
  if (!zeros) { ugn_set_error("wino: cannot allocate the zero block"); return UGN_EINVAL; }
  constexpr int per_img = (HW / 16) * (HW / 16) * (NCF / (wino_wide_ex(KC, NCF, BF, IN_UNPOOL != 0) ? 64 : 32));
  WinoJobs jt;
  const int nitems = make_job_table(jt, jobs, n, njobs, per_img);
  const int grid = nitems < kGrid ? nitems : kGrid;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, st, jt, zeros);
  UGN_CHECK_LAUNCH("wino");
  return 0;
}

template <int KC, int NCF, int HW, int IN_UNPOOL>
int launch_wino_dgrad(const WinoJob* jobs, const int* n, int njobs, bool bf, hipStream_t st) {
  const int flags = (jobs[0].act ? 1 : 0) | (jobs[0].addend ? 2 : 0) | (jobs[0].raw_out ? 4 : 0) | (jobs[0].smax_m ? 8 : 0);
  if (bf) {   // (the BF = true instantiations -- bf16-rounded Winograd-domain operands on fp32 tensors, rounds 1-4 -- are no longer built)
    ugn_set_error("the bf16-operand Winograd kernels (fp32 tensors, 'bf16w') were retired in round 5: conv_precision='bf16' is the configs[4] path"); return UGN_EINVAL;
  }
  if constexpr (KC == 128 && NCF == 64 && HW == 16 && !IN_UNPOOL)   // a5: act + routed set-max gradient
    if (flags == 9) return launch_wino<KC, NCF, HW, IN_UNPOOL, EPI_DGRAD, 9>(jobs, n, njobs, st);
#define UGN_WDG(F_) \
  case F_:          \
    return launch_wino<KC, NCF, HW, IN_UNPOOL, EPI_DGRAD, F_>(jobs, n, njobs, st);
  switch (flags) {
    UGN_WDG(0) UGN_WDG(1) UGN_WDG(3) UGN_WDG(5) UGN_WDG(7)
    default: break;
  }
#undef UGN_WDG
  ugn_set_error("ugn_conv3x3_dgrad_wino: unsupported epilogue combination %d (addend/raw_out need act)", flags);
  return UGN_EINVAL;
}

int dispatch_fwd(const WinoJob* jobs, const int* n, int njobs, int hw, int cin, int cout, int pool, bool bf, hipStream_t st) {
  if (bf) { ugn_set_error("the bf16-operand Winograd kernels (fp32 tensors, 'bf16w') were retired in round 5: conv_precision='bf16' is the configs[4] path"); return UGN_EINVAL; }
  if (wino_tall(cin, cout)) return launch_tall(0, jobs, n, njobs, hw, cin, pool != 0, bf, st);   // (the filter layout differs)
#define WF(KC_, NC_, HW_, P_)                                                                                     \
  if (cin == KC_ && cout == NC_ && hw == HW_ && (pool != 0) == (P_ != 0))                                         \
    return launch_wino<KC_, NC_, HW_, 0, P_ ? EPI_LRELU_POOL : EPI_LRELU, 0>(jobs, n, njobs, st);
  WF(32, 32, 64, 1) WF(32, 64, 32, 0) WF(64, 64, 32, 1) WF(64, 128, 16, 0) WF(128, 128, 16, 0)
#undef WF
  ugn_set_error("ugn_conv3x3_fwd_wino: unsupported shape cin=%d cout=%d hw=%d pool=%d", cin, cout, hw, pool);
  return UGN_EINVAL;
}

int dispatch_dgrad(const WinoJob* jobs, const int* n, int njobs, int hw, int cin, int cout, int unpool, bool bf, hipStream_t st) {
  for (int j = 1; j < njobs; ++j) {   // one kernel instantiation serves all jobs: they must use the same epilogue operands
    const int f0 = (jobs[0].act ? 1 : 0) | (jobs[0].addend ? 2 : 0) | (jobs[0].raw_out ? 4 : 0) | (jobs[0].smax_m ? 8 : 0);
    const int f1 = (jobs[j].act ? 1 : 0) | (jobs[j].addend ? 2 : 0) | (jobs[j].raw_out ? 4 : 0) | (jobs[j].smax_m ? 8 : 0);
    if (f0 != f1) {
      ugn_set_error("ugn_conv3x3_dgrad_wino: all jobs of a launch need the same set of act/addend/raw_out (%d vs %d in job %d)", f0, f1, j);
      return UGN_EINVAL;
    }
  }
  if (wino_tall(cout, cin)) return launch_tall(1, jobs, n, njobs, hw, cout, unpool, bf, st);
#define WD(CI_, CO_, HW_, U_)                                 \
  if (cin == CI_ && cout == CO_ && hw == HW_ && unpool == U_) \
    return launch_wino_dgrad<CO_, CI_, HW_, U_>(jobs, n, njobs, bf, st);
  WD(32, 32, 64, 1) WD(32, 64, 32, 0) WD(64, 64, 32, 1) WD(64, 128, 16, 0) WD(128, 128, 16, 0)
#undef WD
  ugn_set_error("ugn_conv3x3_dgrad_wino: unsupported shape cin=%d cout=%d hw=%d unpool=%d", cin, cout, hw, unpool);
  return UGN_EINVAL;
}

}  // namespace

#if UGN_STAMPS
extern "C" int ugn_debug_stamps(unsigned long long* host_dst, int n) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ugn_stamp_buf), (size_t)n * sizeof(unsigned long long));
}
#endif

// 256 B of zeros in HBM: the LDS-DMA source for halo lanes outside the image.  Allocated once per process, never written.
const float* ugn_wino::zero_block() {
  static float* z = nullptr;
  if (!z) {
    float* p = nullptr;
    if (hipMalloc((void**)&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) return nullptr;
    z = p;
  }
  return z;
}

extern "C" int ugn_wino_pack(const float* w_hwio, float* u_packed, int cin, int cout, int dgrad, void* stream) {
  UGN_REQUIRE(w_hwio && u_packed, "ugn_wino_pack: null pointer");
  UGN_REQUIRE(cin % 32 == 0 && cout % 32 == 0 && cin > 0 && cout > 0, "ugn_wino_pack: channels must be multiples of 32");
  const int total = cin * cout;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_hwio, u_packed, cin,
                     cout, dgrad);
  UGN_CHECK_LAUNCH("wino_pack");
  return 0;
}

extern "C" int ugn_wino_pack_multi(const float* const* w_hwio_host, float* const* u_packed_host, const int* cin_host,
                                   const int* cout_host, const int* dgrad_host, int njobs, void* stream) {
  UGN_REQUIRE(w_hwio_host && u_packed_host && cin_host && cout_host && dgrad_host, "ugn_wino_pack_multi: null pointer");
  UGN_REQUIRE(njobs >= 1 && njobs <= kPackJobs, "ugn_wino_pack_multi: njobs must be 1..%d (got %d)", kPackJobs, njobs);
  WinoPackTable t = {};
  int maxe = 0;
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(w_hwio_host[j] && u_packed_host[j] && cin_host[j] % 32 == 0 && cout_host[j] % 32 == 0 && cin_host[j] > 0 &&
                    cout_host[j] > 0, "ugn_wino_pack_multi: bad job %d", j);
    t.w[j] = w_hwio_host[j]; t.u[j] = u_packed_host[j];
    t.cin[j] = cin_host[j]; t.cout[j] = cout_host[j]; t.dgrad[j] = dgrad_host[j];
    if (cin_host[j] * cout_host[j] > maxe) maxe = cin_host[j] * cout_host[j];
  }
  hipLaunchKernelGGL(wino_pack_multi_kernel, dim3((maxe + 255) / 256, njobs), dim3(256), 0, (hipStream_t)stream, t);
  UGN_CHECK_LAUNCH("wino_pack_multi");
  return 0;
}

static int fwd_one(const float* in, const float* u_packed, float* out, uint8_t* out_idx, int n, int hw, int cin, int cout,
                   int pool, bool bf, void* stream) {
  UGN_REQUIRE(in && u_packed && out && n > 0, "ugn_conv3x3_fwd_wino: null pointer or n <= 0");
  UGN_REQUIRE(!pool || out_idx, "ugn_conv3x3_fwd_wino: pool needs out_idx");
  const WinoJob job = {in, nullptr, u_packed, out, out_idx, nullptr, nullptr, nullptr};
  return dispatch_fwd(&job, &n, 1, hw, cin, cout, pool, bf, (hipStream_t)stream);
}
extern "C" int ugn_conv3x3_fwd_wino(const float* in, const float* u_packed, float* out, uint8_t* out_idx, int n, int hw,
                                    int cin, int cout, int pool, void* stream) {
  return fwd_one(in, u_packed, out, out_idx, n, hw, cin, cout, pool, false, stream);
}

static int fwd_multi(const float* const* in, const float* const* u_packed, float* const* out, uint8_t* const* out_idx,
                     const int* n, int njobs, int hw, int cin, int cout, int pool, bool bf, void* stream) {
  UGN_REQUIRE(in && u_packed && out && n, "ugn_conv3x3_fwd_wino_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_conv3x3_fwd_wino_multi: njobs must be 1..%d (got %d)", kMaxJobs, njobs);
  WinoJob jobs[kMaxJobs];
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(in[j] && u_packed[j] && out[j] && n[j] > 0, "ugn_conv3x3_fwd_wino_multi: null pointer or n <= 0 in job %d", j);
    UGN_REQUIRE(!pool || (out_idx && out_idx[j]), "ugn_conv3x3_fwd_wino_multi: pool needs out_idx");
    jobs[j] = {in[j], nullptr, u_packed[j], out[j], pool ? out_idx[j] : nullptr, nullptr, nullptr, nullptr};
  }
  return dispatch_fwd(jobs, n, njobs, hw, cin, cout, pool, bf, (hipStream_t)stream);
}
static int fwd_pair(const float* const* in, const float* const* u_packed, float* const* out, uint8_t* const* out_idx,
                    const int* n, int hw, int cin, int cout, int pool, bool bf, void* stream) {
  return fwd_multi(in, u_packed, out, out_idx, n, 2, hw, cin, cout, pool, bf, stream);
}
extern "C" int ugn_conv3x3_fwd_wino_multi(const float* const* in, const float* const* u_packed, float* const* out,
                                          uint8_t* const* out_idx, const int* n, int njobs, int hw, int cin, int cout, int pool,
                                          int bf16, void* stream) {
  return fwd_multi(in, u_packed, out, out_idx, n, njobs, hw, cin, cout, pool, bf16 != 0, stream);
}
extern "C" int ugn_conv3x3_fwd_wino_pair(const float* const* in, const float* const* u_packed, float* const* out,
                                         uint8_t* const* out_idx, const int* n, int hw, int cin, int cout, int pool,
                                         void* stream) {
  return fwd_pair(in, u_packed, out, out_idx, n, hw, cin, cout, pool, false, stream);
}

static int dgrad_one(const float* dz, const uint8_t* dz_idx, const float* u_packed, const float* act, const float* addend,
                     float* out, float* raw_out, int n, int hw, int cin, int cout, bool bf, void* stream) {
  UGN_REQUIRE(dz && u_packed && out && n > 0, "ugn_conv3x3_dgrad_wino: null pointer or n <= 0");
  const WinoJob job = {dz, dz_idx, u_packed, out, nullptr, act, addend, raw_out};
  return dispatch_dgrad(&job, &n, 1, hw, cin, cout, dz_idx != nullptr, bf, (hipStream_t)stream);
}
extern "C" int ugn_conv3x3_dgrad_wino(const float* dz, const uint8_t* dz_idx, const float* u_packed, const float* act,
                                      const float* addend, float* out, float* raw_out, int n, int hw, int cin, int cout,
                                      void* stream) {
  return dgrad_one(dz, dz_idx, u_packed, act, addend, out, raw_out, n, hw, cin, cout, false, stream);
}

extern "C" int ugn_conv3x3_dgrad_wino_routed(const float* dz, const float* u_packed, const float* act, const float* smax_m,
                                             const float* smax_g, int frames, float* out, int n, int hw, int cin, int cout,
                                             void* stream) {
  UGN_REQUIRE(dz && u_packed && act && smax_m && smax_g && out && n > 0, "ugn_conv3x3_dgrad_wino_routed: null pointer or n <= 0");
  UGN_REQUIRE(frames > 0 && n % frames == 0, "ugn_conv3x3_dgrad_wino_routed: n (%d) must be a multiple of frames (%d)", n, frames);
  WinoJob job = {dz, nullptr, u_packed, out, nullptr, act, nullptr, nullptr};
  job.smax_m = smax_m;
  job.smax_g = smax_g;
  job.frames = frames;
  return dispatch_dgrad(&job, &n, 1, hw, cin, cout, 0, false, (hipStream_t)stream);
}

static int dgrad_multi(const float* const* dz, const uint8_t* const* dz_idx, const float* const* u_packed,
                       const float* const* act, const float* const* addend, float* const* out, float* const* raw_out,
                       const int* n, int njobs, int hw, int cin, int cout, bool bf, void* stream) {
  UGN_REQUIRE(dz && u_packed && out && n, "ugn_conv3x3_dgrad_wino_multi: null array");
  UGN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, "ugn_conv3x3_dgrad_wino_multi: njobs must be 1..%d (got %d)", kMaxJobs, njobs);
  WinoJob jobs[kMaxJobs];
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(dz[j] && u_packed[j] && out[j] && n[j] > 0, "ugn_conv3x3_dgrad_wino_multi: null pointer or n <= 0 in job %d", j);
    jobs[j] = {dz[j], dz_idx ? dz_idx[j] : nullptr, u_packed[j], out[j], nullptr, act ? act[j] : nullptr,
               addend ? addend[j] : nullptr, raw_out ? raw_out[j] : nullptr};
    UGN_REQUIRE((jobs[0].in_idx != nullptr) == (jobs[j].in_idx != nullptr), "ugn_conv3x3_dgrad_wino_multi: dz_idx for all jobs or none");
  }
  return dispatch_dgrad(jobs, n, njobs, hw, cin, cout, jobs[0].in_idx != nullptr, bf, (hipStream_t)stream);
}
static int dgrad_pair(const float* const* dz, const uint8_t* const* dz_idx, const float* const* u_packed,
                      const float* const* act, const float* const* addend, float* const* out, float* const* raw_out,
                      const int* n, int hw, int cin, int cout, bool bf, void* stream) {
  return dgrad_multi(dz, dz_idx, u_packed, act, addend, out, raw_out, n, 2, hw, cin, cout, bf, stream);
}
extern "C" int ugn_conv3x3_dgrad_wino_multi(const float* const* dz, const uint8_t* const* dz_idx, const float* const* u_packed,
                                            const float* const* act, const float* const* addend, float* const* out,
                                            float* const* raw_out, const int* n, int njobs, int hw, int cin, int cout, int bf16,
                                            void* stream) {
  return dgrad_multi(dz, dz_idx, u_packed, act, addend, out, raw_out, n, njobs, hw, cin, cout, bf16 != 0, stream);
}
extern "C" int ugn_conv3x3_dgrad_wino_pair(const float* const* dz, const uint8_t* const* dz_idx, const float* const* u_packed,
                                           const float* const* act, const float* const* addend, float* const* out,
                                           float* const* raw_out, const int* n, int hw, int cin, int cout, void* stream) {
  return dgrad_pair(dz, dz_idx, u_packed, act, addend, out, raw_out, n, hw, cin, cout, false, stream);
}
