// Everything after the conv stacks: per-bin FC (MatMul layer), modality gate + fMerge, batch-axis L2 normalisation,
// classification head with softmax cross-entropy, batch-all triplet loss, Adam.  Reference call sites:
//   nets/mj_uwyhNets_ba.py:23-54 (MatMul, gate), :814-818 / :1189-1192 (fusion, signature), :847-851 / :1211-1214 (head),
//   nets/triplet_loss_all.py:8-77, mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:169-178 (sign_max), :227 (Adam).
// These tensors are tiny ([62,B,256]); the kernels are latency/HBM-bound and kept simple: coalesced global access,
// operands shared through LDS, one workgroup per bin.
#include "common.h"

namespace {

constexpr int NBINS = 62, FEAT = 128, HID = 256;

// ------------------------------------------------------------------------------------------------------
// per-bin FC
// ------------------------------------------------------------------------------------------------------
constexpr int FC_BT = 24;   // samples per pass (the reference batch per GPU)

// out[k][b][o] = sum_i feat[k][b][i] * W[k][i][o].  Workgroup (bin k, output quarter oq): 256 threads = 64 outputs x 4
// input quarters, so a thread streams 32 weights instead of 128 and 248 workgroups instead of 62 share the 8 MB of weights;
// the four partial sums are combined in a fixed order through LDS (bitwise reproducible).
// (blockIdx.z = job: the per-bin FCs of up to kFcJobs modality branches in one launch)
constexpr int kFcJobs = 4;
struct FcJobs {
  const float* feat[kFcJobs];
  const float* w[kFcJobs];
  const float* dout[kFcJobs];
  float* out[kFcJobs];       // fwd: out; bwd: dW
  float* dfeat[kFcJobs];
  int b[kFcJobs];
};
__global__ __launch_bounds__(256) void binfc_fwd_kernel(const FcJobs jt) {
  __shared__ float sF[FC_BT][FEAT];          // 12 KB
  __shared__ float sR[3][FC_BT][64];         // partials of input quarters 1..3
  const float* __restrict__ feat = jt.feat[blockIdx.z];
  const float* __restrict__ w = jt.w[blockIdx.z];
  float* __restrict__ out = jt.out[blockIdx.z];
  const int bsz = jt.b[blockIdx.z];
  const int k = blockIdx.x, oq = blockIdx.y, ol = threadIdx.x & 63, iq = threadIdx.x >> 6;
  const float* wk = w + ((size_t)k * FEAT + iq * 32) * HID + oq * 64 + ol;
  for (int b0 = 0; b0 < bsz; b0 += FC_BT) {
    __syncthreads();
    for (int e = threadIdx.x; e < FC_BT * FEAT; e += 256) {
      const int bb = e / FEAT, i = e % FEAT;
      sF[bb][i] = b0 + bb < bsz ? feat[((size_t)k * bsz + b0 + bb) * FEAT + i] : 0.f;
    }
    __syncthreads();
    float acc[FC_BT];
#pragma unroll
    for (int bb = 0; bb < FC_BT; ++bb) acc[bb] = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const float wv = wk[(size_t)i * HID];
#pragma unroll
      for (int bb = 0; bb < FC_BT; ++bb) acc[bb] = fmaf(sF[bb][iq * 32 + i], wv, acc[bb]);
    }
    if (iq > 0) {
#pragma unroll
      for (int bb = 0; bb < FC_BT; ++bb) sR[iq - 1][bb][ol] = acc[bb];
    }
    __syncthreads();
    if (iq == 0) {
#pragma unroll
      for (int bb = 0; bb < FC_BT; ++bb)
        if (b0 + bb < bsz)
          out[((size_t)k * bsz + b0 + bb) * HID + oq * 64 + ol] = ((acc[bb] + sR[0][bb][ol]) + sR[1][bb][ol]) + sR[2][bb][ol];
    }
  }
}

constexpr int FCB_BT = 24;   // samples per pass (one pass for the reference batch: dW is written once, not read-modified)
constexpr int FCB_IQ = 32;   // input features per workgroup (grid.y = 128 / FCB_IQ)

// dW[k][i][o] = sum_b feat[k][b][i] * dout[k][b][o];  dfeat[k][b][i] = sum_o dout[k][b][o] * W[k][i][o].
// Workgroup (k, iq) owns input features i in [32*iq, 32*iq + 32): 248 workgroups instead of 62.
// PARTS: 1 = dW only, 2 = dfeat only, 3 = both.  dfeat is on the critical path of the backward pass (HPP backward waits for it), dW is
// not: the engine launches the two halves on different streams (ugn_binfc_bwd_parts_multi).
template <int PARTS>
__global__ __launch_bounds__(256) void binfc_bwd_kernel(const FcJobs jt) {
  const float* __restrict__ feat = jt.feat[blockIdx.z];
  const float* __restrict__ w = jt.w[blockIdx.z];
  const float* __restrict__ dout = jt.dout[blockIdx.z];
  float* __restrict__ dw = jt.out[blockIdx.z];
  float* __restrict__ dfeat = jt.dfeat[blockIdx.z];
  const int bsz = jt.b[blockIdx.z];
  // (round 6: the same sums in the same order, read with 16-byte LDS instructions -- the feature slice transposed to [i][sample] so that
  //  a thread's 24 samples of an input feature are six ds_read_b128 instead of 24 ds_read_b32, the gradient rows four outputs at a time:
  //  2.5x fewer LDS instructions in a kernel that was bound by issuing them)
  __shared__ __attribute__((aligned(16))) float sFt[FCB_IQ][FCB_BT];      // 3 KB, [input feature][sample]
  __shared__ __attribute__((aligned(16))) float sD[FCB_BT][HID];          // 24 KB
  __shared__ float sW[FCB_IQ][HID + 1];                                   // 32.9 KB: this workgroup's slice of W[k]
  static_assert(FCB_BT % 4 == 0 && HID % 4 == 0, "16-byte LDS reads");
  const int k = blockIdx.x, i0 = blockIdx.y * FCB_IQ, tid = threadIdx.x;
  const float* wk = w + ((size_t)k * FEAT + i0) * HID;
  float* dwk = dw + ((size_t)k * FEAT + i0) * HID;
  if (PARTS & 2)
    for (int e = tid; e < FCB_IQ * HID; e += 256) sW[e / HID][e % HID] = wk[e];
  for (int b0 = 0; b0 < bsz; b0 += FCB_BT) {
    const int nb = min(FCB_BT, bsz - b0);
    __syncthreads();
    if (PARTS & 1)
      for (int e = tid; e < FCB_BT * FCB_IQ; e += 256) {
        const int bb = e / FCB_IQ, i = e % FCB_IQ;
        sFt[i][bb] = bb < nb ? feat[((size_t)k * bsz + b0 + bb) * FEAT + i0 + i] : 0.f;
      }
    for (int e = tid; e < FCB_BT * HID; e += 256) {
      const int bb = e / HID, o = e % HID;
      sD[bb][o] = bb < nb ? dout[((size_t)k * bsz + b0 + bb) * HID + o] : 0.f;
    }
    __syncthreads();
    if (PARTS & 1) {   // dW rows of this slice; thread = output feature o
      float dreg[FCB_BT];
#pragma unroll
      for (int bb = 0; bb < FCB_BT; ++bb) dreg[bb] = sD[bb][tid];
#pragma unroll 4
      for (int i = 0; i < FCB_IQ; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int b4 = 0; b4 < FCB_BT / 4; ++b4) {
          const float4 f = *reinterpret_cast<const float4*>(&sFt[i][4 * b4]);
          acc = fmaf(f.x, dreg[4 * b4], acc);
          acc = fmaf(f.y, dreg[4 * b4 + 1], acc);
          acc = fmaf(f.z, dreg[4 * b4 + 2], acc);
          acc = fmaf(f.w, dreg[4 * b4 + 3], acc);
        }
        if (b0 == 0) dwk[(size_t)i * HID + tid] = acc;
        else dwk[(size_t)i * HID + tid] += acc;
      }
    }
    if (PARTS & 2) {   // dfeat for this slice; thread = (i, sample triple): 32 x 8 threads, 3 samples each
      const int i = tid & 31, bq = tid >> 5;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll 4
      for (int o = 0; o < HID; o += 4) {
        const float4 d0 = *reinterpret_cast<const float4*>(&sD[3 * bq][o]);
        const float4 d1 = *reinterpret_cast<const float4*>(&sD[3 * bq + 1][o]);
        const float4 d2 = *reinterpret_cast<const float4*>(&sD[3 * bq + 2][o]);
        const float w0 = sW[i][o], w1 = sW[i][o + 1], w2 = sW[i][o + 2], w3 = sW[i][o + 3];
        a0 = fmaf(d0.x, w0, a0); a1 = fmaf(d1.x, w0, a1); a2 = fmaf(d2.x, w0, a2);
        a0 = fmaf(d0.y, w1, a0); a1 = fmaf(d1.y, w1, a1); a2 = fmaf(d2.y, w1, a2);
        a0 = fmaf(d0.z, w2, a0); a1 = fmaf(d1.z, w2, a1); a2 = fmaf(d2.z, w2, a2);
        a0 = fmaf(d0.w, w3, a0); a1 = fmaf(d1.w, w3, a1); a2 = fmaf(d2.w, w3, a2);
      }
      if (3 * bq < nb) dfeat[((size_t)k * bsz + b0 + 3 * bq) * FEAT + i0 + i] = a0;
      if (3 * bq + 1 < nb) dfeat[((size_t)k * bsz + b0 + 3 * bq + 1) * FEAT + i0 + i] = a1;
      if (3 * bq + 2 < nb) dfeat[((size_t)k * bsz + b0 + 3 * bq + 2) * FEAT + i0 + i] = a2;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// gate + fMerge
// ------------------------------------------------------------------------------------------------------
struct ModPtrs {
  const float* x[4];
  const float* use[4];
  float* dx[4];
};

__global__ void gate_fuse_fwd_kernel(ModPtrs mp, int nmod, int mode, float* __restrict__ fused, uint8_t* __restrict__ sel,
                                     int bsz, size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int b = (int)((e / HID) % bsz);
  float g[4];
  for (int m = 0; m < nmod; ++m) g[m] = mp.x[m][e] * mp.use[m][b];
  float out;
  int s = 0;
  if (mode == UGN_FUSE_SIGN_MAX) {
    float best = fabsf(g[0]);
    for (int m = 1; m < nmod; ++m)
      if (fabsf(g[m]) > best) { best = fabsf(g[m]); s = m; }   // first index wins ties (tf.argmax)
    out = g[s];
  } else if (mode == UGN_FUSE_MAX) {
    out = g[0];
    for (int m = 1; m < nmod; ++m)
      if (g[m] > out) { out = g[m]; s = m; }                   // tf.maximum gradient: ties to the first argument
  } else {
    out = 0.f;
    for (int m = 0; m < nmod; ++m) out += g[m];
    out /= (float)nmod;
    s = 255;
  }
  fused[e] = out;
  sel[e] = (uint8_t)s;
}

__global__ void gate_fuse_bwd_kernel(ModPtrs mp, int nmod, int mode, const float* __restrict__ df,
                                     const uint8_t* __restrict__ sel, int bsz, size_t total) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int b = (int)((e / HID) % bsz);
  const float g = df[e];
  const int s = sel[e];
  for (int m = 0; m < nmod; ++m) {
    const float part = mode == UGN_FUSE_AVG ? g / (float)nmod : (s == m ? g : 0.f);
    mp.dx[m][e] = part * mp.use[m][b];
  }
}

// ------------------------------------------------------------------------------------------------------
// batch-axis L2 normalisation: one thread per (bin, feature) column of length B
// ------------------------------------------------------------------------------------------------------
__global__ void l2norm_fwd_kernel(const float* __restrict__ f, float* __restrict__ sig, int bsz) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= NBINS * HID) return;
  const int k = col / HID, d = col % HID;
  const float* src = f + (size_t)k * bsz * HID + d;
  float ss = 0.f;
  for (int b = 0; b < bsz; ++b) { const float v = src[(size_t)b * HID]; ss = fmaf(v, v, ss); }
  const float inv = 1.f / sqrtf(fmaxf(ss, 1e-12f));
  float* dst = sig + (size_t)k * bsz * HID + d;
  for (int b = 0; b < bsz; ++b) dst[(size_t)b * HID] = src[(size_t)b * HID] * inv;
}

__global__ void l2norm_bwd_kernel(const float* __restrict__ f, const float* __restrict__ sig, const float* __restrict__ dsig,
                                  float* __restrict__ df, int bsz) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= NBINS * HID) return;
  const int k = col / HID, d = col % HID;
  const size_t base = (size_t)k * bsz * HID + d;
  float ss = 0.f, dot = 0.f;
  for (int b = 0; b < bsz; ++b) {
    const float v = f[base + (size_t)b * HID];
    ss = fmaf(v, v, ss);
    dot = fmaf(sig[base + (size_t)b * HID], dsig[base + (size_t)b * HID], dot);
  }
  const bool active = ss > 1e-12f;  // below the clamp the norm is a constant
  const float inv = 1.f / sqrtf(fmaxf(ss, 1e-12f));
  for (int b = 0; b < bsz; ++b) {
    const size_t o = base + (size_t)b * HID;
    const float g = dsig[o];
    df[o] = active ? (g - sig[o] * dot) * inv : g * inv;
  }
}

// ------------------------------------------------------------------------------------------------------
// classification head
// ------------------------------------------------------------------------------------------------------
constexpr int HD_BT = 24;                    // samples per pass: one pass for the reference batch, weights read once
constexpr int HD_DQ = 64;                    // features per workgroup
constexpr int HD_PARTS = NBINS * (HID / HD_DQ);   // 248 partial-logit slabs

// part[(k*4+q)][b][c] = sum_{d in quarter q} sig[k][b][d] * wc[(k*256+d)*ncls + c]
__global__ __launch_bounds__(256) void head_partial_kernel(const float* __restrict__ sig, const float* __restrict__ wc,
                                                           float* __restrict__ part, int bsz, int ncls) {
  __shared__ float sS[HD_BT][HD_DQ];
  const int k = blockIdx.x, q = blockIdx.y, c = threadIdx.x;
  const int d0 = q * HD_DQ;
  for (int b0 = 0; b0 < bsz; b0 += HD_BT) {
    __syncthreads();
    for (int e = threadIdx.x; e < HD_BT * HD_DQ; e += 256) {
      const int bb = e / HD_DQ, d = e % HD_DQ;
      sS[bb][d] = b0 + bb < bsz ? sig[((size_t)k * bsz + b0 + bb) * HID + d0 + d] : 0.f;
    }
    __syncthreads();
    if (c < ncls) {
      // (round 6 tried the 16-byte-read form that helps the backward kernels here and in binfc_fwd: 16 -> 40 us and 17 -> 30 us --
      //  these two stream their weights from HBM inside the loop, and the wider LDS reads cost them the overlap of those loads)
      float acc[HD_BT];
#pragma unroll
      for (int bb = 0; bb < HD_BT; ++bb) acc[bb] = 0.f;
#pragma unroll 8
      for (int d = 0; d < HD_DQ; ++d) {
        const float wv = wc[((size_t)k * HID + d0 + d) * ncls + c];
#pragma unroll
        for (int bb = 0; bb < HD_BT; ++bb) acc[bb] = fmaf(sS[bb][d], wv, acc[bb]);
      }
#pragma unroll
      for (int bb = 0; bb < HD_BT; ++bb)
        if (b0 + bb < bsz) part[((size_t)(k * (HID / HD_DQ) + q) * bsz + b0 + bb) * ncls + c] = acc[bb];
    }
  }
}

__device__ __forceinline__ float block_reduce(float v, float* sRed, bool is_max) {
  // 256 threads
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_down(v, off, 64);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sRed[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sRed[i]) : r + sRed[i];
  return r;
}

// one workgroup per sample: logits = bc + sum_k part; softmax; loss; dlogits; top-1 hit
__global__ __launch_bounds__(1024) void head_softmax_kernel(const float* __restrict__ part, const float* __restrict__ bc,
                                                           const float* __restrict__ onehot, float* __restrict__ probs,
                                                           float* __restrict__ row_loss, float* __restrict__ dlogits,
                                                           float* __restrict__ hit, float grad_scale, int bsz, int ncls) {
  __shared__ float sRed[4];
  __shared__ int sArg[2];
  __shared__ float sZ[3][256];
  // 1024 threads: slab lane sl sums slabs sl, sl+4, ...; lanes 1..3 hand their sums to lane 0 (fixed order), which goes on
  const int b = blockIdx.x, c = threadIdx.x & 255, sl = threadIdx.x >> 8;
  float zp = 0.f;
  if (c < ncls)
    for (int k = sl; k < HD_PARTS; k += 4) zp += part[((size_t)k * bsz + b) * ncls + c];
  if (sl > 0) sZ[sl - 1][c] = zp;
  __syncthreads();
  if (sl > 0) return;
  float z = -INFINITY, t = 0.f;
  if (c < ncls) {
    z = bc[c] + (((zp + sZ[0][c]) + sZ[1][c]) + sZ[2][c]);
    t = onehot[(size_t)b * ncls + c];
  }
  const float zmax = block_reduce(z, sRed, true);
  const float ex = c < ncls ? expf(z - zmax) : 0.f;
  const float esum = block_reduce(ex, sRed, false);
  const float lse = logf(esum);
  const float logp = (z - zmax) - lse;
  const float lsum = block_reduce(c < ncls ? -t * logp : 0.f, sRed, false);
  const float tmax = block_reduce(c < ncls ? t : -INFINITY, sRed, true);
  if (c == 0) { sArg[0] = ncls; sArg[1] = ncls; }
  __syncthreads();
  if (c < ncls && z == zmax) atomicMin(&sArg[0], c);   // first maximum, as np/tf argmax
  if (c < ncls && t == tmax) atomicMin(&sArg[1], c);
  __syncthreads();
  if (c < ncls) {
    const float p = ex / esum;
    probs[(size_t)b * ncls + c] = p;
    dlogits[(size_t)b * ncls + c] = (p - t) * grad_scale;
  }
  if (c == 0) {
    row_loss[b] = lsum;
    hit[b] = sArg[0] == sArg[1] ? 1.f : 0.f;
  }
}

constexpr int HB_BT = 24;   // one pass for the reference batch: dwc is written once

// Workgroup (k, q) owns features d in [64q, 64q+64) of bin k: rows (k*256 + d) of wc / dwc and columns d of dsig.
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ sig, const float* __restrict__ wc,
                                                       const float* __restrict__ dlogits, float* __restrict__ dwc,
                                                       float* __restrict__ dbc, float* __restrict__ dsig, int accumulate,
                                                       int bsz, int ncls) {
  __shared__ __attribute__((aligned(16))) float sSt[HD_DQ][HB_BT];    // 6 KB, [feature][sample]
  __shared__ __attribute__((aligned(16))) float sL[HB_BT][HID];      // dlogits rows (ncls <= 256), 24 KB
  extern __shared__ float sWc[];        // [HD_DQ][ncls + 1]: this workgroup's slice of wc, read coalesced once
  const int k = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
  const int d0 = q * HD_DQ;
  const int ldw = ncls + 1;
  for (int e = tid; e < HD_DQ * ncls; e += 256) sWc[(e / ncls) * ldw + e % ncls] = wc[((size_t)k * HID + d0) * ncls + e];
  float bsum = 0.f;
  for (int b0 = 0; b0 < bsz; b0 += HB_BT) {
    const int nb = min(HB_BT, bsz - b0);
    __syncthreads();
    for (int e = tid; e < HB_BT * HD_DQ; e += 256) {
      const int bb = e / HD_DQ, d = e % HD_DQ;
      sSt[d][bb] = bb < nb ? sig[((size_t)k * bsz + b0 + bb) * HID + d0 + d] : 0.f;
    }
    for (int e = tid; e < HB_BT * HID; e += 256) {
      const int bb = e / HID, c = e % HID;
      sL[bb][c] = (bb < nb && c < ncls) ? dlogits[(size_t)(b0 + bb) * ncls + c] : 0.f;
    }
    __syncthreads();
    if (tid < ncls) {   // dwc rows of this slice; thread = class c
      float lreg[HB_BT];
#pragma unroll
      for (int bb = 0; bb < HB_BT; ++bb) { lreg[bb] = sL[bb][tid]; bsum += lreg[bb]; }
#pragma unroll 4
      for (int d = 0; d < HD_DQ; ++d) {
        float acc = 0.f;
#pragma unroll
        for (int b4 = 0; b4 < HB_BT / 4; ++b4) {
          const float4 sv = *reinterpret_cast<const float4*>(&sSt[d][4 * b4]);
          acc = fmaf(sv.x, lreg[4 * b4], acc);
          acc = fmaf(sv.y, lreg[4 * b4 + 1], acc);
          acc = fmaf(sv.z, lreg[4 * b4 + 2], acc);
          acc = fmaf(sv.w, lreg[4 * b4 + 3], acc);
        }
        const size_t o = ((size_t)k * HID + d0 + d) * ncls + tid;
        if (b0 == 0) dwc[o] = acc; else dwc[o] += acc;
      }
    }
    {   // dsig[k][b][d] (+)= sum_c dlogits[b][c] * wc[k*256+d][c]; thread = (d, sample sextet): 64 x 4 threads
      const int d = tid & 63, bq = tid >> 6;
      float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* wrow = sWc + d * ldw;
      int c = 0;
      for (; c + 4 <= ncls; c += 4) {          // (the same sums in the same order: the dlogits rows read 16 bytes at a time)
        const float w0 = wrow[c], w1 = wrow[c + 1], w2 = wrow[c + 2], w3 = wrow[c + 3];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float4 lv = *reinterpret_cast<const float4*>(&sL[6 * bq + j][c]);
          acc[j] = fmaf(lv.x, w0, acc[j]);
          acc[j] = fmaf(lv.y, w1, acc[j]);
          acc[j] = fmaf(lv.z, w2, acc[j]);
          acc[j] = fmaf(lv.w, w3, acc[j]);
        }
      }
      for (; c < ncls; ++c) {
        const float wv = wrow[c];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = fmaf(sL[6 * bq + j][c], wv, acc[j]);
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (6 * bq + j < nb) {
          const size_t o = ((size_t)k * bsz + b0 + 6 * bq + j) * HID + d0 + d;
          dsig[o] = accumulate ? dsig[o] + acc[j] : acc[j];
        }
    }
  }
  if (k == 0 && q == 0 && tid < ncls) dbc[tid] = bsum;
}

// ------------------------------------------------------------------------------------------------------
// gate + fMerge + batch-axis normalisation in ONE launch (and their gradients in one): the per-replica batch (<= GF_MAXB clips) of a
// (bin, feature) column stays in registers between the two stages.  Same arithmetic, statement for statement, as
// gate_fuse_fwd_kernel + l2norm_fwd_kernel and l2norm_bwd_kernel + gate_fuse_bwd_kernel (bit-identical results); two launches and a
// round trip of `fused` / `df` through HBM less on the turn-around of the step, where nothing else runs.
// ------------------------------------------------------------------------------------------------------
constexpr int GF_MAXB = 32;
constexpr int GF_COLS = 32, GF_BG = 8;        // a workgroup: 32 consecutive features of a bin x 8 clip groups (clips bg, bg + 8, ...)

// gate + fMerge of one element (NMOD a template parameter: no indexed private arrays); s = the selected modality (255: average)
template <int NMOD>
__device__ __forceinline__ float gate_one(const ModPtrs& mp, int mode, size_t e, int b, int& s) {
  float g[NMOD];
#pragma unroll
  for (int m = 0; m < NMOD; ++m) g[m] = mp.x[m][e] * mp.use[m][b];
  float out = g[0];
  s = 0;
  if (mode == UGN_FUSE_SIGN_MAX) {
    float best = fabsf(g[0]);
#pragma unroll
    for (int m = 1; m < NMOD; ++m)
      if (fabsf(g[m]) > best) { best = fabsf(g[m]); s = m; out = g[m]; }      // first index wins ties (tf.argmax)
  } else if (mode == UGN_FUSE_MAX) {
#pragma unroll
    for (int m = 1; m < NMOD; ++m)
      if (g[m] > out) { out = g[m]; s = m; }                                    // tf.maximum gradient: ties to the first argument
  } else {
    out = 0.f;
#pragma unroll
    for (int m = 0; m < NMOD; ++m) out += g[m];
    out /= (float)NMOD;
    s = 255;
  }
  return out;
}

// The squares of a column are summed by ONE thread in clip order (fmaf chain), as l2norm_fwd_kernel does: bit-identical.
template <int NMOD>
__global__ __launch_bounds__(GF_COLS * GF_BG) void gate_norm_fwd_kernel(ModPtrs mp, int mode, float* __restrict__ fused,
                                                                        uint8_t* __restrict__ sel, float* __restrict__ sig, int bsz) {
  __shared__ float sV[GF_MAXB][GF_COLS];
  __shared__ float sInv[GF_COLS];
  const int c = threadIdx.x % GF_COLS, bg = threadIdx.x / GF_COLS;
  const int col = blockIdx.x * GF_COLS + c, k = col / HID, d = col % HID;      // (HID is a multiple of GF_COLS: one bin per workgroup)
  const size_t base = (size_t)k * bsz * HID + d;
  float v[GF_MAXB / GF_BG];
#pragma unroll
  for (int i = 0; i < GF_MAXB / GF_BG; ++i) {
    const int b = bg + GF_BG * i;
    if (b < bsz) {
      int s;
      v[i] = gate_one<NMOD>(mp, mode, base + (size_t)b * HID, b, s);
      fused[base + (size_t)b * HID] = v[i];
      sel[base + (size_t)b * HID] = (uint8_t)s;
      sV[b][c] = v[i];
    }
  }
  __syncthreads();
  if (bg == 0) {
    float ss = 0.f;
    for (int b = 0; b < bsz; ++b) ss = fmaf(sV[b][c], sV[b][c], ss);
    sInv[c] = 1.f / sqrtf(fmaxf(ss, 1e-12f));
  }
  __syncthreads();
  const float inv = sInv[c];
#pragma unroll
  for (int i = 0; i < GF_MAXB / GF_BG; ++i) {
    const int b = bg + GF_BG * i;
    if (b < bsz) sig[base + (size_t)b * HID] = v[i] * inv;
  }
}

template <int NMOD>
__global__ __launch_bounds__(GF_COLS * GF_BG) void gate_norm_bwd_kernel(ModPtrs mp, int mode, const float* __restrict__ f,
                                                                        const float* __restrict__ sig, const float* __restrict__ dsig,
                                                                        const uint8_t* __restrict__ sel, int bsz) {
  __shared__ float sF[GF_MAXB][GF_COLS], sS[GF_MAXB][GF_COLS], sG[GF_MAXB][GF_COLS];
  __shared__ float sDot[GF_COLS], sInv[GF_COLS];
  __shared__ int sAct[GF_COLS];
  const int c = threadIdx.x % GF_COLS, bg = threadIdx.x / GF_COLS;
  const int col = blockIdx.x * GF_COLS + c, k = col / HID, d = col % HID;
  const size_t base = (size_t)k * bsz * HID + d;
  float sv[GF_MAXB / GF_BG], gv[GF_MAXB / GF_BG];
#pragma unroll
  for (int i = 0; i < GF_MAXB / GF_BG; ++i) {
    const int b = bg + GF_BG * i;
    if (b < bsz) {
      const size_t o = base + (size_t)b * HID;
      sv[i] = sig[o];
      gv[i] = dsig[o];
      sF[b][c] = f[o];
      sS[b][c] = sv[i];
      sG[b][c] = gv[i];
    }
  }
  __syncthreads();
  if (bg == 0) {       // the two sums of l2norm_bwd_kernel, in its order
    float ss = 0.f, dot = 0.f;
    for (int b = 0; b < bsz; ++b) {
      ss = fmaf(sF[b][c], sF[b][c], ss);
      dot = fmaf(sS[b][c], sG[b][c], dot);
    }
    sAct[c] = ss > 1e-12f ? 1 : 0;
    sInv[c] = 1.f / sqrtf(fmaxf(ss, 1e-12f));
    sDot[c] = dot;
  }
  __syncthreads();
  const bool active = sAct[c] != 0;
  const float inv = sInv[c], dot = sDot[c];
#pragma unroll
  for (int i = 0; i < GF_MAXB / GF_BG; ++i) {
    const int b = bg + GF_BG * i;
    if (b < bsz) {
      const size_t o = base + (size_t)b * HID;
      const float g = active ? (gv[i] - sv[i] * dot) * inv : gv[i] * inv;
      const int s = sel[o];
#pragma unroll
      for (int m = 0; m < NMOD; ++m) {
        const float part = mode == UGN_FUSE_AVG ? g / (float)NMOD : (s == m ? g : 0.f);
        mp.dx[m][o] = part * mp.use[m][b];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
constexpr int TR_DC = 32;  // feature chunk staged in LDS

// The part every triplet variant shares (nets/triplet_loss_all.py:70-77 `batch_dist`): Gram matrix of a bin's m embeddings in
// LDS, then d_ij = sqrt(max(|y_i|^2 + |y_j|^2 - 2 y_i.y_j, 0)), exactly 0 where the argument is <= 0.  Leaves sD = distances,
// sG = 0; sY is scratch ([m][TR_DC+1], its first m floats are reused as a vector afterwards).
// WHOLE (batches of at most TR_WHOLE_M clips): the bin's m x 256 embeddings are staged in LDS ONCE (sY = [m][HID + 1], sN behind it)
// and stay there for the backward half -- instead of eight chunk rounds with two barriers each here and 2 m^2 global reads per
// thread there.  The sums run in the same order (32-feature partial sums added chunk by chunk): bit-identical results.
constexpr int TR_WHOLE_M = 64;
template <bool WHOLE>
__device__ __forceinline__ void bin_distances(const float* __restrict__ Y, float* sD, float* sG, float* sY, float* sN, int m, int tid) {
  const int mm = m * m;
  if constexpr (WHOLE) {
    for (int e = tid; e < m * HID; e += 256) sY[(e / HID) * (HID + 1) + (e % HID)] = Y[e];
    for (int p = tid; p < mm; p += 256) sG[p] = 0.f;
    __syncthreads();
    for (int p = tid; p < mm; p += 256) {
      const int i = p / m, j = p % m;
      if (j < i) continue;  // symmetric: compute the upper triangle, mirror below
      const float* yi = sY + i * (HID + 1);
      const float* yj = sY + j * (HID + 1);
      float tot = 0.f;
      for (int d0 = 0; d0 < HID; d0 += TR_DC) {
        float acc = 0.f;
#pragma unroll
        for (int dd = 0; dd < TR_DC; ++dd) acc = fmaf(yi[d0 + dd], yj[d0 + dd], acc);
        tot += acc;
      }
      sD[p] = tot;
    }
  } else {
    for (int p = tid; p < mm; p += 256) { sD[p] = 0.f; sG[p] = 0.f; }
    for (int d0 = 0; d0 < HID; d0 += TR_DC) {
      __syncthreads();
      for (int e = tid; e < m * TR_DC; e += 256) sY[(e / TR_DC) * (TR_DC + 1) + (e % TR_DC)] = Y[(size_t)(e / TR_DC) * HID + d0 + (e % TR_DC)];
      __syncthreads();
      for (int p = tid; p < mm; p += 256) {
        const int i = p / m, j = p % m;
        if (j < i) continue;  // symmetric: compute the upper triangle, mirror below
        float acc = 0.f;
#pragma unroll
        for (int dd = 0; dd < TR_DC; ++dd) acc = fmaf(sY[i * (TR_DC + 1) + dd], sY[j * (TR_DC + 1) + dd], acc);
        sD[p] += acc;
      }
    }
  }
  __syncthreads();
  for (int p = tid; p < mm; p += 256) {
    const int i = p / m, j = p % m;
    if (j < i) sD[p] = sD[j * m + i];
  }
  __syncthreads();
  // squared norms = diagonal of the Gram matrix (same summation order -> d_ii is exactly 0)
  for (int i = tid; i < m; i += 256) sN[i] = sD[i * m + i];
  __syncthreads();
  for (int p = tid; p < mm; p += 256) {
    const int i = p / m, j = p % m;
    const float q = fmaxf(sN[i] + sN[j] - 2.f * sD[p], 0.f);
    // reference: sqrt(q + [q<=0]*1e-16) * [q>0]  ==  q > 0 ? sqrt(q) : 0
    sD[p] = q > 0.f ? sqrtf(q) : 0.f;
  }
  __syncthreads();
}

// From sG = dL/d(dist) (unscaled) to dL/dY of the bin: dL/d(sqdist) = dL/d(dist) * scale / (2 dist) (0 where dist == 0),
// S = dq + dq^T, dY_i = 2 * (rowsum_i * Y_i - sum_j S_ij Y_j); thread = feature d.
template <bool WHOLE>
__device__ __forceinline__ void bin_backprop(const float* __restrict__ Y, const float* sY, float* __restrict__ dY, float* sD, float* sG,
                                             float* sN, float scale, int m, int tid) {
  const int mm = m * m;
  // dL/d(sqdist) = dL/d(dist) / (2 dist), 0 where dist == 0
  for (int p = tid; p < mm; p += 256) {
    const float d = sD[p];
    sG[p] = d > 0.f ? sG[p] * scale / (2.f * d) : 0.f;
  }
  __syncthreads();
  // S = dq + dq^T into sD; row sums into sN
  for (int p = tid; p < mm; p += 256) {
    const int i = p / m, j = p % m;
    sD[p] = sG[p] + sG[j * m + i];
  }
  __syncthreads();
  for (int i = tid; i < m; i += 256) {
    float s = 0.f;
    for (int j = 0; j < m; ++j) s += sD[i * m + j];
    sN[i] = s;
  }
  __syncthreads();
  // dY_i = 2 * (rows_i * Y_i - sum_j S_ij Y_j); thread = feature d
  for (int i = 0; i < m; ++i) {
    float acc = sN[i] * (WHOLE ? sY[i * (HID + 1) + tid] : Y[(size_t)i * HID + tid]);
    for (int j = 0; j < m; ++j) acc = fmaf(-sD[i * m + j], WHOLE ? sY[j * (HID + 1) + tid] : Y[(size_t)j * HID + tid], acc);
    dY[(size_t)i * HID + tid] = 2.f * acc;
  }
}

// batch-all triplet loss: one workgroup per bin, Gram/distance matrix in LDS (m <= 128)
// ------------------------------------------------------------------------------------------------------
template <bool WHOLE>
__global__ __launch_bounds__(256) void triplet_kernel(const float* __restrict__ sig, const int32_t* __restrict__ hp,
                                                      const int32_t* __restrict__ hn, int kp, int kn, float margin,
                                                      float* __restrict__ bin_loss, float* __restrict__ bin_num,
                                                      float* __restrict__ dsig, float grad_scale, int m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sD = smem;                 // [m*m] Gram -> distance -> S
  float* sG = smem + m * m;         // [m*m] dL/d(dist) -> dL/d(sqdist)
  float* sY = sG + m * m;           // [m][TR_DC+1]
  __shared__ float sRed[4];
  const int k = blockIdx.x, tid = threadIdx.x;
  const float* Y = sig + (size_t)k * m * HID;

  float* sN = WHOLE ? sY + m * (HID + 1) : sY;        // [m] (chunked form: the chunk buffer is free by then)
  bin_distances<WHOLE>(Y, sD, sG, sY, sN, m, tid);
  // pass A: items (row r, positive a): hinge sum, active count, gradient of the positive distance
  float lsum = 0.f, lnum = 0.f;
  for (int it = tid; it < m * kp; it += 256) {
    const int r = it / kp;
    const int pi = hp[it];
    const float dpv = sD[pi] + margin;
    float cnt = 0.f;
    for (int b = 0; b < kn; ++b) {
      const float h = dpv - sD[hn[r * kn + b]];
      if (h > 0.f) { lsum += h; cnt += 1.f; }
    }
    lnum += cnt;
    sG[pi] = cnt;
  }
  // pass B: items (row r, negative b): gradient of the negative distance
  for (int it = tid; it < m * kn; it += 256) {
    const int r = it / kn;
    const int ni = hn[it];
    const float dnv = sD[ni];
    float cnt = 0.f;
    for (int a = 0; a < kp; ++a) {
      const float h = sD[hp[r * kp + a]] + margin - dnv;
      if (h > 0.f) cnt += 1.f;
    }
    sG[ni] = -cnt;
  }
  const float tsum = block_reduce(lsum, sRed, false);
  const float tnum = block_reduce(lnum, sRed, false);
  if (tid == 0) {
    bin_loss[k] = tnum != 0.f ? tsum / tnum : 0.f;
    bin_num[k] = tnum;
  }
  const float scale = tnum != 0.f ? grad_scale / (tnum * (float)NBINS) : 0.f;
  __syncthreads();
  bin_backprop<WHOLE>(Y, sY, dsig + (size_t)k * m * HID, sD, sG, sN, scale, m, tid);
}

// Batch-HARD triplet loss per bin (the mode nets/mj_uwyhNets_ba.py:1301-1306 `compile_hard` names: tfa.losses.TripletHardLoss,
// soft = False, L2 distances), applied to each of the 62 bins of the signature like the batch-all loss and averaged over them:
//   hp_a = max over the OTHER samples of a's identity of d_ap   (tfa _masked_maximum; no such sample: the row minimum = d_aa = 0)
//   hn_a = min over the samples of other identities of d_an     (tfa _masked_minimum; no such sample: the row maximum)
//   L_bin = mean_a max(hp_a - hn_a + margin, 0)
// Gradient: reduce_max / reduce_min split it equally among tied extrema.  bin_num = anchors with a positive hinge.
template <bool WHOLE>
__global__ __launch_bounds__(256) void triplet_hard_kernel(const float* __restrict__ sig, const int32_t* __restrict__ labels,
                                                           float margin, float* __restrict__ bin_loss,
                                                           float* __restrict__ bin_num, float* __restrict__ dsig,
                                                           float grad_scale, int m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sD = smem;
  float* sG = smem + m * m;
  float* sY = sG + m * m;
  __shared__ float sRed[4];
  const int k = blockIdx.x, tid = threadIdx.x;
  const float* Y = sig + (size_t)k * m * HID;
  float* sN = WHOLE ? sY + m * (HID + 1) : sY;        // [m] (chunked form: the chunk buffer is free by then)
  bin_distances<WHOLE>(Y, sD, sG, sY, sN, m, tid);
  float lsum = 0.f, lnum = 0.f;
  for (int a = tid; a < m; a += 256) {
    const int la = labels[a];
    float hp = -INFINITY, hn = INFINITY, rmax = -INFINITY;
    int np = 0, nn = 0;
    for (int j = 0; j < m; ++j) {
      const float d = sD[a * m + j];
      rmax = fmaxf(rmax, d);
      if (labels[j] == la) { if (j != a) { hp = fmaxf(hp, d); ++np; } }
      else { hn = fminf(hn, d); ++nn; }
    }
    if (np == 0) hp = 0.f;         // row minimum: the zeroed diagonal
    if (nn == 0) hn = rmax;
    const float h = hp - hn + margin;
    if (h > 0.f) {
      lsum += h;
      lnum += 1.f;
      float cp = 0.f, cn = 0.f;
      for (int j = 0; j < m; ++j) {
        const float d = sD[a * m + j];
        const bool same = labels[j] == la;
        if (np > 0 && same && j != a && d == hp) cp += 1.f;
        if (nn > 0 ? (!same && d == hn) : (d == hn)) cn += 1.f;
      }
      for (int j = 0; j < m; ++j) {
        const float d = sD[a * m + j];
        const bool same = labels[j] == la;
        float g = 0.f;
        if (np > 0 && same && j != a && d == hp) g += 1.f / cp;
        if (nn > 0 ? (!same && d == hn) : (d == hn)) g -= 1.f / cn;
        sG[a * m + j] = g;
      }
    }
  }
  const float tsum = block_reduce(lsum, sRed, false);
  const float tnum = block_reduce(lnum, sRed, false);
  if (tid == 0) {
    bin_loss[k] = tsum / (float)m;
    bin_num[k] = tnum;
  }
  __syncthreads();
  bin_backprop<WHOLE>(Y, sY, dsig + (size_t)k * m * HID, sD, sG, sN, grad_scale / ((float)m * (float)NBINS), m, tid);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, float lr_t, float b1, float b2, float eps, float gscale) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float gv = g[e] * gscale;
  const float mv = b1 * m[e] + (1.f - b1) * gv;
  const float vv = b2 * v[e] + (1.f - b2) * gv * gv;
  m[e] = mv;
  v[e] = vv;
  p[e] = p[e] - lr_t * mv / (sqrtf(vv) + eps);
}

// the same update with the step-dependent learning rate read from device memory: the launch can then live in a captured
// hipGraph that is replayed every step while the host only rewrites that one float
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                size_t n, const float* __restrict__ lr_t_dev, float b1, float b2, float eps, float gscale) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float lr_t = *lr_t_dev;
  const float gv = g[e] * gscale;
  const float mv = b1 * m[e] + (1.f - b1) * gv;
  const float vv = b2 * v[e] + (1.f - b2) * gv * gv;
  m[e] = mv;
  v[e] = vv;
  p[e] = p[e] - lr_t * mv / (sqrtf(vv) + eps);
}

}  // namespace

extern "C" int ugn_binfc_fwd_multi(const float* const* feat, const float* const* w, float* const* out, const int* b, int njobs,
                                   void* stream) {
  UGN_REQUIRE(feat && w && out && b && njobs >= 1 && njobs <= kFcJobs, "ugn_binfc_fwd_multi: bad arguments (1..%d jobs)", kFcJobs);
  FcJobs jt = {};
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(feat[j] && w[j] && out[j] && b[j] > 0, "ugn_binfc_fwd_multi: bad job %d", j);
    jt.feat[j] = feat[j]; jt.w[j] = w[j]; jt.out[j] = out[j]; jt.b[j] = b[j];
  }
  hipLaunchKernelGGL(binfc_fwd_kernel, dim3(NBINS, HID / 64, njobs), dim3(256), 0, (hipStream_t)stream, jt);
  UGN_CHECK_LAUNCH("binfc_fwd");
  return 0;
}

extern "C" int ugn_binfc_fwd(const float* feat, const float* w, float* out, int b, void* stream) {
  return ugn_binfc_fwd_multi(&feat, &w, &out, &b, 1, stream);
}

extern "C" int ugn_binfc_bwd_parts_multi(const float* const* feat, const float* const* w, const float* const* dout, float* const* dw,
                                         float* const* dfeat, const int* b, int njobs, int parts, void* stream);
extern "C" int ugn_binfc_bwd_multi(const float* const* feat, const float* const* w, const float* const* dout, float* const* dw,
                                   float* const* dfeat, const int* b, int njobs, void* stream) {
  return ugn_binfc_bwd_parts_multi(feat, w, dout, dw, dfeat, b, njobs, 3, stream);
}

extern "C" int ugn_binfc_bwd_parts_multi(const float* const* feat, const float* const* w, const float* const* dout, float* const* dw,
                                         float* const* dfeat, const int* b, int njobs, int parts, void* stream) {
  UGN_REQUIRE(parts >= 1 && parts <= 3, "ugn_binfc_bwd_parts_multi: parts must be 1 (dW), 2 (dfeat) or 3 (got %d)", parts);
  UGN_REQUIRE(feat && w && dout && dw && dfeat && b && njobs >= 1 && njobs <= kFcJobs,
              "ugn_binfc_bwd_multi: bad arguments (1..%d jobs)", kFcJobs);
  FcJobs jt = {};
  for (int j = 0; j < njobs; ++j) {
    UGN_REQUIRE(feat[j] && w[j] && dout[j] && dw[j] && dfeat[j] && b[j] > 0, "ugn_binfc_bwd_multi: bad job %d", j);
    jt.feat[j] = feat[j]; jt.w[j] = w[j]; jt.dout[j] = dout[j]; jt.out[j] = dw[j]; jt.dfeat[j] = dfeat[j]; jt.b[j] = b[j];
  }
  const dim3 grid(NBINS, FEAT / FCB_IQ, njobs);
  if (parts == 1) hipLaunchKernelGGL(binfc_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, jt);
  else if (parts == 2) hipLaunchKernelGGL(binfc_bwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, jt);
  else hipLaunchKernelGGL(binfc_bwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, jt);
  UGN_CHECK_LAUNCH("binfc_bwd");
  return 0;
}

extern "C" int ugn_binfc_bwd(const float* feat, const float* w, const float* dout, float* dw, float* dfeat, int b,
                             void* stream) {
  return ugn_binfc_bwd_multi(&feat, &w, &dout, &dw, &dfeat, &b, 1, stream);
}

extern "C" int ugn_gate_fuse_fwd(const float* const* outs_host, const float* const* uses_host, int nmod, int mode,
                                 float* fused, uint8_t* sel, int b, void* stream) {
  UGN_REQUIRE(outs_host && uses_host && fused && sel && b > 0, "ugn_gate_fuse_fwd: bad arguments");
  UGN_REQUIRE(nmod >= 1 && nmod <= 4, "ugn_gate_fuse_fwd: nmod must be 1..4 (got %d)", nmod);
  UGN_REQUIRE(mode >= 0 && mode <= 2, "ugn_gate_fuse_fwd: unknown mode %d", mode);
  ModPtrs mp = {};
  for (int m = 0; m < nmod; ++m) { mp.x[m] = outs_host[m]; mp.use[m] = uses_host[m]; }
  const size_t total = (size_t)NBINS * b * HID;
  hipLaunchKernelGGL(gate_fuse_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mp,
                     nmod, mode, fused, sel, b, total);
  UGN_CHECK_LAUNCH("gate_fuse_fwd");
  return 0;
}

extern "C" int ugn_gate_fuse_bwd(const float* dfused, const uint8_t* sel, const float* const* uses_host,
                                 float* const* douts_host, int nmod, int mode, int b, void* stream) {
  UGN_REQUIRE(dfused && sel && uses_host && douts_host && b > 0, "ugn_gate_fuse_bwd: bad arguments");
  UGN_REQUIRE(nmod >= 1 && nmod <= 4, "ugn_gate_fuse_bwd: nmod must be 1..4 (got %d)", nmod);
  ModPtrs mp = {};
  for (int m = 0; m < nmod; ++m) { mp.dx[m] = douts_host[m]; mp.use[m] = uses_host[m]; }
  const size_t total = (size_t)NBINS * b * HID;
  hipLaunchKernelGGL(gate_fuse_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mp,
                     nmod, mode, dfused, sel, b, total);
  UGN_CHECK_LAUNCH("gate_fuse_bwd");
  return 0;
}

/* ugn_gate_fuse_fwd + ugn_l2norm_batch_fwd in one launch (b <= 32): fused, sel and sig as the two calls write them */
extern "C" int ugn_gate_norm_fwd(const float* const* outs_host, const float* const* uses_host, int nmod, int mode, float* fused,
                                 uint8_t* sel, float* sig, int b, void* stream) {
  UGN_REQUIRE(outs_host && uses_host && fused && sel && sig && b > 0 && b <= GF_MAXB, "ugn_gate_norm_fwd: bad arguments (b = 1..%d)", GF_MAXB);
  UGN_REQUIRE(nmod >= 1 && nmod <= 4 && mode >= 0 && mode <= 2, "ugn_gate_norm_fwd: nmod 1..4, mode 0..2");
  ModPtrs mp = {};
  for (int m = 0; m < nmod; ++m) { mp.x[m] = outs_host[m]; mp.use[m] = uses_host[m]; }
  const dim3 grid(NBINS * HID / GF_COLS), block(GF_COLS * GF_BG);
  hipStream_t st = (hipStream_t)stream;
  switch (nmod) {
    case 1: hipLaunchKernelGGL(gate_norm_fwd_kernel<1>, grid, block, 0, st, mp, mode, fused, sel, sig, b); break;
    case 2: hipLaunchKernelGGL(gate_norm_fwd_kernel<2>, grid, block, 0, st, mp, mode, fused, sel, sig, b); break;
    case 3: hipLaunchKernelGGL(gate_norm_fwd_kernel<3>, grid, block, 0, st, mp, mode, fused, sel, sig, b); break;
    default: hipLaunchKernelGGL(gate_norm_fwd_kernel<4>, grid, block, 0, st, mp, mode, fused, sel, sig, b); break;
  }
  UGN_CHECK_LAUNCH("gate_norm_fwd");
  return 0;
}

/* ugn_l2norm_batch_bwd + ugn_gate_fuse_bwd in one launch (b <= 32): douts as the two calls write them */
extern "C" int ugn_gate_norm_bwd(const float* f, const float* sig, const float* dsig, const uint8_t* sel,
                                 const float* const* uses_host, float* const* douts_host, int nmod, int mode, int b, void* stream) {
  UGN_REQUIRE(f && sig && dsig && sel && uses_host && douts_host && b > 0 && b <= GF_MAXB, "ugn_gate_norm_bwd: bad arguments (b = 1..%d)", GF_MAXB);
  UGN_REQUIRE(nmod >= 1 && nmod <= 4 && mode >= 0 && mode <= 2, "ugn_gate_norm_bwd: nmod 1..4, mode 0..2");
  ModPtrs mp = {};
  for (int m = 0; m < nmod; ++m) { mp.dx[m] = douts_host[m]; mp.use[m] = uses_host[m]; }
  const dim3 grid(NBINS * HID / GF_COLS), block(GF_COLS * GF_BG);
  hipStream_t st = (hipStream_t)stream;
  switch (nmod) {
    case 1: hipLaunchKernelGGL(gate_norm_bwd_kernel<1>, grid, block, 0, st, mp, mode, f, sig, dsig, sel, b); break;
    case 2: hipLaunchKernelGGL(gate_norm_bwd_kernel<2>, grid, block, 0, st, mp, mode, f, sig, dsig, sel, b); break;
    case 3: hipLaunchKernelGGL(gate_norm_bwd_kernel<3>, grid, block, 0, st, mp, mode, f, sig, dsig, sel, b); break;
    default: hipLaunchKernelGGL(gate_norm_bwd_kernel<4>, grid, block, 0, st, mp, mode, f, sig, dsig, sel, b); break;
  }
  UGN_CHECK_LAUNCH("gate_norm_bwd");
  return 0;
}

extern "C" int ugn_l2norm_batch_fwd(const float* f, float* sig, int b, void* stream) {
  UGN_REQUIRE(f && sig && b > 0, "ugn_l2norm_batch_fwd: bad arguments");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(NBINS * HID / 64), dim3(64), 0, (hipStream_t)stream, f, sig, b);
  UGN_CHECK_LAUNCH("l2norm_fwd");
  return 0;
}

extern "C" int ugn_l2norm_batch_bwd(const float* f, const float* sig, const float* dsig, float* df, int b, void* stream) {
  UGN_REQUIRE(f && sig && dsig && df && b > 0, "ugn_l2norm_batch_bwd: bad arguments");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(NBINS * HID / 64), dim3(64), 0, (hipStream_t)stream, f, sig, dsig, df, b);
  UGN_CHECK_LAUNCH("l2norm_bwd");
  return 0;
}

extern "C" int ugn_head_fwd(const float* sig, const float* wc, const float* bc, const float* onehot, float* part,
                            float* probs, float* row_loss, float* dlogits, float* hit, float grad_scale, int b, int ncls,
                            void* stream) {
  UGN_REQUIRE(sig && wc && bc && onehot && part && probs && row_loss && dlogits && hit && b > 0,
              "ugn_head_fwd: bad arguments");
  UGN_REQUIRE(ncls >= 1 && ncls <= 256, "ugn_head_fwd: ncls must be 1..256 (got %d)", ncls);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(head_partial_kernel, dim3(NBINS, HID / HD_DQ), dim3(256), 0, st, sig, wc, part, b, ncls);
  hipLaunchKernelGGL(head_softmax_kernel, dim3(b), dim3(1024), 0, st, part, bc, onehot, probs, row_loss, dlogits, hit,
                     grad_scale, b, ncls);
  UGN_CHECK_LAUNCH("head_fwd");
  return 0;
}

extern "C" int ugn_head_bwd(const float* sig, const float* wc, const float* dlogits, float* dwc, float* dbc, float* dsig,
                            int accumulate, int b, int ncls, void* stream) {
  UGN_REQUIRE(sig && wc && dlogits && dwc && dbc && dsig && b > 0, "ugn_head_bwd: bad arguments");
  UGN_REQUIRE(ncls >= 1 && ncls <= 256, "ugn_head_bwd: ncls must be 1..256 (got %d)", ncls);
  static bool attr_done = false;
  if (!attr_done) {   // static 30 KB + up to 64 x 257 floats of dynamic LDS
    hipError_t e = hipFuncSetAttribute((const void*)head_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HD_DQ * 257 * 4);
    UGN_REQUIRE(e == hipSuccess, "ugn_head_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL(head_bwd_kernel, dim3(NBINS, HID / HD_DQ), dim3(256), (size_t)HD_DQ * (ncls + 1) * sizeof(float),
                     (hipStream_t)stream, sig, wc, dlogits, dwc, dbc, dsig,
                     accumulate, b, ncls);
  UGN_CHECK_LAUNCH("head_bwd");
  return 0;
}

extern "C" int ugn_triplet_indices_host(const int32_t* labels, int m, int32_t* hp, int32_t* hn, int* kp, int* kn) {
  UGN_REQUIRE(labels && hp && hn && kp && kn && m > 0, "ugn_triplet_indices_host: bad arguments");
  int np = 0, nn = 0;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      if (labels[i] == labels[j]) hp[np++] = i * m + j;
      else hn[nn++] = i * m + j;
    }
  UGN_REQUIRE(np % m == 0 && nn % m == 0,
              "triplet_loss: reshape([n,m,-1,1]) needs pair counts divisible by the batch size (m=%d, positives=%d, "
              "negatives=%d)", m, np, nn);
  *kp = np / m;
  *kn = nn / m;
  return 0;
}

extern "C" int ugn_triplet_fwd_bwd(const float* sig, const int32_t* hp, const int32_t* hn, int kp, int kn, float margin,
                                   float* bin_loss, float* bin_num, float* dsig, float grad_scale, int m, void* stream) {
  UGN_REQUIRE(sig && bin_loss && bin_num && dsig, "ugn_triplet_fwd_bwd: null pointer");
  UGN_REQUIRE(m >= 1 && m <= 128, "ugn_triplet_fwd_bwd: batch size must be 1..128 (got %d)", m);
  UGN_REQUIRE(kp >= 0 && kn >= 0 && kp + kn == m, "ugn_triplet_fwd_bwd: kp + kn must equal m (kp=%d kn=%d m=%d)", kp, kn, m);
  if (kp == 0 || kn == 0) {   // a batch without negatives (or positives) has no triplet: loss 0, count 0, zero gradient
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(bin_loss, 0, NBINS * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(bin_num, 0, NBINS * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(dsig, 0, (size_t)NBINS * m * HID * sizeof(float), st);
    if (e != hipSuccess) { ugn_set_error("triplet: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
  }
  UGN_REQUIRE(hp && hn, "ugn_triplet_fwd_bwd: null index list");
  const bool whole = m <= TR_WHOLE_M;
  const int lds = (2 * m * m + (whole ? m * (HID + 1) + m : m * (TR_DC + 1))) * (int)sizeof(float);
  static int lds_set[2] = {0, 0};
  if (lds > lds_set[whole]) {
    hipError_t e = hipFuncSetAttribute(whole ? (const void*)triplet_kernel<true> : (const void*)triplet_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { ugn_set_error("triplet: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    lds_set[whole] = lds;
  }
  if (whole)
    hipLaunchKernelGGL(triplet_kernel<true>, dim3(NBINS), dim3(256), lds, (hipStream_t)stream, sig, hp, hn, kp, kn, margin,
                       bin_loss, bin_num, dsig, grad_scale, m);
  else
    hipLaunchKernelGGL(triplet_kernel<false>, dim3(NBINS), dim3(256), lds, (hipStream_t)stream, sig, hp, hn, kp, kn, margin,
                       bin_loss, bin_num, dsig, grad_scale, m);
  UGN_CHECK_LAUNCH("triplet");
  return 0;
}

extern "C" int ugn_triplet_hard_fwd_bwd(const float* sig, const int32_t* labels, float margin, float* bin_loss, float* bin_num,
                                        float* dsig, float grad_scale, int m, void* stream) {
  UGN_REQUIRE(sig && labels && bin_loss && bin_num && dsig, "ugn_triplet_hard_fwd_bwd: null pointer");
  UGN_REQUIRE(m >= 1 && m <= 128, "ugn_triplet_hard_fwd_bwd: batch size must be 1..128 (got %d)", m);
  const bool whole = m <= TR_WHOLE_M;
  const int lds = (2 * m * m + (whole ? m * (HID + 1) + m : m * (TR_DC + 1))) * (int)sizeof(float);
  static int lds_set[2] = {0, 0};
  if (lds > lds_set[whole]) {
    hipError_t e = hipFuncSetAttribute(whole ? (const void*)triplet_hard_kernel<true> : (const void*)triplet_hard_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { ugn_set_error("triplet hard: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    lds_set[whole] = lds;
  }
  if (whole)
    hipLaunchKernelGGL(triplet_hard_kernel<true>, dim3(NBINS), dim3(256), lds, (hipStream_t)stream, sig, labels, margin, bin_loss,
                       bin_num, dsig, grad_scale, m);
  else
    hipLaunchKernelGGL(triplet_hard_kernel<false>, dim3(NBINS), dim3(256), lds, (hipStream_t)stream, sig, labels, margin, bin_loss,
                       bin_num, dsig, grad_scale, m);
  UGN_CHECK_LAUNCH("triplet hard");
  return 0;
}

extern "C" int ugn_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                             float eps, float grad_scale, void* stream) {
  UGN_REQUIRE(p && g && m && v, "ugn_adam_step: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t,
                     b1, b2, eps, grad_scale);
  UGN_CHECK_LAUNCH("adam");
  return 0;
}

extern "C" int ugn_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* lr_t_dev, float b1,
                                 float b2, float eps, float grad_scale, void* stream) {
  UGN_REQUIRE(p && g && m && v && lr_t_dev, "ugn_adam_step_dev: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                     lr_t_dev, b1, b2, eps, grad_scale);
  UGN_CHECK_LAUNCH("adam (device lr)");
  return 0;
}
