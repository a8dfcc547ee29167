// k-nearest-neighbour classification of gait signatures on the GPU: replaces
// sklearn.neighbors.KNeighborsClassifier(n_neighbors=knn).fit(gallery, labels).predict(probes) of the reference's
// evaluation main (mains/mj_testUWYHGaitNet_open_tum.py:328-341): Euclidean metric, uniform weights, majority vote among
// the k nearest gallery codes, the smallest label winning a tied vote.
//
//   1. row norms |q|^2, |g|^2                                     (knn_rownorm_kernel)
//   2. d2[q][g] = |q|^2 + |g|^2 - 2 q.g  -- an NT GEMM, K = 15,872 -- on v_mfma_f32_32x32x2_f32, 64x64 tiles (knn_dist_kernel)
//   3. per probe: the k smallest d2 (ties -> lower gallery index), then the vote            (knn_vote_kernel)
// fp32 throughout; the Gram form loses ~1e-6 |q||g| absolutely, which can reorder neighbours whose distances agree to
// 6 digits (sklearn evaluates the same expansion in float64).
#include "common.h"

namespace {

constexpr int KT = 32;         // K tile
constexpr int LDT = KT + 1;    // LDS row stride (floats): 32 rows -> 32 different banks for the per-lane operand reads

__global__ __launch_bounds__(256) void knn_rownorm_kernel(const float* __restrict__ x, float* __restrict__ nrm, int rows, int d) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = x + (size_t)row * d;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) s += p[k] * p[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) nrm[row] = s;
}

// d2[Q][G] tile 64 x 64 per 256-thread workgroup; wave (wm, wn) of the 2 x 2 owns a 32 x 32 accumulator.
__global__ __launch_bounds__(256) void knn_dist_kernel(const float* __restrict__ q, const float* __restrict__ g,
                                                       const float* __restrict__ qn, const float* __restrict__ gn,
                                                       float* __restrict__ d2, int nq, int ng, int d) {
  __shared__ float sA[64 * LDT];
  __shared__ float sB[64 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int q0 = blockIdx.y * 64, g0 = blockIdx.x * 64;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // staging role: row tid/4 (0..63), 8 consecutive k at (tid%4)*8
  const int sr = tid >> 2, sk = (tid & 3) * 8;
  const int qa = q0 + sr < nq ? q0 + sr : nq - 1, ga = g0 + sr < ng ? g0 + sr : ng - 1;   // clamped rows are never stored
  const float* pa = q + (size_t)qa * d + sk;
  const float* pb = g + (size_t)ga * d + sk;
  // software pipeline: the global loads of K-tile t+1 are in flight while tile t is multiplied out of LDS
  float4 va[2], vb[2];
  const bool vec_ok = (d & 3) == 0;   // rows are 16-byte aligned only then
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kk = k0 + sk + 4 * h;
      if (vec_ok && kk + 3 < d) {
        va[h] = *reinterpret_cast<const float4*>(pa + k0 + 4 * h);
        vb[h] = *reinterpret_cast<const float4*>(pb + k0 + 4 * h);
      } else {   // ragged tail of K or unaligned rows (the signature path has d = 15,872)
        float ta[4], tb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ta[e] = kk + e < d ? pa[k0 + 4 * h + e] : 0.f;
          tb[e] = kk + e < d ? pb[k0 + 4 * h + e] : 0.f;
        }
        va[h] = make_float4(ta[0], ta[1], ta[2], ta[3]);
        vb[h] = make_float4(tb[0], tb[1], tb[2], tb[3]);
      }
    }
  };
  load_tile(0);
  for (int k0 = 0; k0 < d; k0 += KT) {
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float* da = sA + sr * LDT + sk + 4 * h;
      float* db = sB + sr * LDT + sk + 4 * h;
      da[0] = va[h].x; da[1] = va[h].y; da[2] = va[h].z; da[3] = va[h].w;
      db[0] = vb[h].x; db[1] = vb[h].y; db[2] = vb[h].z; db[3] = vb[h].w;
    }
    __syncthreads();
    if (k0 + KT < d) load_tile(k0 + KT);
#pragma unroll
    for (int s = 0; s < KT / 2; ++s)
      acc = ugn_mfma(sA[(wm * 32 + li) * LDT + 2 * s + lh], sB[(wn * 32 + li) * LDT + 2 * s + lh], acc);
  }
  const int col = g0 + wn * 32 + li;
  if (col < ng) {
    const float gnv = gn[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = q0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < nq) {
        const float v = qn[row] + gnv - 2.f * acc[r];
        d2[(size_t)row * ng + col] = v > 0.f ? v : 0.f;
      }
    }
  }
}

constexpr int KMAX = 16;

// one wave per probe: every lane keeps the k best of its strided share of the row, then k rounds of wave-wide argmin
__global__ __launch_bounds__(64) void knn_vote_kernel(const float* __restrict__ d2, const int32_t* __restrict__ labels,
                                                      int32_t* __restrict__ pred, int32_t* __restrict__ nbr, int ng, int k) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* p = d2 + (size_t)row * ng;
  float bd[KMAX];
  int bi[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { bd[j] = __builtin_huge_valf(); bi[j] = 0x7fffffff; }
  for (int c = lane; c < ng; c += 64) {
    float v = p[c];
    int vi = c;
    // insertion into the ascending list (stable: an equal distance stays behind the earlier index)
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < k) {
        const bool lt = v < bd[j] || (v == bd[j] && vi < bi[j]);
        const float td = lt ? bd[j] : v;
        const int ti = lt ? bi[j] : vi;
        bd[j] = lt ? v : bd[j];
        bi[j] = lt ? vi : bi[j];
        v = td; vi = ti;
      }
    }
  }
  int win_lab[KMAX];
  for (int round = 0; round < k; ++round) {
    float v = bd[0];
    int vi = bi[0];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(v, off);
      const int oi = __shfl_xor(vi, off);
      if (ov < v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
    // the owner pops its head
    if (bi[0] == vi && bd[0] == v) {
#pragma unroll
      for (int j = 0; j + 1 < KMAX; ++j) { bd[j] = bd[j + 1]; bi[j] = bi[j + 1]; }
      bd[KMAX - 1] = __builtin_huge_valf(); bi[KMAX - 1] = 0x7fffffff;
    }
    const int lab = vi < ng ? labels[vi] : 0x7fffffff;
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if (j == round) win_lab[j] = lab;
    if (lane == 0 && nbr) nbr[(size_t)row * k + round] = vi;
  }
  if (lane == 0) {
    int best_lab = 0x7fffffff, best_cnt = 0;
#pragma unroll
    for (int a = 0; a < KMAX; ++a) {
      if (a < k) {
        int cnt = 0;
#pragma unroll
        for (int b = 0; b < KMAX; ++b)
          if (b < k && win_lab[b] == win_lab[a]) ++cnt;
        if (cnt > best_cnt || (cnt == best_cnt && win_lab[a] < best_lab)) { best_cnt = cnt; best_lab = win_lab[a]; }
      }
    }
    pred[row] = best_lab;
  }
}

}  // namespace

extern "C" size_t ugn_knn_ws(int ngallery, int nprobe) {
  if (ngallery <= 0 || nprobe <= 0) return 0;
  return ((size_t)ngallery * nprobe + ngallery + nprobe) * sizeof(float);
}

extern "C" int ugn_knn_predict(const float* gallery, const int32_t* gallery_labels, const float* probes, int ngallery,
                               int nprobe, int dim, int k, int32_t* pred, int32_t* neighbours, void* ws, size_t ws_bytes,
                               void* stream) {
  UGN_REQUIRE(gallery && gallery_labels && probes && pred && ws, "ugn_knn_predict: null pointer");
  UGN_REQUIRE(ngallery > 0 && nprobe > 0 && dim > 0, "ugn_knn_predict: empty gallery, probe set or dimension");
  UGN_REQUIRE(k >= 1 && k <= KMAX && k <= ngallery, "ugn_knn_predict: k must be in 1..min(%d, ngallery) (got %d)", KMAX, k);
  UGN_REQUIRE(ws_bytes >= ugn_knn_ws(ngallery, nprobe), "ugn_knn_predict: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* d2 = (float*)ws;
  float* gn = d2 + (size_t)ngallery * nprobe;
  float* qn = gn + ngallery;
  hipLaunchKernelGGL(knn_rownorm_kernel, dim3((ngallery + 3) / 4), dim3(256), 0, st, gallery, gn, ngallery, dim);
  hipLaunchKernelGGL(knn_rownorm_kernel, dim3((nprobe + 3) / 4), dim3(256), 0, st, probes, qn, nprobe, dim);
  UGN_CHECK_LAUNCH("knn row norms");
  hipLaunchKernelGGL(knn_dist_kernel, dim3((ngallery + 63) / 64, (nprobe + 63) / 64), dim3(256), 0, st, probes, gallery,
                     (const float*)qn, (const float*)gn, d2, nprobe, ngallery, dim);
  UGN_CHECK_LAUNCH("knn distances");
  hipLaunchKernelGGL(knn_vote_kernel, dim3(nprobe), dim3(64), 0, st, (const float*)d2, gallery_labels, pred, neighbours, ngallery,
                     k);
  UGN_CHECK_LAUNCH("knn vote");
  return 0;
}
