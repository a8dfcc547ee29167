// Operand layout of v_smfmac_f32_16x16x64_f16 (gfx950), found by trying hypotheses against a host product.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/smfmac_probe.hip -o /tmp/smfmac_probe && /tmp/smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const h8* a, const h16* b, const int* idx, f4* c) {
  const int l = threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a[l], b[l], acc, idx[l], 0, 0);
  c[l] = acc;
}
int main() {
  std::vector<_Float16> A(64 * 8), B(64 * 16);
  std::vector<int> IDX(64);
  std::vector<float> C(64 * 4);
  srand(7);
  // compressed A: value of (lane, slot); the two indices of a pair ascending (as the ISA requires), all 6 combinations occur
  static const int combos[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
  int pos[64][8];
  for (int l = 0; l < 64; ++l) {
    int w = 0;
    for (int g = 0; g < 4; ++g) {
      const int* cb = combos[rand() % 6];
      pos[l][2 * g] = cb[0]; pos[l][2 * g + 1] = cb[1];
      w |= (cb[0] | (cb[1] << 2)) << (4 * g);
    }
    IDX[l] = w;
    for (int s = 0; s < 8; ++s) A[l * 8 + s] = (_Float16)((rand() % 17) - 8);
  }
  for (size_t i = 0; i < B.size(); ++i) B[i] = (_Float16)((rand() % 9) - 4);
  h8* da; h16* db; int* di; f4* dc;
  hipMalloc(&da, 64 * sizeof(h8)); hipMalloc(&db, 64 * sizeof(h16)); hipMalloc(&di, 64 * 4); hipMalloc(&dc, 64 * sizeof(f4));
  hipMemcpy(da, A.data(), 64 * sizeof(h8), hipMemcpyHostToDevice);
  hipMemcpy(db, B.data(), 64 * sizeof(h16), hipMemcpyHostToDevice);
  hipMemcpy(di, IDX.data(), 64 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, di, dc);
  if (hipMemcpy(C.data(), dc, 64 * sizeof(f4), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
  // hypotheses: ha = A slot -> dense k, hb = B element -> k
  for (int ha = 0; ha < 3; ++ha)
    for (int hb = 0; hb < 3; ++hb) {
      double Ad[16][64] = {}, Bd[64][16] = {};
      for (int l = 0; l < 64; ++l) {
        const int m = l & 15, kb = l >> 4;
        for (int s = 0; s < 8; ++s) {
          const int g = s >> 1, p = pos[l][s];
          int kk;
          if (ha == 0) kk = 16 * kb + 4 * g + p;                                  // the lane's 16 consecutive k
          else if (ha == 1) kk = 32 * (s >> 2) + 8 * kb + 4 * (g & 1) + p;      // two K = 32 halves, 8 k each
          else kk = 4 * (4 * g + kb) + p;                                         // groups interleaved over the k blocks
          Ad[m][kk] += (double)A[l * 8 + s];
        }
        for (int i = 0; i < 16; ++i) {
          int kk;
          if (hb == 0) kk = 16 * kb + i;
          else if (hb == 1) kk = 32 * (i >> 3) + 8 * kb + (i & 7);
          else kk = 16 * (i >> 2) + 4 * kb + (i & 3);
          Bd[kk][l & 15] = (double)B[l * 16 + i];
        }
      }
      double worst = 0;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
          const int m = 4 * (l >> 4) + r, n = l & 15;
          double ref = 0;
          for (int kk = 0; kk < 64; ++kk) ref += Ad[m][kk] * Bd[kk][n];
          worst = fmax(worst, fabs(ref - C[l * 4 + r]));
        }
      printf("A hypothesis %d, B hypothesis %d: max |err| = %g\n", ha, hb, worst);
    }
  return 0;
}
