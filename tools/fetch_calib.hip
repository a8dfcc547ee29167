// FETCH_SIZE calibration for LDS-DMA streams (VERDICT r03 item 7 / Weak #11): does the guide's x2 correction of FETCH_SIZE on
// gfx950 (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read") also apply when a
// wave's global_load_lds_dwordx4 fetches 64-BYTE SEGMENTS -- what the K >= 64 convolution kernels do (a 32-channel chunk of a
// 64- / 128-channel planar H2 pixel = 64 B of the H plane + 64 B of the L plane, the other halves of those 128-B lines fetched by the
// NEXT chunk's DMA)?  Three kernels over the same 2 GiB buffer (>> L2 + Infinity Cache), known bytes each:
//   full   : every byte once, 1 KiB contiguous per wave instruction                              (bytes = N)
//   half   : bytes [0, 64) of every 128-B line, 64-B segments (4 lanes x 16 B), the rest never     (bytes = N / 2 requested)
//   halves : [0, 64) of every line of a 32 KiB block, then [64, 128) of the same block             (bytes = N requested, second
//            halves a few microseconds later, the line still in L2: the access pattern of two consecutive K chunks)
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/_bin/fetch_calib tools/fetch_calib.hip
//               rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- tools/_bin/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// MODE 0 full, 1 half, 2 halves.  256 threads, each workgroup walks 4 MiB (halves: 32 KiB) blocks; per wave instruction 1 KiB lands in LDS.
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ buf, size_t nbytes, unsigned* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 4096];
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t BLK = MODE == 2 ? (32u << 10) : (4u << 20);
  unsigned acc = 0;
  for (size_t b = (size_t)blockIdx.x * BLK; b < nbytes; b += (size_t)gridDim.x * BLK) {
    for (int pass = 0; pass < (MODE == 2 ? 2 : 1); ++pass) {
      // a wave instruction covers 64 lanes x 16 B: MODE 0 = 1 KiB of consecutive bytes; MODE 1/2 = sixteen 64-B segments, one per
      // 128-B line (lane l: line l / 4, quarter l % 4 of the chosen half)
      const size_t span = MODE == 0 ? 1024 : 2048;                 // buffer bytes a wave instruction walks over
      for (size_t o = (size_t)wave * span; o < BLK; o += 4 * span * 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t w = b + o + (size_t)u * 4 * span;
          const size_t off = MODE == 0 ? w + (size_t)lane * 16 : w + (size_t)(lane >> 2) * 128 + (size_t)pass * 64 + (size_t)(lane & 3) * 16;
          dma16(buf + off, base + (unsigned)(wave * 4096 + u * 1024));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *reinterpret_cast<const unsigned*>(lds + wave * 4096 + lane * 4);
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const size_t N = 2ull << 30;
  char* buf;
  unsigned* sink;
  if (hipMalloc(&buf, N) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, N);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream_kernel<0>, dim3(512), dim3(256), 0, 0, buf, N, sink);
    hipLaunchKernelGGL(stream_kernel<1>, dim3(512), dim3(256), 0, 0, buf, N, sink);
    hipLaunchKernelGGL(stream_kernel<2>, dim3(512), dim3(256), 0, 0, buf, N, sink);
  }
  hipDeviceSynchronize();
  printf("bytes per launch: full %zu  half %zu requested (%zu if whole lines)  halves %zu\n", N, N / 2, N, N);
  return 0;
}
