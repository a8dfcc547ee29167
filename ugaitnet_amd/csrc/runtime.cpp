// Process-wide state of libugaitnet_hip.so besides the error message: the persistent grid (ugn_set_persistent_wgs) and the 256-byte
// block of zeros the staging loads of the 3x3 kernels read for out-of-image pixels.  Shared by every kernel set (x3, bf16, Winograd,
// and the opt-in f16x2 set), so it lives in a file of its own (round 6: the f16x2 sources are no longer part of the default build).
#include "mm_common.h"

namespace {
// Persistent workgroups of the forward / data-gradient launches (one per CU by default).  Under data parallelism RCCL's channels
// need CUs of their own while the backward pass still runs: ugn_set_persistent_wgs(n < 256) leaves 256 - n of them free.  Results
// do not depend on it (an item's arithmetic is the same whichever workgroup runs it).
int g_persistent_wgs = ugn_mm::kGrid;
}  // namespace

int ugn_mm::persistent_wgs() { return g_persistent_wgs; }

const void* ugn_mm::zero_block() {
  static void* z = nullptr;
  if (!z) {
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) return nullptr;
    z = p;
  }
  return z;
}

extern "C" int ugn_set_persistent_wgs(int n) {
  UGN_REQUIRE(n == 0 || (n >= 8 && n <= ugn_mm::kGrid), "ugn_set_persistent_wgs: 8..%d workgroups, or 0 for the default (got %d)",
              ugn_mm::kGrid, n);
  g_persistent_wgs = n == 0 ? ugn_mm::kGrid : n;
  return 0;
}

extern "C" int ugn_get_persistent_wgs(void) { return g_persistent_wgs; }
