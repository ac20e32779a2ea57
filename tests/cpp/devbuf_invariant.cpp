// DevBuf invariant (hdp_internal.hpp): bytes == 0 whenever p == nullptr, also after a FAILED allocation -- the
// plan-owned scratch buffers are guarded by "bytes >= need", so a failure must leave the buffer empty.  Runs on the
// CPU: without a HIP device every hipMalloc fails, which is exactly the case under test.
#include "hdp_internal.hpp"

#include <cstdio>

int main() {
  hdp::DevBuf b;
  int bad = 0;
  if (b.p != nullptr || b.bytes != 0) bad |= 1;
  const hipError_t e = b.alloc(size_t(1) << 20);
  if (e != hipSuccess) {
    if (b.p != nullptr || b.bytes != 0) bad |= 2;                        // failed: must stay empty
    if (b.alloc(0) != hipSuccess || b.p != nullptr || b.bytes != 0) bad |= 4;
    printf("alloc failed as expected without a device (%s); invariant %s\n", hipGetErrorString(e), bad ? "BROKEN" : "holds");
  } else {
    if (b.p == nullptr || b.bytes != (size_t(1) << 20)) bad |= 8;         // a device is present: ordinary success path
    b.release();
    if (b.p != nullptr || b.bytes != 0) bad |= 16;
    printf("alloc succeeded (a device is visible); invariant %s\n", bad ? "BROKEN" : "holds");
  }
  return bad;
}
