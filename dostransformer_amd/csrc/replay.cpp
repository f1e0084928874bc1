// Host-side replay of a recorded launch list (include/dosx.h: dosx_replay).
//
// A training step on a fixed shape bucket is the same ~220 library calls every time.  `Trainer(replay=True)` records
// them once (function pointer + marshalled arguments on static buffers, stream fork/join events included) and later
// re-issues the list; doing that from Python costs ~6.6 us per entry in ctypes (1.5 ms per step, 80 % of the GPU time
// of the step).  This loop does it in C: ~0.3 us per entry on top of the launch itself.
//
// Calling convention (x86-64 System V, the only host this library targets): integer-class arguments (pointers,
// int32/int64/size_t, by-pointer descriptors) travel in rdi, rsi, rdx, rcx, r8, r9 and then on the stack in order;
// float / double arguments travel in xmm0.. independently of the integer ones.  So a recorded call is its integer
// arguments in order plus its floating-point arguments in order, and the callee is invoked through a function-pointer
// type with MAXI integer parameters followed by the floating-point ones: surplus integer arguments are ignored by the
// callee (caller-cleaned stack).  `kind` selects the floating-point signature class.
#include <stdint.h>

#include "../../include/dosx.h"

void dosx_set_error(const char* fmt, ...);

#define I19 int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, \
            int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t
#define A19(c) c.iarg[0], c.iarg[1], c.iarg[2], c.iarg[3], c.iarg[4], c.iarg[5], c.iarg[6], c.iarg[7], c.iarg[8], c.iarg[9], \
               c.iarg[10], c.iarg[11], c.iarg[12], c.iarg[13], c.iarg[14], c.iarg[15], c.iarg[16], c.iarg[17], c.iarg[18]

extern "C" int dosx_replay(const DosxCall* calls, int n, int* failed_index) {
  for (int i = 0; i < n; ++i) {
    const DosxCall& c = calls[i];
    int rc;
    switch (c.kind) {
      case DOSX_CALL_INTS:
        rc = reinterpret_cast<int (*)(I19)>(c.fn)(A19(c));
        break;
      case DOSX_CALL_F1:
        rc = reinterpret_cast<int (*)(I19, float)>(c.fn)(A19(c), (float)c.farg[0]);
        break;
      case DOSX_CALL_F1D1:
        rc = reinterpret_cast<int (*)(I19, float, double)>(c.fn)(A19(c), (float)c.farg[0], c.farg[1]);
        break;
      case DOSX_CALL_F6:
        rc = reinterpret_cast<int (*)(I19, float, float, float, float, float, float)>(c.fn)(
            A19(c), (float)c.farg[0], (float)c.farg[1], (float)c.farg[2], (float)c.farg[3], (float)c.farg[4], (float)c.farg[5]);
        break;
      default:
        dosx_set_error("dosx_replay: entry %d has unknown kind %d", i, c.kind);
        if (failed_index) *failed_index = i;
        return -22;
    }
    if (rc != 0) {
      if (failed_index) *failed_index = i;
      return rc;
    }
  }
  return 0;
}
