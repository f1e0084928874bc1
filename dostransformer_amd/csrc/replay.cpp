// Host-side replay of a recorded launch list (include/dosx.h: dosx_replay).
//
// A training step on a fixed shape bucket is the same ~200 library calls every time.  `Trainer(replay=True)` records
// them once (entry point + marshalled arguments on static buffers, stream fork/join events included) and later
// re-issues the list; doing that from Python costs ~6.6 us per entry in ctypes (1.5 ms per step, as much as the GPU
// time of the step).  This loop does it in C: ~0.3 us per entry on top of the launch itself.
//
// Every entry names its callee by an OP index (dosx_replay_op("dosx_gemm") ...) and carries its integer-class
// arguments (pointers, integers, by-pointer descriptors) and its floating-point arguments, each in declaration order.
// The callee is reached through a TYPED thunk generated from include/dosx.h (tools/gen_replay_thunks.py ->
// replay_thunks.inc): an ordinary C call with every argument cast to its declared parameter type, so nothing depends
// on how a particular ABI passes surplus or mismatched arguments.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dosx.h"

void dosx_set_error(const char* fmt, ...);

namespace {

struct DosxThunk {
  const char* name;
  int (*call)(const DosxCall&);
  int nint, nflt;
};

// stream fork / join of the recorded program (GradSink): plain HIP runtime calls on raw handles
int thunk_hipEventRecord(const DosxCall& c) {
  return (int)hipEventRecord((hipEvent_t)(uintptr_t)c.iarg[0], (hipStream_t)(uintptr_t)c.iarg[1]);
}
int thunk_hipStreamWaitEvent(const DosxCall& c) {
  return (int)hipStreamWaitEvent((hipStream_t)(uintptr_t)c.iarg[0], (hipEvent_t)(uintptr_t)c.iarg[1], (unsigned)c.iarg[2]);
}

#include "replay_thunks.inc"

constexpr int kNumThunks = (int)(sizeof(kThunks) / sizeof(kThunks[0]));
const DosxThunk kHipThunks[] = {
    {"hipEventRecord", thunk_hipEventRecord, 2, 0},
    {"hipStreamWaitEvent", thunk_hipStreamWaitEvent, 3, 0},
};
constexpr int kNumHip = 2;

inline const DosxThunk* thunk_of(int op) {
  if (op >= 0 && op < kNumThunks) return &kThunks[op];
  if (op >= DOSX_OP_HIP_BASE && op < DOSX_OP_HIP_BASE + kNumHip) return &kHipThunks[op - DOSX_OP_HIP_BASE];
  return nullptr;
}

}  // namespace

extern "C" int dosx_replay_op(const char* name, int* n_int, int* n_float) {
  if (!name) return -1;
  for (int i = 0; i < kNumThunks + kNumHip; ++i) {
    const int op = i < kNumThunks ? i : DOSX_OP_HIP_BASE + (i - kNumThunks);
    const DosxThunk* t = thunk_of(op);
    if (strcmp(t->name, name) == 0) {
      if (n_int) *n_int = t->nint;
      if (n_float) *n_float = t->nflt;
      return op;
    }
  }
  return -1;
}

// Diagnostic twin of dosx_replay: the same list, with a HIP event pair around every libdosx entry on the stream that entry
// launches on (its last integer-class argument, by the convention of this header), so the duration of every launch is
// measured INSIDE the replayed two-stream step - the configuration bench.py times - rather than in a separate eager
// pass.  Synchronises the device at the end (it is a measurement call, not a product path); ms_out[i] = elapsed time of
// entry i in milliseconds (0 for the stream fork / join entries).
extern "C" int dosx_replay_timed(const DosxCall* calls, int n, float* ms_out, int* failed_index) {
  if (n <= 0) return 0;
  hipEvent_t* ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * (size_t)n);
  if (!ev) { dosx_set_error("dosx_replay_timed: out of host memory"); return -12; }
  int made = 0, rc = 0;
  for (; made < 2 * n; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) { rc = -5; dosx_set_error("dosx_replay_timed: hipEventCreate failed"); break; }
  for (int i = 0; rc == 0 && i < n; ++i) {
    const DosxCall& c = calls[i];
    const DosxThunk* t = thunk_of(c.op);
    if (!t || c.nint != t->nint || c.nflt != t->nflt) {
      dosx_set_error("dosx_replay_timed: entry %d is malformed (op %d)", i, c.op);
      rc = -22;
    } else {
      const bool timed = c.op < DOSX_OP_HIP_BASE && c.nint > 0;
      hipStream_t st = timed ? (hipStream_t)(uintptr_t)c.iarg[c.nint - 1] : nullptr;
      if (timed) (void)hipEventRecord(ev[2 * i], st);
      rc = t->call(c);
      if (timed) (void)hipEventRecord(ev[2 * i + 1], st);
    }
    if (rc != 0 && failed_index) *failed_index = i;
  }
  (void)hipDeviceSynchronize();
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    if (rc == 0 && ms_out && calls[i].op < DOSX_OP_HIP_BASE && calls[i].nint > 0)
      if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) ms = 0.f;
    if (ms_out) ms_out[i] = ms;
  }
  for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
  free(ev);
  return rc;
}

extern "C" int dosx_replay(const DosxCall* calls, int n, int* failed_index) {
  for (int i = 0; i < n; ++i) {
    const DosxCall& c = calls[i];
    const DosxThunk* t = thunk_of(c.op);
    int rc;
    if (!t) {
      dosx_set_error("dosx_replay: entry %d has unknown op %d", i, c.op);
      rc = -22;
    } else if (c.nint != t->nint || c.nflt != t->nflt) {
      dosx_set_error("dosx_replay: entry %d (%s) carries %d+%d arguments, the entry point takes %d+%d", i, t->name, c.nint,
                     c.nflt, t->nint, t->nflt);
      rc = -22;
    } else {
      rc = t->call(c);
    }
    if (rc != 0) {
      if (failed_index) *failed_index = i;
      return rc;
    }
  }
  return 0;
}
