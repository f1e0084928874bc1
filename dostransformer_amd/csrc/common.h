// Shared device helpers for libdosx (gfx950 / CDNA4 only: wave64, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dosx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DOSX_WAVE 64

// Wave priority of the kernels of the dgrad / forward chain (everything except the weight-gradient kernels, which keep the
// default 0): where a chain kernel shares a SIMD with a weight-gradient workgroup, the instruction arbiter then issues the
// chain's MFMAs first and the weight gradients fill the issue slots the chain leaves (barriers, load waits).
#ifndef DOSX_MAIN_PRIO
#define DOSX_MAIN_PRIO 0
#endif
#define DOSX_SET_MAIN_PRIO() do { if (DOSX_MAIN_PRIO) __builtin_amdgcn_s_setprio(DOSX_MAIN_PRIO); } while (0)

void dosx_set_error(const char* fmt, ...);

#define DOSX_CHECK_ARG(cond, ...)   \
  do {                              \
    if (!(cond)) {                  \
      dosx_set_error(__VA_ARGS__);  \
      return -22;                   \
    }                               \
  } while (0)

#define DOSX_LAUNCH_CHECK()                                                    \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      dosx_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return -5;                                                               \
    }                                                                          \
  } while (0)

static inline hipStream_t to_stream(dosx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ int dosx_map_row(const DosxRowMap& rm, int r) {
  int t;
  if (rm.d >= (1 << 30)) t = r * rm.c + rm.off;     // no div/mod part (identity maps): skip the two integer divisions
  else t = (r / rm.d) * rm.m + (r % rm.d) * rm.c + rm.off;
  return rm.idx ? rm.idx[t] : t;
}

// Wave64 all-reduce on the DPP path.  __shfl_xor() lowers to ds_bpermute_b32 (an LDS-pipe round
// trip, ~100+ cycles per step, 6 dependent steps): measured 1.4-1.6k cycles per LayerNorm row in
// the row-wise epilogues.  Here: 4 single-instruction DPP steps make every lane of a 16-lane row hold
// its row total, then the 4 row totals are read with v_readlane and combined on uniform registers.
//   quad_perm[1,0,3,2] = xor 1 ; quad_perm[2,3,0,1] = xor 2 ; row_half_mirror / row_mirror exchange
//   the (already uniform) quads / halves of a row.
#define DOSX_DPP_F(v, ctrl) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, false))
__device__ __forceinline__ float wave_sum(float v) {
  v += DOSX_DPP_F(v, 0xB1);    // quad_perm [1,0,3,2]
  v += DOSX_DPP_F(v, 0x4E);    // quad_perm [2,3,0,1]
  v += DOSX_DPP_F(v, 0x141);   // row_half_mirror
  v += DOSX_DPP_F(v, 0x140);   // row_mirror
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, DOSX_DPP_F(v, 0xB1));
  v = fmaxf(v, DOSX_DPP_F(v, 0x4E));
  v = fmaxf(v, DOSX_DPP_F(v, 0x141));
  v = fmaxf(v, DOSX_DPP_F(v, 0x140));
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Sum / max over each 16-lane DPP row (quarter wave): 4 single-instruction steps, the result lands in
// all 16 lanes, no cross-row step and no v_readlane.  The row phases of attention.hip give every query
// row one quarter wave, so a wave reduces 4 rows per instruction sequence.
__device__ __forceinline__ float row16_sum(float v) {
  v += DOSX_DPP_F(v, 0xB1);
  v += DOSX_DPP_F(v, 0x4E);
  v += DOSX_DPP_F(v, 0x141);
  v += DOSX_DPP_F(v, 0x140);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, DOSX_DPP_F(v, 0xB1));
  v = fmaxf(v, DOSX_DPP_F(v, 0x4E));
  v = fmaxf(v, DOSX_DPP_F(v, 0x141));
  v = fmaxf(v, DOSX_DPP_F(v, 0x140));
  return v;
}

// x / d for a wave-uniform run-time d: a shift when d is a power of two (hidden 64 / 128: every divisor of the attention tiles'
// job and staging index arithmetic is), else the division.  A run-time integer division is ~40 instructions; the tile code did
// 8-16 of them per thread and launch (tools/stamp_attn_aligned.py: 5.5 -> 4.9 k clk in the stand-alone attention kernels).
struct UDiv {
  int d, sh;
  __device__ __forceinline__ explicit UDiv(int d_) : d(d_), sh((d_ & (d_ - 1)) == 0 ? __builtin_ctz(d_) : -1) {}
  __device__ __forceinline__ int div(int x) const { return sh >= 0 ? (x >> sh) : x / d; }
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float f4get(const float4& v, int i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// Row statistics of a row held as `per` float4 per lane of one wave (whole row over 64 lanes).
// Two-pass (mean, then centred second moment) like torch's LayerNorm; eps = 1e-5.
#define DOSX_LN_EPS 1e-5f

// ---- in-launch reductions: publish / ticket / read-back (gemm.hip wgrad_finish + EPI_SEGSUM, attention.hip dK/dV) ----------
// Protocol: a workgroup PUBLISHES its partial result with sc1 stores (buffer stores with aux bit 4, or agent-scope relaxed
// atomic stores - on gfx942 / gfx950 both are `... sc1`: written through to the agent-coherent level, not left dirty in this
// XCD's L2), every storing wave DRAINS (`s_waitcnt vmcnt(0)`: its stores have completed there), a workgroup barrier, then ONE
// lane draws a TICKET with an agent-scope fetch_add; the workgroup that draws the last ticket READS the partials BACK with
// sc1 loads (agent-scope: served from the coherent level, never from a stale line of this CU's L1 / this XCD's L2).
// That is the agent-scope release / acquire pair of the gfx942 memory model spelled out access by access (LLVM AMDGPU usage,
// memory model gfx942: agent-scope atomic load / store = `sc1=1`; release = complete prior stores before the atomic, which
// the drain does for the write-through stores; acquire = later loads must not hit stale lines, which sc1 loads do not) -
// instead of the whole-cache `buffer_wbl2 sc1` / `buffer_inv sc1` an __ATOMIC_ACQ_REL ticket would emit.  Build with
// -DDOSX_TICKET_ORDER=__ATOMIC_ACQ_REL to get exactly that stronger form (measured cost: DESIGN.md §2, round 4); the stress
// tests (tests/test_gpu_gemm.py, tests/test_gpu_graph.py: thousands of launches under a bandwidth hog, every one compared bitwise) run on either.
#ifndef DOSX_TICKET_ORDER
#define DOSX_TICKET_ORDER __ATOMIC_RELAXED
#endif
__device__ __forceinline__ int dosx_ticket(int* counter) {
  return __hip_atomic_fetch_add(counter, 1, DOSX_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
}
