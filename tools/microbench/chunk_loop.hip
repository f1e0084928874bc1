// Microbenchmark: where the cycles of a wave-specialised k-chunk loop go (the structure of wgrad_body / gemm_kernel /
// ffn_*_kernel: 4 matrix waves + 4 staging waves, one barrier per 32-row chunk, three LDS stage buffers).
// Each MODE adds one ingredient; prints s_memtime ticks per chunk (workgroup 0) and whole-chip TFLOP/s.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/chunk_loop.hip -o tools/microbench/chunk_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
constexpr int LDT = 68, STG = 128 * 36;     // (>= 2 * 32 * 68; the k-contiguous variant holds [128][36])

// MODE bits: 1 = barrier per chunk, 2 = staging waves store 4 float4 per lane per chunk, 4 = staging waves load them from
// global memory (2 chunks deep), 8 = staging waves run 8 dependent VALU ops per chunk, 16 = matrix waves read their fragments
// from LDS (else registers only), 32 = staging waves run 24 VALU ops per chunk
template <int MODE, int MPC /* MFMAs per chunk and wave */>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* out, unsigned long long* ticks, int nch, int rows) {
  __shared__ __align__(16) float Sm[3 * STG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  for (int i = tid; i < 3 * STG; i += 512) Sm[i] = (float)(i & 1023) * 1e-6f;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  if (wave >= 4) {
    const int st = tid - 256, r = st >> 3, c4 = (st & 7) * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    // MODE & 512: every workgroup streams its own rows (blocks * nch * 16 KB in total: HBM / L2 traffic like the real kernels');
    // else 64 distinct row blocks, L2-resident
    const uint32_t vo = (MODE & 512) ? (uint32_t)((r * 128 + c4) * 4) : (uint32_t)(((size_t)(blockIdx.x % 64) * 32 + r) * 128 + c4) * 4;
    float4 s0[4], s1[4];
    float x = (float)lane;
    auto issue = [&](float4 (&s)[4], int c) {
      if (MODE & 4) {
        const int so = (MODE & 512) ? __builtin_amdgcn_readfirstlane((int)((((size_t)blockIdx.x * (nch + 8) + c) % (rows / 32)) * 32 * 128 * 4))
                                    : __builtin_amdgcn_readfirstlane((c % (rows / 32)) * 32 * 128 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 128 * j, so, 0));
      }
    };
    auto store = [&](float* buf, float4 (&s)[4]) {
      if (MODE & 128) {                                     // transposed: [column][row], 16 ds_write_b32 per lane and chunk
        constexpr int LDK = (MODE & 256) ? 34 : 36;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* d = &buf[((j >> 1) * 64 + c4 + 32 * (j & 1)) * LDK + r];
          d[0] = s[j].x; d[LDK] = s[j].y; d[2 * LDK] = s[j].z; d[3 * LDK] = s[j].w;
        }
      } else if (MODE & 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&buf[(j >> 1) * 32 * LDT + r * LDT + c4 + 32 * (j & 1)]) = s[j];
      }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) { s0[j] = make_float4(1, 2, 3, 4); s1[j] = make_float4(4, 3, 2, 1); }
    issue(s0, 0); issue(s1, 1);
    if (MODE & 1) __syncthreads();
    int b2 = 2;
    for (int c = 0; c < nch; c += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float4 (&q)[4] = u ? s1 : s0;
        store(Sm + b2 * STG, q);
        issue(q, c + u + 4);
        if (MODE & 8) {
#pragma unroll
          for (int v = 0; v < 8; ++v) x = x * 1.0001f + 0.5f;
        }
        if (MODE & 32) {
#pragma unroll
          for (int v = 0; v < 24; ++v) x = x * 1.0001f + 0.5f;
        }
        b2 = b2 == 2 ? 0 : b2 + 1;
        if (MODE & 1) __syncthreads();
      }
    }
    out[blockIdx.x * 512 + tid] = x + s0[0].x + s1[0].x;
  } else {
    const int wn = wave >> 1, wk = wave & 1;
    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
    if (MODE & 1) __syncthreads();
    constexpr int H = MPC / 2;       // MFMAs per half chunk
    struct Frag { float a[H], b[H]; };
    auto fetch = [&](Frag& f, int buf, int half) {
      const float* Ys = Sm + buf * STG;
      const float* Xs = Ys + 32 * LDT;
      if (MODE & 256) {                                     // k-contiguous, rows of 34 floats: ds_read_b64, 2 MFMAs each
#pragma unroll
        for (int i = 0; i < H; i += 2) {
          const int kk = ((H * 2 * half) & 31) + 2 * (i & ~3) + 4 * hh + (i & 2);
          const float2 va = *reinterpret_cast<const float2*>(&Ys[(wn * 32 + l31) * 34 + kk]);
          const float2 vb = *reinterpret_cast<const float2*>(&Ys[(64 + wk * 32 + l31) * 34 + kk]);
          f.a[i] = va.x; f.a[i + 1] = va.y; f.b[i] = vb.x; f.b[i + 1] = vb.y;
        }
        return;
      }
      if (MODE & 64) {                                      // operands k-contiguous in LDS ([n][36]): one ds_read_b128 feeds 4 MFMAs
#pragma unroll
        for (int i = 0; i < H; i += 4) {
          const int kk = ((H * 2 * half) & 31) + 2 * i + 4 * hh;      // (k permuted inside a group of 8: both operands alike)
          const float4 va = *reinterpret_cast<const float4*>(&Ys[(wn * 32 + l31) * 36 + kk]);
          const float4 vb = *reinterpret_cast<const float4*>(&Ys[(64 + wk * 32 + l31) * 36 + kk]);
          f.a[i] = va.x; f.a[i + 1] = va.y; f.a[i + 2] = va.z; f.a[i + 3] = va.w;
          f.b[i] = vb.x; f.b[i + 1] = vb.y; f.b[i + 2] = vb.z; f.b[i + 3] = vb.w;
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const int row = (16 * half + 2 * i + hh) & 31;
        if (MODE & 16) { f.a[i] = Ys[row * LDT + wn * 32 + l31]; f.b[i] = Xs[row * LDT + wk * 32 + l31]; }
        else { f.a[i] = (float)(row + lane); f.b[i] = (float)(row - lane); }
      }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
      for (int i = 0; i < H; i += 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i], f.b[i], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i + 1], f.b[i + 1], acc2, 0, 0, 0);
      }
    };
    Frag f0, f1;
    int cur = 0;
    fetch(f0, 0, 0);
    t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nch; ++c) {
      const int nxt = cur == 2 ? 0 : cur + 1;
      fetch(f1, cur, 1);
      mma(f0);
      fetch(f0, nxt, 0);
      mma(f1);
      cur = nxt;
      if (MODE & 1) __syncthreads();
    }
    t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  }
}

template <int MODE, int MPC>
void run(const char* name, int blocks, const float* src, int rows) {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, sizeof(float) * 512 * blocks);
  hipMalloc(&ticks, 8);
  const int nch = 400;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, MPC>), dim3(blocks), dim3(512), 0, 0, src, out, ticks, nch, rows);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, MPC>), dim3(blocks), dim3(512), 0, 0, src, out, ticks, nch, rows);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  const double flops = (double)nch * MPC * 4096.0 * 4 * blocks;
  printf("%-58s mode=%2d mfma/chunk=%2d blocks=%4d: %7.1f ticks/chunk (MFMA alone %4d)  %7.1f TF/s (%5.1f %%)\n", name, MODE, MPC, blocks,
         (double)t / nch, MPC * 64, flops / ms / 1e9, 100 * flops / ms / 1e9 / 157.3);
  hipFree(out); hipFree(ticks);
}

int main() {
  const int rows = 32 * 120000;           // 1.97 GB: 120000 chunks of 16 KB
  float* src; hipMalloc(&src, (size_t)rows * 128 * 4 + (1 << 20)); hipMemset(src, 0, (size_t)rows * 128 * 4 + (1 << 20));
  for (int blocks : {256, 512}) {
    run<0, 16>("MFMA only (fragments in registers, no barrier)", blocks, src, rows);
    run<16, 16>("+ fragments from LDS", blocks, src, rows);
    run<17, 16>("+ barrier per chunk (staging waves idle)", blocks, src, rows);
    run<19, 16>("+ staging waves: 4 ds_write_b128 per lane and chunk", blocks, src, rows);
    run<23, 16>("+ staging waves: 4 buffer_load_dwordx4 (2 chunks deep)", blocks, src, rows);
    run<31, 16>("+ staging waves: 8 VALU ops per chunk", blocks, src, rows);
    run<55, 16>("+ staging waves: 24 VALU ops per chunk", blocks, src, rows);
    run<80, 16>("fragments by ds_read_b128 (k-contiguous LDS), no barrier", blocks, src, rows);
    run<64 + 16 + 128 + 7, 16>("ds_read_b128 [n][36] + barrier + loads + TRANSPOSED b32 stores", blocks, src, rows);
    run<256 + 16 + 128 + 7, 16>("ds_read_b64 [n][34] + barrier + loads + TRANSPOSED b32 stores", blocks, src, rows);
    run<256 + 16 + 7, 16>("ds_read_b64 [n][34] + barrier + loads + b128 stores (wrong layout)", blocks, src, rows);
    run<87, 16>("ds_read_b128 fragments + barrier + stores + loads", blocks, src, rows);
    run<87, 32>("32 MFMAs per chunk: ds_read_b128 fragments + barrier + stores + loads", blocks, src, rows);
    run<23 + 512, 16>("current scheme, STREAMING loads (own rows per workgroup)", blocks, src, rows);
    run<215 + 512, 16>("b128 reads + transposed stores, STREAMING loads", blocks, src, rows);
    run<87 + 512, 32>("32 MFMAs per chunk, b128 reads, STREAMING loads", blocks, src, rows);
    run<23, 32>("32 MFMAs per chunk: LDS frags + barrier + stores + loads", blocks, src, rows);
    run<55, 32>("32 MFMAs per chunk: ... + 24 VALU ops", blocks, src, rows);
  }
  return 0;
}
