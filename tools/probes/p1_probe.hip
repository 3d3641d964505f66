// Probe: phase 1 of wn_fused_kernel<2> in isolation (8 waves, 12 accumulator tiles per wave, weights streamed from global
// one step ahead, A fragments from LDS), with pieces switched off to see what keeps the MFMA pipe at 62 %.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// MODE bit 0: no global loads after the first step; bit 1: no LDS reads after the first step; bit 2: no sched_barrier
// bit 3: s-major -> j-major MFMA order ; bit 4: setprio 1 for waves 4-7
template <int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) p1(const f32x4* __restrict__ W, float* out, int layers) {
  constexpr int NC = 6, KB = 8, T = NC * 2 * 8 / WAVES;  // tiles per wave per step (8 waves: 12)
  __shared__ f32x4 Af[NC * KB * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < NC * KB * 64; i += WAVES * 64) Af[i] = f32x4{1.f + i, 0.5f, 0.25f, 2.f};
  __syncthreads();
  if (MODE & 16) { if (w >= WAVES / 2) __builtin_amdgcn_s_setprio(1); }
  f32x4 acc[T];
  for (int j = 0; j < T; ++j) acc[j] = f32x4{0, 0, 0, 0};
  for (int l = 0; l < layers; ++l) {
    const f32x4* w1 = W + (size_t)l * (8 * KB * NC * 2 * 64) + (size_t)w * KB * (T * 64) + lane;
    f32x4 b0[T], b1[T];
    auto load = [&](f32x4(&d)[T], int u) {
#pragma unroll
      for (int j = 0; j < T; ++j) d[j] = w1[(u * T + j) * 64];
    };
    auto step = [&](int t, const f32x4(&cur)[T]) {
      f32x4 av[T / 2];
#pragma unroll
      for (int j = 0; j < T / 2; ++j) av[j] = Af[((j % NC) * KB + ((MODE & 2) ? 0 : t)) * 64 + lane];
      if (MODE & 8) {
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[j] = mfma4(av[j / 2][s], cur[j][s], acc[j]);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < T; ++j) acc[j] = mfma4(av[j / 2][s], cur[j][s], acc[j]);
      }
    };
    load(b0, 0);
    if (MODE & 32) {
#pragma unroll 1
      for (int it = 0; it < KB / 2; ++it) {
        load(b1, 2 * it + 1);
        step(2 * it, b0);
#pragma unroll
        for (int q = 0; q < T; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (2 * it + 2 < KB) load(b0, 2 * it + 2);
        step(2 * it + 1, b1);
#pragma unroll
        for (int q = 0; q < T; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
#pragma unroll 1
    for (int it = 0; it < KB / 2; ++it) {
      if (!(MODE & 1) || it == 0) load(b1, 2 * it + 1);
      if (!(MODE & 4)) __builtin_amdgcn_sched_barrier(0);
      step(2 * it, b0);
      if (!(MODE & 1) && 2 * it + 2 < KB) load(b0, 2 * it + 2);
      if (!(MODE & 4)) __builtin_amdgcn_sched_barrier(0);
      step(2 * it + 1, b1);
    }
  }
  float r = 0;
  for (int j = 0; j < T; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * WAVES * 64 + tid] = r;
}

template <int MODE, int WAVES>
void run(const f32x4* W, float* out, const char* what) {
  const int layers = 32, blocks = 240;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  p1<MODE, WAVES><<<blocks, WAVES * 64>>>(W, out, layers);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) p1<MODE, WAVES><<<blocks, WAVES * 64>>>(W, out, layers);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  const double mfma_per_simd = 32.0 * 8 * 48 * 8 / 4;  // per layer: 8 steps x 48 x 8 waves / 4 SIMDs
  printf("%-58s waves %2d: %7.2f us per layer, %5.1f ns per MFMA per SIMD (ideal ~13.5-15)\n", what, WAVES, ms * 1e3 / layers, ms * 1e6 / mfma_per_simd);
}

int main() {
  f32x4* W;
  float* out;
  const size_t bytes = (size_t)32 * 8 * 8 * 12 * 64 * 16;
  hipMalloc(&W, bytes);
  hipMemset(W, 0, bytes);
  hipMalloc(&out, 240 * 1024 * 4);
  run<0, 8>(W, out, "full (loads 1 step ahead, LDS A, sched barriers)");
  run<1, 8>(W, out, "no global loads after step 0");
  run<2, 8>(W, out, "LDS reads always the same block");
  run<3, 8>(W, out, "no global loads, same LDS block");
  run<4, 8>(W, out, "full, compiler-scheduled (no sched barriers)");
  run<8, 8>(W, out, "full, accumulator-major MFMA order");
  run<16, 8>(W, out, "full, setprio 1 on waves 4-7");
  run<0, 16>(W, out, "full, 16 waves (6 tiles each)");
  run<4, 16>(W, out, "compiler-scheduled, 16 waves");
  run<1, 16>(W, out, "no global loads after step 0, 16 waves");
  run<0, 4>(W, out, "full, 4 waves (24 tiles each)");
  run<4, 4>(W, out, "compiler-scheduled, 4 waves");
  run<1, 4>(W, out, "no global loads after step 0, 4 waves");
  run<32, 4>(W, out, "1 load per 4 MFMAs (sched_group_barrier), 4 waves");
  run<32, 8>(W, out, "1 load per 4 MFMAs (sched_group_barrier), 8 waves");
  run<32, 16>(W, out, "1 load per 4 MFMAs (sched_group_barrier), 16 waves");
  run<8, 4>(W, out, "accumulator-major, 4 waves");
  return 0;
}
