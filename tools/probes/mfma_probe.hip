// Probe: issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 with 1 or 2 waves per SIMD, operands in registers.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ void __launch_bounds__(1024) k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  float r = 0;
  for (int j = 0; j < NACC; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NACC>
__global__ void __launch_bounds__(1024) k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float r = 0;
  for (int j = 0; j < NACC; ++j)
    for (int q = 0; q < 16; ++q) r += acc[j][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename K>
double run(K kern, int threads, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kern<<<256, threads>>>(out, iters, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<256, threads>>>(out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 2000;
  struct { const char* name; double ms; double mfma_per_simd; double flop; } rows[16];
  int n = 0;
  for (int threads : {256, 512, 1024}) {
    const double waves_per_simd = threads / 256.0;
    double ms = run(k16<12>, threads, iters, out);
    printf("16x16x4 f32, 12 acc, %g wave(s)/SIMD: %.3f ms -> %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", waves_per_simd, ms,
           ms * 1e6 / (iters * 48.0 * waves_per_simd), 256.0 * 4 * waves_per_simd * iters * 48.0 * 2048 / (ms * 1e-3) / 1e12);
    ms = run(k16<4>, threads, iters, out);
    printf("16x16x4 f32,  4 acc, %g wave(s)/SIMD: %.3f ms -> %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", waves_per_simd, ms,
           ms * 1e6 / (iters * 16.0 * waves_per_simd), 256.0 * 4 * waves_per_simd * iters * 16.0 * 2048 / (ms * 1e-3) / 1e12);
    ms = run(k32<4>, threads, iters, out);
    printf("32x32x2 f32,  4 acc, %g wave(s)/SIMD: %.3f ms -> %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", waves_per_simd, ms,
           ms * 1e6 / (iters * 16.0 * waves_per_simd), 256.0 * 4 * waves_per_simd * iters * 16.0 * 4096 / (ms * 1e-3) / 1e12);
  }
  (void)rows; (void)n;
  return 0;
}
