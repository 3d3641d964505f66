// Probe: how fast can every CU stream the SAME weight tensor out of L2 / Infinity Cache into registers?
// (the access pattern of wn_fused_kernel's phase 1: each of W waves of a block reads its own contiguous slice with
//  coalesced 16-byte loads, all blocks read the same bytes at the same time)
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int DEPTH>
__global__ void __launch_bounds__(1024) stream_k(const f32x4* __restrict__ w, int per_wave_vec, int layers, long layer_stride_vec, float* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  for (int l = 0; l < layers; ++l) {
    const f32x4* p = w + l * layer_stride_vec + (long)wv * per_wave_vec + lane;
    for (int i = 0; i < per_wave_vec / 64; i += DEPTH) {
      f32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = p[(i + d) * 64];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}

template <int DEPTH>
void run(const f32x4* w, int waves, int blocks, int layers, size_t layer_bytes, float* out, const char* tag) {
  const int per_wave_vec = (int)(layer_bytes / 16 / waves);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  stream_k<DEPTH><<<blocks, waves * 64>>>(w, per_wave_vec, layers, (long)(layer_bytes / 16), out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) stream_k<DEPTH><<<blocks, waves * 64>>>(w, per_wave_vec, layers, (long)(layer_bytes / 16), out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double bytes_per_block = (double)layer_bytes * layers;
  printf("%-10s blocks %3d waves %2d depth %2d layers %2d x %4zu KB: %.3f ms -> %6.1f GB/s per CU, %5.2f TB/s chip, %.1f us per layer\n", tag, blocks, waves, DEPTH,
         layers, layer_bytes >> 10, ms, bytes_per_block / (ms * 1e-3) / 1e9, bytes_per_block * blocks / (ms * 1e-3) / 1e12, ms * 1e3 / layers);
}

int main() {
  const size_t layer_bytes = 768 << 10;
  const int layers = 32;
  f32x4* w;
  float* out;
  hipMalloc(&w, layer_bytes * layers);
  hipMemset(w, 0, layer_bytes * layers);
  hipMalloc(&out, 4);
  for (int blocks : {30, 240, 256}) {
    run<12>(w, 8, blocks, layers, layer_bytes, out, "32 layers");
    run<24>(w, 8, blocks, layers, layer_bytes, out, "32 layers");
    run<12>(w, 16, blocks, layers, layer_bytes, out, "32 layers");
    run<6>(w, 16, blocks, layers, layer_bytes, out, "32 layers");
    run<12>(w, 4, blocks, layers, layer_bytes, out, "32 layers");
  }
  // the same layer over and over (L2-hot)
  run<12>(w, 8, 240, 1, layer_bytes, out, "1 layer");
  run<24>(w, 8, 240, 1, layer_bytes, out, "1 layer");
  return 0;
}
