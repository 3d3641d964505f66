// Probe / unit harness for conv_gemm16_kernel (csrc/gemm16.hip.h): correctness against a naive device reference (double
// accumulation over the same 16-bit operands) and throughput, (the register-staged 256 x 256 tile it replaces is timed by tools/gemm_bench.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/bin/gemm16_probe tools/probes/gemm16_probe.hip
//   gemm16_probe [n_utt rows_per_utt cin cout k [check]]     (no arguments: the layer shapes of the 16-bit frame path at B = 64)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../stylish_tts_amd/csrc/gemm16.hip.h"

using namespace stts;

static unsigned short h_bf16(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__global__ void ref_kernel(const unsigned short* X, int ldx, const unsigned short* W, int kc, int ntaps, int pad, const int* seg_off, int n_utt,
                           const float* bias, const float* R, int ldr, float alpha, int N, float* Y, int ldy, long rows) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * N) return;
  const int row = (int)(idx / N), n = (int)(idx % N);
  int u = 0;
  while (u + 1 < n_utt && row >= seg_off[u + 1]) ++u;
  const int lo = seg_off[u], hi = seg_off[u + 1];
  double s = 0;
  for (int t = 0; t < ntaps; ++t) {
    const int r = row + t - pad;
    if (r < lo || r >= hi) continue;
    for (int c = 0; c < kc; ++c) {
      const float x = __uint_as_float((unsigned)X[(long)r * ldx + c] << 16), w = __uint_as_float((unsigned)W[((long)n * ntaps + t) * kc + c] << 16);
      s += (double)x * w;
    }
  }
  float v = (float)s + bias[n];
  if (R) v += R[(long)row * ldr + n];
  Y[(long)row * ldy + n] = v * alpha;
}

static int run(int n_utt, int rows_per_utt, int cin, int cout, int k, bool check, bool ragged) {
  const int kc = round_up(cin, 64), npad = round_up(cout, 256), ldy = round_up(cout, 32);
  std::vector<int> h(n_utt + 1, 0);
  for (int i = 0; i < n_utt; ++i) h[i + 1] = h[i] + (ragged ? std::max(1, rows_per_utt - 37 * (i % 5) - (i == 1 ? rows_per_utt / 2 : 0)) : rows_per_utt);
  const long R = h[n_utt];
  unsigned short *X, *W;
  float *Y, *Yref, *B, *Res;
  int* so;
  STTS_HIP(hipMalloc(&X, (R * kc + 64) * 2));
  STTS_HIP(hipMalloc(&W, (size_t)npad * k * kc * 2));
  STTS_HIP(hipMalloc(&Y, R * ldy * 4));
  STTS_HIP(hipMalloc(&Yref, R * ldy * 4));
  STTS_HIP(hipMalloc(&Res, R * ldy * 4));
  STTS_HIP(hipMalloc(&B, npad * 4));
  STTS_HIP(hipMalloc(&so, (n_utt + 1) * 4));
  STTS_HIP(hipMemcpy(so, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  {
    std::vector<unsigned short> t((size_t)std::max<long>(R * kc, (long)npad * k * kc));
    uint32_t s = 12345;
    for (auto& v : t) { s = s * 1664525u + 1013904223u; v = h_bf16(((s >> 8) & 0xFFFF) / 32768.0f - 1.0f); }
    STTS_HIP(hipMemcpy(X, t.data(), R * kc * 2, hipMemcpyHostToDevice));
    for (auto& v : t) { s = s * 1664525u + 1013904223u; v = h_bf16((((s >> 8) & 0xFFFF) / 32768.0f - 1.0f) * 0.05f); }
    // rows >= cout and channels >= cin of the packed weight are zero (as pack_rows leaves them)
    for (int n = 0; n < npad; ++n)
      for (int tt = 0; tt < k; ++tt)
        for (int c = 0; c < kc; ++c)
          if (n >= cout || c >= cin) t[((size_t)n * k + tt) * kc + c] = 0;
    STTS_HIP(hipMemcpy(W, t.data(), (size_t)npad * k * kc * 2, hipMemcpyHostToDevice));
    std::vector<float> b(npad), rr((size_t)R * ldy);
    for (auto& v : b) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    for (auto& v : rr) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    STTS_HIP(hipMemcpy(B, b.data(), npad * 4, hipMemcpyHostToDevice));
    STTS_HIP(hipMemcpy(Res, rr.data(), rr.size() * 4, hipMemcpyHostToDevice));
  }
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.seg_off = so; a.seg_host = h.data(); a.n_utt = n_utt; a.rows_total = (int)R; a.zeros = zero_page();
  a.nseg = 1; a.prec = PREC_BF16; a.wrows = cout;
  a.seg[0].X = reinterpret_cast<const float*>(X); a.seg[0].W = reinterpret_cast<const float*>(W); a.seg[0].W16 = W; a.seg[0].ldx = kc; a.seg[0].kc = kc;
  a.seg[0].ntaps = k; a.seg[0].dil = 1; a.seg[0].pad = (k - 1) / 2; a.seg[0].kreal = cin;
  a.x16 = 1;
  a.N = cout; a.bias = B; a.Y = Y; a.ldy = ldy; a.R = Res; a.ldr = ldy; a.alpha = 0.5f;
  hipStream_t st = nullptr;
  int rc = 0;
  if (check) {
    STTS_HIP(hipMemset(Y, 0xff, R * ldy * 4));
    STTS_TRY(launch_conv_gemm16(st, a, npad, n_utt));
    const long work = R * cout;
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, X, kc, W, kc, k, (k - 1) / 2, so, n_utt, B, Res, ldy, 0.5f, cout, Yref, ldy, R);
    STTS_HIP(hipDeviceSynchronize());
    std::vector<float> y((size_t)R * ldy), yr((size_t)R * ldy);
    STTS_HIP(hipMemcpy(y.data(), Y, y.size() * 4, hipMemcpyDeviceToHost));
    STTS_HIP(hipMemcpy(yr.data(), Yref, yr.size() * 4, hipMemcpyDeviceToHost));
    double err = 0, mx = 0;
    long bad = 0;
    for (long r = 0; r < R; ++r)
      for (int n = 0; n < cout; ++n) {
        const double d = fabs((double)y[r * ldy + n] - yr[r * ldy + n]);
        if (!(d == d) || d > 1e-3) { if (bad < 5) fprintf(stderr, "  mismatch row %ld col %d: %g vs %g\n", r, n, y[r * ldy + n], yr[r * ldy + n]); ++bad; }
        err = std::max(err, d);
        mx = std::max(mx, fabs((double)yr[r * ldy + n]));
      }
    printf("check n_utt=%d rows=%ld cin=%d cout=%d k=%d%s: max abs err %.3e (max |ref| %.3f), %ld bad\n", n_utt, R, cin, cout, k, ragged ? " ragged" : "", err, mx, bad);
    rc = bad != 0;
  } else {
    hipEvent_t e0, e1;
    STTS_HIP(hipEventCreate(&e0));
    STTS_HIP(hipEventCreate(&e1));
    const double flops = 2.0 * R * cout * (double)cin * k;
    for (int variant = 0; variant < 1; ++variant) {  // (the register-staged 256 x 256 tile this replaces: tools/gemm_bench.py TILE=14 TUNE=1152)
      auto go = [&]() { return launch_conv_gemm16(st, a, npad, n_utt); };
      for (int i = 0; i < 3; ++i) STTS_TRY(go());
      STTS_HIP(hipEventRecord(e0, st));
      const int iters = 20;
      for (int i = 0; i < iters; ++i) STTS_TRY(go());
      STTS_HIP(hipEventRecord(e1, st));
      STTS_HIP(hipEventSynchronize(e1));
      float ms = 0;
      STTS_HIP(hipEventElapsedTime(&ms, e0, e1));
      printf("%-18s n_utt=%d rows=%ld cin=%d cout=%d k=%d: %8.1f us  %7.1f TFLOP/s\n", variant == 0 ? "conv_gemm16" : "conv_gemm_f32<14>", n_utt, R, cin, cout, k, 1e3 * ms / iters,
             flops / (ms / iters * 1e-3) * 1e-12);
    }
  }
  (void)hipFree(X); (void)hipFree(W); (void)hipFree(Y); (void)hipFree(Yref); (void)hipFree(Res); (void)hipFree(B); (void)hipFree(so);
  return rc;
}

int main(int argc, char** argv) {
  if (argc >= 6) {
    const int rc = run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argc > 6 && atoi(argv[6]) != 0, argc > 7 && atoi(argv[7]) != 0);
    if (rc) fprintf(stderr, "error: %s\n", last_error().c_str());
    return rc;
  }
  int bad = 0;
  // correctness: small and ragged shapes, k = 1 / 3 / 7, one and several K tiles
  bad += run(1, 256, 64, 256, 1, true, false);
  bad += run(2, 300, 128, 256, 3, true, false);
  bad += run(5, 700, 192, 512, 7, true, true);
  bad += run(3, 960, 512, 1000, 1, true, true);
  if (bad) { fprintf(stderr, "FAILED: %s\n", last_error().c_str()); return 1; }
  // throughput: the layers of the 16-bit frame path at B = 64 x 3 s (61 440 rows)
  const int shapes[][3] = {{768, 1024, 7}, {1088, 256, 7}, {512, 1536, 1}, {1536, 512, 1}, {512, 512, 3}, {640, 512, 3}, {512, 512, 1}};
  for (auto& s : shapes) bad += run(64, 960, s[0], s[1], s[2], false, false);
  return bad;
}
