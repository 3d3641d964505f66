// Harmonic source, STFT and iSTFT of the vocoder.
//   generate_pcph      models/generator.py:247-315   (fp64 phase accumulator, <=16 harmonics)
//   TorchSTFT.transform models/generator.py:32-44    torch.stft(n_fft 2048, hop 75, periodic Hann 1200 centred, reflect pad)
//   TorchSTFT.inverse   models/generator.py:46-56    torch.istft(...) : irfft, window, overlap-add, /sum(w^2), trim
// One workgroup per STFT frame: the transform runs entirely in LDS (Stockham radix-2, twiddles from a table
// computed in fp64 on the host); forward in fp64 (see stft_kernel), inverse in fp32.  These kernels are HBM/LDS-bound and small next to the
// contractions; they are written for exactness first (fp64 phase, full-precision exp/sin/cos).
#pragma once
#include "common.h"

namespace stts {

constexpr int kNfft = 2048, kHop = 75, kWin = 1200, kBins = 1025, kWinLo = (kNfft - kWin) / 2;  // 424
constexpr float kSampleRate = 24000.0f;

// ---------------------------------------------------------------------------------------------
// generate_pcph, step 1: per-utterance statistics + exclusive fp64 prefix of the per-frame phase increment
//   radious = f0.double()/sr ; cumsum over samples (generator.py:304-308); f0 is constant within a frame, so
//   cumsum at sample (j, i) = prefix[j] + (i+1) * f0[j]/sr  (+ the shared random initial phase).
// stats[u] = { min f0 over frames with f0 > 20 (inf if none), any frame voiced (f0 > 10) }
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pcph_prep_kernel(const float* __restrict__ f0, const int* __restrict__ seg_off,
                                                        double* __restrict__ prefix, float* __restrict__ stats) {
  __shared__ double part[256];
  __shared__ float mn[256];
  __shared__ int anyv[256];
  const int u = blockIdx.x;
  const int lo = seg_off[u], n = seg_off[u + 1] - lo;
  const int per = (n + 255) / 256;
  const int a = threadIdx.x * per, b = min(n, a + per);
  double s = 0.0;
  float m = INFINITY;
  int av = 0;
  for (int j = a; j < b; ++j) {
    const float f = f0[lo + j];
    s += (double)kHop * ((double)f / (double)kSampleRate);
    if (f > 20.0f) m = fminf(m, f);
    if (f > 10.0f) av = 1;
  }
  part[threadIdx.x] = s;
  mn[threadIdx.x] = m;
  anyv[threadIdx.x] = av;
  __syncthreads();
  if (threadIdx.x == 0) {
    double run = 0.0;
    float mm = INFINITY;
    int aa = 0;
    for (int i = 0; i < 256; ++i) {
      const double t = part[i];
      part[i] = run;
      run += t;
      mm = fminf(mm, mn[i]);
      aa |= anyv[i];
    }
    stats[2 * u] = mm;
    stats[2 * u + 1] = (float)aa;
  }
  __syncthreads();
  double run = part[threadIdx.x];
  for (int j = a; j < b; ++j) {
    prefix[lo + j] = run;
    run += (double)kHop * ((double)f0[lo + j] / (double)kSampleRate);
  }
}

// step 2: one thread per output sample.
//   batch_scope != 0: harmonic count K from the minimum over ALL utterances of the call (what the reference does
//   for a batched call, generator.py:285-287); 0: per utterance (= the reference called with B=1 per utterance).
//   err[0] is set when some frame is voiced but no f0 exceeds 20 Hz (the reference raises there, generator.py:285).
__global__ void __launch_bounds__(256) pcph_kernel(const float* __restrict__ f0, const int* __restrict__ seg_off, int n_utt,
                                                   const double* __restrict__ prefix, const float* __restrict__ stats,
                                                   const float* __restrict__ noise, const float* __restrict__ init_phase, int batch_scope,
                                                   float* __restrict__ out, int* __restrict__ err) {
  const int u = blockIdx.y;
  const int lo = seg_off[u], nfr = seg_off[u + 1] - lo;
  // the STFT reflects kNfft / 2 samples at both ends: shorter utterances cannot be padded (torch.stft raises; the host checks this
  // when it knows the lengths, here for lengths that only exist on the device)
  if (blockIdx.x == 0 && threadIdx.x == 0 && (long)nfr * kHop <= kNfft / 2) atomicOr(err, 4);  // (bit flags: several conditions can be pending at once, stts_check_status reports each)
  float mnf = stats[2 * u];
  int anyv = stats[2 * u + 1] > 0.5f;
  if (batch_scope) {  // minimum / any over the call's utterances: lanes take utterances, one wave reduction (every thread looping over all of them was half of this kernel at B = 64)
    for (int v = threadIdx.x & 63; v < n_utt; v += 64) {
      mnf = fminf(mnf, stats[2 * v]);
      anyv |= stats[2 * v + 1] > 0.5f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mnf = fminf(mnf, __shfl_xor(mnf, o, 64));
      anyv |= __shfl_xor(anyv, o, 64);
    }
  }
  int K = 0;
  if (anyv) {
    if (isinf(mnf)) {
      if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(err, 1);
      K = 16;
    } else {
      K = min(16, (int)(12000.0 / (double)mnf));
    }
  }
  const double ph0 = (double)init_phase[0];
  const long nsamp = (long)nfr * kHop, base = (long)lo * kHop;
  for (long sidx = (long)blockIdx.x * 256 + threadIdx.x; sidx < nsamp; sidx += (long)gridDim.x * 256) {
    const int j = (int)(sidx / kHop), i = (int)(sidx % kHop);
    const float f = f0[lo + j];
    float val = 0.01f * noise[base + sidx];
    if (f > 10.0f && K > 0) {
      const double rad = ph0 + prefix[lo + j] + (double)(i + 1) * ((double)f / (double)kSampleRate);
      const float nh = (kSampleRate * 0.5f) / f;
      const float amp = 0.1f * sqrtf(2.0f / nh);
      // sin(2 pi rad k), k = 1..K (generator.py:309-310 evaluates each in fp64 and rounds to fp32): 1-periodic in rad, so with
      // theta = 2 pi frac(rad) the harmonics are sin(k theta) - one fp64 sincos and the three-term recurrence
      // sin((k+1) theta) = 2 cos(theta) sin(k theta) - sin((k-1) theta), whose error (<= k^2 ulp64 ~ 3e-14 at k = 16) is far below
      // both the fp32 rounding that follows and the reference's own rounding of the product 2 pi rad k (~3e-11 late in a 10-s utterance)
      double s1, c1;
      sincospi(2.0 * (rad - floor(rad)), &s1, &c1);
      const double tc = 2.0 * c1;
      double sp = 0.0, sk = s1;
      float acc = 0.f;
      for (int k = 1; k <= K; ++k) {
        if (f * (float)k <= kSampleRate * 0.5f) acc += (float)sk;
        const double nx = tc * sk - sp;
        sp = sk;
        sk = nx;
      }
      val += amp * acc;
    }
    out[base + sidx] = val;
  }
}

// ---------------------------------------------------------------------------------------------
// N-point complex FFT in LDS (Stockham autosort, radix 4), 256 threads.  tw[m * tw_scale] = exp(-2*pi*i*m/N)
// for m < N/2 (forward); the inverse conjugates on the fly.  Input in `a`; returns the buffer holding the result.
// ---------------------------------------------------------------------------------------------
template <typename T2, int N, bool INVERSE>
__device__ __forceinline__ T2* fft_lds(T2* a, T2* b, const T2* tw, int tw_scale) {
  // radix-4 Stockham autosort: N = 4^5 = 1024 points, one butterfly per thread per stage, 5 stages / 5 barriers
  static_assert(N == 1024, "fft_lds: 256 threads x radix 4 x 5 stages");
  using T = decltype(T2().x);
  const int j = threadIdx.x;  // 0 .. N/4-1
  T2* in = a;
  T2* out = b;
  auto cmul = [](const T2 x, const T2 w) {
    T2 r;
    r.x = x.x * w.x - x.y * w.y;
    r.y = x.x * w.y + x.y * w.x;
    return r;
  };
  auto twid = [&](int idx) {  // exp(-+ 2 pi i idx / N); the table covers half a turn (idx * tw_scale < table length)
    const int half = N / 2;
    T2 w = tw[(idx >= half ? idx - half : idx) * tw_scale];
    if (idx >= half) {
      w.x = -w.x;
      w.y = -w.y;
    }
    if (INVERSE) w.y = -w.y;
    return w;
  };
#pragma unroll 1
  for (int Ns = 1; Ns < N; Ns <<= 2) {
    const int k = j & (Ns - 1);
    const int base = k * (N / (4 * Ns));  // angle unit 2 pi / N
    T2 v0 = in[j], v1 = in[j + N / 4], v2 = in[j + N / 2], v3 = in[j + 3 * N / 4];
    if (Ns > 1) {
      v1 = cmul(v1, twid(base));
      v2 = cmul(v2, twid(2 * base));
      v3 = cmul(v3, twid(3 * base));
    }
    T2 a0, a1, a2, a3;
    a0.x = v0.x + v2.x; a0.y = v0.y + v2.y;
    a1.x = v0.x - v2.x; a1.y = v0.y - v2.y;
    a2.x = v1.x + v3.x; a2.y = v1.y + v3.y;
    const T dx = v1.x - v3.x, dy = v1.y - v3.y;
    if (INVERSE) {  // (v1 - v3) * (+i)
      a3.x = -dy; a3.y = dx;
    } else {        // (v1 - v3) * (-i)
      a3.x = dy; a3.y = -dx;
    }
    const int j0 = ((j - k) << 2) + k;
    T2 o;
    o.x = a0.x + a2.x; o.y = a0.y + a2.y; out[j0] = o;
    o.x = a1.x + a3.x; o.y = a1.y + a3.y; out[j0 + Ns] = o;
    o.x = a0.x - a2.x; o.y = a0.y - a2.y; out[j0 + 2 * Ns] = o;
    o.x = a1.x - a3.x; o.y = a1.y - a3.y; out[j0 + 3 * Ns] = o;
    __syncthreads();
    T2* tmp = in;
    in = out;
    out = tmp;
  }
  return in;
}

// ---------------------------------------------------------------------------------------------
// 1024-point complex FFT by ONE WAVE (Stockham autosort, radices 16 x 16 x 4): a lane holds 16 points in registers, the
// two radix-16 stages are 4 x 4 butterflies in registers, stages exchange through a per-wave LDS buffer — in place,
// because a wave issues all its reads before its writes and LDS serves one wave's requests in order — so the
// transform needs no workgroup barrier and a block is simply four independent frames.  The buffer is padded one
// element per 16 (`fphys`): with it every access pattern of the three stages is bank-conflict free for 16-byte
// elements.  Twiddles: tw[t] = exp(-2 pi i t / 2048), t < 1024 (the model's table); the inverse conjugates.
//   stage 1: v[r] = x[lane + 64 r] (handed in registers)       -> buf[16 lane + q]
//   stage 2: v[r] = buf[lane + 64 r] * W256^(r k), k = lane&15 -> buf[256 (lane>>4) + k + 16 q]
//   stage 3: x[r] = buf[j + 256 r] * W1024^(r j), j = lane + 64 t, t < 4 -> buf[j + 256 q]   (natural order)
// ---------------------------------------------------------------------------------------------
constexpr int kFftPts = 1024, kFftBuf = kFftPts + kFftPts / 16 + 4;  // 1092 elements per wave
__device__ __forceinline__ int fphys(int i) { return i + (i >> 4); }

template <typename T2>
__device__ __forceinline__ T2 cmul2(const T2 x, const T2 w) {
  T2 r;
  r.x = x.x * w.x - x.y * w.y;
  r.y = x.x * w.y + x.y * w.x;
  return r;
}
template <typename T2, bool INV>
__device__ __forceinline__ void dft4(T2& x0, T2& x1, T2& x2, T2& x3) {
  using T = decltype(T2().x);
  T2 a0, a1, a2, a3;
  a0.x = x0.x + x2.x; a0.y = x0.y + x2.y;
  a1.x = x0.x - x2.x; a1.y = x0.y - x2.y;
  a2.x = x1.x + x3.x; a2.y = x1.y + x3.y;
  const T dx = x1.x - x3.x, dy = x1.y - x3.y;
  if (INV) { a3.x = -dy; a3.y = dx; } else { a3.x = dy; a3.y = -dx; }
  x0.x = a0.x + a2.x; x0.y = a0.y + a2.y;
  x1.x = a1.x + a3.x; x1.y = a1.y + a3.y;
  x2.x = a0.x - a2.x; x2.y = a0.y - a2.y;
  x3.x = a1.x - a3.x; x3.y = a1.y - a3.y;
}
// DFT-16 in registers; on return the output with index q = q1 + 4 q2 sits in v[4 q1 + q2].
template <typename T2, bool INV>
__device__ __forceinline__ void dft16(T2 (&v)[16]) {
  using T = decltype(T2().x);
  constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
#pragma unroll
  for (int c = 0; c < 4; ++c) dft4<T2, INV>(v[c], v[c + 4], v[c + 8], v[c + 12]);
  auto tw = [&](T2& x, double cr, double ci) {  // x *= cr -+ i ci
    T2 w;
    w.x = (T)cr;
    w.y = INV ? (T)ci : (T)-ci;
    x = cmul2(x, w);
  };
  tw(v[1 + 4], c1, s1);    // W16^1
  tw(v[1 + 8], h, h);      // W16^2
  tw(v[1 + 12], s1, c1);   // W16^3
  tw(v[2 + 4], h, h);      // W16^2
  tw(v[2 + 8], 0.0, 1.0);  // W16^4
  tw(v[2 + 12], -h, h);    // W16^6
  tw(v[3 + 4], s1, c1);    // W16^3
  tw(v[3 + 8], -h, h);     // W16^6
  tw(v[3 + 12], -c1, -s1); // W16^9
#pragma unroll
  for (int q = 0; q < 4; ++q) dft4<T2, INV>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
// cos(pi i / 16), i = 0..16 (sin(pi i / 16) = cos of index |8 - i|)
__device__ constexpr double kCosPi16[17] = {1.0, 0.98078528040323043058, 0.92387953251128673848, 0.83146961230254523567,
                                            0.70710678118654752440, 0.55557023301960228867, 0.38268343236508983729, 0.19509032201612833135,
                                            0.0, -0.19509032201612833135, -0.38268343236508983729, -0.55557023301960228867,
                                            -0.70710678118654752440, -0.83146961230254523567, -0.92387953251128673848, -0.98078528040323043058,
                                            -1.0};
__device__ __forceinline__ double2 unit32(int i) {  // exp(-2 pi i * i / 32), 0 <= i < 16
  return make_double2(kCosPi16[i], -kCosPi16[i <= 8 ? 8 - i : i - 8]);
}
template <typename T2>
__device__ __forceinline__ T2 to_t2(const double2 w, bool conj) {
  using T = decltype(T2().x);
  T2 r;
  r.x = (T)w.x;
  r.y = (T)(conj ? -w.y : w.y);
  return r;
}
// The twiddles a lane needs, all derived in fp64 from THREE table entries fetched when the kernel starts (a wave shares its
// SIMD with one to three others, so every dependent global round trip inside the transform shows): exp(-2 pi i t / 2048) for
//   t = 8 (lane & 15)  -> W256^(r k), r < 16, by products       (stage 2)
//   t = 2 lane         -> W1024^(r (lane + 64 t)), r < 4, t < 4 (stage 3; W1024^64 = W16)
//   t = lane           -> exp(-2 pi i (lane + 64 i) / 2048), i < 16 (real <-> half-size complex split; 64/2048 = 1/32)
struct FftTw {
  double2 w256k, w1024l, w2048l;
};
__device__ __forceinline__ FftTw fft_load_tw(const double2* __restrict__ tw, int lane) {
  FftTw t;
  t.w256k = tw[8 * (lane & 15)];
  t.w1024l = tw[2 * lane];
  t.w2048l = tw[lane];
  return t;
}
// (workgroup scope = an s_waitcnt lgkmcnt(0) on both sides of every exchange; a wave's LDS operations complete in order, so wavefront scope - no
//  instruction at all - is what the memory model asks for, and it measured the same; the wait is kept as the form that does not lean on that)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <typename T2, bool INV>
__device__ __forceinline__ void fft1024_wave(T2 (&v)[16], T2* buf, const FftTw& tw, int lane) {
  dft16<T2, INV>(v);
#pragma unroll
  for (int q = 0; q < 16; ++q) buf[fphys(16 * lane + q)] = v[4 * (q & 3) + (q >> 2)];
  wave_lds_fence();
  const int k = lane & 15, a = lane >> 4;
  double2 p[16];
  p[1] = tw.w256k;
#pragma unroll
  for (int r = 2; r < 16; ++r) p[r] = cmul2(p[r >> 1], p[r - (r >> 1)]);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = buf[fphys(lane + 64 * r)];
#pragma unroll
  for (int r = 1; r < 16; ++r) v[r] = cmul2(v[r], to_t2<T2>(p[r], INV));
  dft16<T2, INV>(v);
  wave_lds_fence();
#pragma unroll
  for (int q = 0; q < 16; ++q) buf[fphys(256 * a + k + 16 * q)] = v[4 * (q & 3) + (q >> 2)];
  wave_lds_fence();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = lane + 64 * t;
    const double2 w1 = cmul2(tw.w1024l, unit32(2 * t)), w2 = cmul2(w1, w1), w3 = cmul2(w2, w1);  // W1024^(r j)
    T2 x0 = buf[fphys(j)], x1 = buf[fphys(j + 256)], x2 = buf[fphys(j + 512)], x3 = buf[fphys(j + 768)];
    x1 = cmul2(x1, to_t2<T2>(w1, INV));
    x2 = cmul2(x2, to_t2<T2>(w2, INV));
    x3 = cmul2(x3, to_t2<T2>(w3, INV));
    dft4<T2, INV>(x0, x1, x2, x3);
    v[4 * t] = x0; v[4 * t + 1] = x1; v[4 * t + 2] = x2; v[4 * t + 3] = x3;
  }
  wave_lds_fence();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = lane + 64 * t;
#pragma unroll
    for (int q = 0; q < 4; ++q) buf[fphys(j + 256 * q)] = v[4 * t + q];
  }
  wave_lds_fence();
}

// atan2 for the phase channel: octant reduction to t = min/max in [0, 1], atan(t) = t P(t^2) (degree-8 minimax fit, 6e-9;
// 1.2e-7 evaluated in fp32 = the rounding of the result), quadrant fix-up.  atan2(0, 0) = 0 and the sign conventions at
// the cut (y = +-0, x < 0 -> +-pi) follow torch.atan2.
__device__ __forceinline__ float atan2_poly(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float t = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
  const float q = t * t;
  float r = 2.456697810e-03f;
  r = r * q + -1.440124539e-02f;
  r = r * q + 3.978102917e-02f;
  r = r * q + -7.234839583e-02f;
  r = r * q + 1.049893683e-01f;
  r = r * q + -1.416122648e-01f;
  r = r * q + 1.998590634e-01f;
  r = r * q + -3.333259700e-01f;
  r = r * q + 9.999998864e-01f;
  r *= t;
  if (ay > ax) r = 1.57079632679489662f - r;
  if (x < 0.f || (x == 0.f && __builtin_signbitf(x))) r = 3.14159265358979324f - r;
  return __builtin_copysignf(r, y);
}

// Forward STFT of the harmonic prior + magnitude / atan2 phase (generator.py:406-410).
// One wave per frame, four frames per block: grid (ceil(max frames per utterance / 4), n_utt); frame f of utterance u
// covers samples [75f - 600, 75f + 600) of the reflect-padded signal.  Outputs time-major [rows, ld] with bins
// 0..1024; pad columns are zeroed.
//
// The transform runs in FP64 (real 2048-point FFT as one 1024-point complex FFT + split): atan2 is discontinuous
// at the +-pi cut and meaningless at ~0 magnitude, so the SIGN of a rounding-level real/imaginary part decides a
// 2*pi jump that the next conv sees linearly.  FP64 makes those signs those of the exact transform of the fp32
// windowed samples.
// out16: 0 = fp32 rows; 1 (bf16) / 2 (fp16) = spec / phase are 16-bit row buffers (ld in elements): the operands of the prior convs
// in the 16-bit modes, written here instead of being rounded in a separate pass.
constexpr int kFftWaves = 4;
__global__ void __launch_bounds__(64 * kFftWaves) stft_kernel(const float* __restrict__ sig, const int* __restrict__ seg_off,
                                                              const float* __restrict__ hann, const double2* __restrict__ twiddle,
                                                              float* __restrict__ spec, float* __restrict__ phase, int ld, int out16) {
  constexpr int H = kNfft / 2;  // 1024
  __shared__ double2 bufs[kFftWaves][kFftBuf];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int u = blockIdx.y, f = blockIdx.x * kFftWaves + wv;
  const int lo = seg_off[u], nfr = seg_off[u + 1] - lo;
  if (f >= nfr) return;  // whole waves leave: nothing below synchronises across waves
  double2* Z = bufs[wv];
  const FftTw tw = fft_load_tw(twiddle, lane);
  const long L = (long)nfr * kHop;
  const float* x = sig + (long)lo * kHop;
  double2 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = lane + 64 * r;  // complex point n = real samples 2n, 2n+1 of the 2048-sample frame
    double e[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int p = 2 * n + c;
      double val = 0.0;
      if (p >= kWinLo && p < kWinLo + kWin) {
        long m = (long)f * kHop - kNfft / 2 + p;
        if (m < 0) m = -m;
        if (m >= L) m = 2 * (L - 1) - m;
        val = (double)(x[m] * hann[p - kWinLo]);  // the product is formed in fp32 like torch.stft's windowing
      }
      e[c] = val;
    }
    v[r] = make_double2(e[0], e[1]);
  }
  fft1024_wave<double2, false>(v, Z, tw, lane);
  // Split + magnitude / phase.  All of a lane's operands are fetched before any is used (a wave has one other wave to hide
  // behind on its SIMD: 17.5 KB of LDS per frame), so the 17 column groups are unrolled and loaded as a batch.
  const long row = lo + f;
  constexpr int G = (kBins + 63) / 64;  // 17 groups of 64 columns; ld <= 64 G (checked by the launcher)
  double2 za[G - 1], zb[G - 1];
#pragma unroll
  for (int i = 0; i < G - 1; ++i) {
    const int k = lane + 64 * i;  // 0 .. 1023
    za[i] = Z[fphys(k)];
    zb[i] = Z[fphys(H - k)];  // k = 0 reads the pad slot fphys(1024): unused below
  }
  const double2 z0 = Z[0];
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int k = lane + 64 * i;
    if (k >= ld) break;
    float m = 0.f, p = 0.f;
    if (k < kBins) {
      double re, im;
      if (k == 0 || k == H) {
        re = k == 0 ? z0.x + z0.y : z0.x - z0.y;
        im = 0.0;
      } else {
        const double2 a = za[i < G - 1 ? i : 0], b = zb[i < G - 1 ? i : 0];
        const double2 w = cmul2(tw.w2048l, unit32(i < G - 1 ? i : 0));  // exp(-2 pi i k / 2048)
        const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);    // even part  (Z[k] + conj(Z[H-k]))/2
        const double orr = 0.5 * (a.y + b.y), oi = -0.5 * (a.x - b.x);  // odd part   (Z[k] - conj(Z[H-k]))/(2i)
        re = er + orr * w.x - oi * w.y;
        im = ei + orr * w.y + oi * w.x;
      }
      const float fr = (float)re, fi = (float)im;
      m = __builtin_amdgcn_sqrtf(fr * fr + fi * fi);  // 1 ulp
      // generator.py:409 takes atan2(im / (m + 1e-9), re / (m + 1e-9)): a common positive scale, which the ratio inside
      // atan2 cancels; (0, 0) -> 0 either way
      p = atan2_poly(fi, fr);
    }
    if (out16 == 0) {
      spec[row * ld + k] = m;
      phase[row * ld + k] = p;
    } else if (out16 == 1) {
      reinterpret_cast<__bf16*>(spec)[row * ld + k] = (__bf16)m;
      reinterpret_cast<__bf16*>(phase)[row * ld + k] = (__bf16)p;
    } else {
      reinterpret_cast<_Float16*>(spec)[row * ld + k] = (_Float16)m;
      reinterpret_cast<_Float16*>(phase)[row * ld + k] = (_Float16)p;
    }
  }
}

// sin / cos of the phase channel (|p| <= 1 in the model: the phase head ends in sin(), generator.py:429): quadrant reduction
// r = p - k pi/2 (two-term Cody-Waite, exact for the |k| <= 1 the model produces, ~1e-7 |k| beyond), then the Taylor
// polynomials on [-pi/4, pi/4] (truncation 2e-9 / 2.4e-8) — no slow path, so seventeen unrolled copies stay small.
__device__ __forceinline__ void sincos_unit(float p, float& sn, float& cs) {
  const float kf = rintf(p * 0.636619772367581343f);
  float r = fmaf(-kf, 1.5707963705062866f, p);
  r = fmaf(-kf, -4.371139000186243e-8f, r);
  const float q = r * r;
  const float s = r + r * q * (-1.6666667e-1f + q * (8.3333333e-3f + q * (-1.9841270e-4f + q * 2.7557319e-6f)));
  const float c = 1.0f + q * (-0.5f + q * (4.1666667e-2f + q * (-1.3888889e-3f + q * 2.4801587e-5f)));
  const int k = (int)kf;
  const float a = (k & 1) ? c : s, b = (k & 1) ? s : c;
  sn = (k & 2) ? -a : a;
  cs = ((k + 1) & 2) ? -b : b;
}

// Inverse: frame f in [0, T4] of utterance u (T4+1 frames; the last repeats row T4-1: F.pad replicate,
// generator.py:425-426).  X = exp(logamp) * (cos(phase) + i sin(phase)) (generator.py:428-430), Hermitian
// extension (imag of DC / Nyquist ignored like a C2R transform), inverse FFT, 1/N, window.
// yw rows: utterance u starts at seg_off[u] + u.  One wave per frame, grid (ceil((max frames + 1) / 4), n_utt).
// The Hermitian spectrum makes the output real, so the 2048-point inverse runs as ONE 1024-point complex transform:
//   E[k] = (X[k] + conj(X[H-k]))/2 ,  O[k] = (X[k] - conj(X[H-k]))/2 * e^{+2 pi i k/N} ,  Z[k] = E[k] + i O[k]  (H = N/2)
//   z = IFFT_H(Z) / H ;  x[2n] = Re z[n], x[2n+1] = Im z[n]
// NO PACKED-FP32 INSTRUCTIONS in this kernel (round 4).  Compiled with them (456 v_pk_add / v_pk_mul / v_pk_fma_f32 in the butterflies) the kernel
// returned DIFFERENT frames for the same rows whenever blocks of the split-fp32 contractions (bf16 MFMA + their own v_pk_* split arithmetic, 96+ KB of
// LDS: an istft block fits beside one on a CU) ran on other streams at the same time: two back-to-back launches on identical, host-verified inputs
// differed in 16-point groups of single frames (values off by 1e-3 .. 1e-1, never garbage), with the stream idle on both sides of each launch and with
// s_waitcnt lgkmcnt(0) around every LDS exchange; alone on the GPU, or next to the f32-MFMA contractions, never.  Without packed ops: 0 differences in
// 1 400 concurrent launches (tools/debug/stage_race.py, tests/test_hip_concurrency.py).  The forward transform (fp64, no packed fp32) never showed it.
// Cost: none measurable (33 us per launch at B = 8 either way).  DESIGN.md section 5d.
#ifndef STTS_ISTFT_PACKED_FP32
__attribute__((target("no-packed-fp32-ops")))
#endif
__global__ void __launch_bounds__(64 * kFftWaves) istft_frames_kernel(const float* __restrict__ logamp, const float* __restrict__ phase, int ld,
                                                                      const int* __restrict__ seg_off, const float* __restrict__ hann,
                                                                      const double2* __restrict__ twiddle, float* __restrict__ yw) {
  constexpr int H = kNfft / 2;
  __shared__ float2 bufs[kFftWaves][kFftBuf];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int u = blockIdx.y, f = blockIdx.x * kFftWaves + wv;
  const int lo = seg_off[u], nfr = seg_off[u + 1] - lo;
  if (f > nfr) return;
  float2* Z = bufs[wv];
  const FftTw tw = fft_load_tw(twiddle, lane);
  const long row = lo + min(f, nfr - 1);
  const float* la = logamp + row * ld;
  const float* ph = phase + row * ld;
  constexpr int G = (kBins + 63) / 64;  // 17 column groups, fetched as one batch (see stft_kernel)
  float lav[G], phv[G];
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int k = lane + 64 * i;
    lav[i] = k < kBins ? la[k] : 0.f;
    phv[i] = k < kBins ? ph[k] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int k = lane + 64 * i;
    const float a = expf(lav[i]);
    float sn, cs;
    sincos_unit(phv[i], sn, cs);
    float re = a * cs, im = a * sn;
    if (k == 0 || k == H) im = 0.f;  // a C2R transform ignores them (torch.istft / pocketfft)
    if (k < kBins) Z[fphys(k)] = make_float2(re, im);
  }
  wave_lds_fence();
  float2 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = lane + 64 * r;
    const float2 x = Z[fphys(k)], y = Z[fphys(H - k)];
    const float er = 0.5f * (x.x + y.x), ei = 0.5f * (x.y - y.y);    // E = (X[k] + conj(X[H-k]))/2
    const float dr = 0.5f * (x.x - y.x), di = 0.5f * (x.y + y.y);    // D = (X[k] - conj(X[H-k]))/2
    const float2 w = to_t2<float2>(cmul2(tw.w2048l, unit32(r)), false);  // exp(-2 pi i k / 2048); conj(w) = e^{+2 pi i k/N}
    const float orr = dr * w.x + di * w.y, oi = di * w.x - dr * w.y; // O = D * conj(w)
    v[r] = make_float2(er - oi, ei + orr);                           // Z = E + i O
  }
  wave_lds_fence();
  fft1024_wave<float2, true>(v, Z, tw, lane);
  float2* o = reinterpret_cast<float2*>(yw + (long)(lo + u + f) * kWin);
  const float2* hw = reinterpret_cast<const float2*>(hann);
#pragma unroll
  for (int t = 0; t < (kWin / 2 + 63) / 64; ++t) {  // samples 2i, 2i+1 of the window = point kWinLo/2 + i
    const int i = lane + 64 * t;
    if (i < kWin / 2) {
      const float2 z = Z[fphys(kWinLo / 2 + i)], w = hw[i];
      o[i] = make_float2(z.x * (1.0f / H) * w.x, z.y * (1.0f / H) * w.y);
    }
  }
}

// Overlap-add + window-envelope normalisation + centre trim + tanh (generator.py:432-433; torch.istft).
// out sample s of utterance u (0 <= s < 75*T4) sits at t = s + 1024 of the untrimmed signal; frame f contributes
// yw[f][t - 75 f - 424] while that index is in [0, 1200).
__global__ void __launch_bounds__(256) istft_ola_kernel(const float* __restrict__ yw, const int* __restrict__ seg_off, const float* __restrict__ hann,
                                                        float* __restrict__ audio) {
  const int u = blockIdx.y;
  const int lo = seg_off[u], nfr = seg_off[u + 1] - lo;
  const long nsamp = (long)nfr * kHop;
  const float* y = yw + (long)(lo + u) * kWin;
  for (long s = (long)blockIdx.x * 256 + threadIdx.x; s < nsamp; s += (long)gridDim.x * 256) {
    const int t = (int)s + kNfft / 2 - kWinLo;  // offset inside frame 0's window
    int f_hi = t / kHop;
    if (f_hi > nfr) f_hi = nfr;
    int f_lo = (t - (kWin - 1) + kHop - 1) / kHop;
    if (t - (kWin - 1) < 0) f_lo = 0;
    float acc = 0.f, env = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {
      const int i = t - f * kHop;
      const float w = hann[i];
      acc += y[(long)f * kWin + i];
      env += w * w;
    }
    audio[(long)lo * kHop + s] = tanhf(acc / env);
  }
}

// ---------------------------------------------------------------------------------------------
// Conv-form STFT of the reference's ONNX export (models/stft.py:98-187): conv1d with the
// windowed DFT matrices = a windowed DFT per frame, with these differences from torch.stft / torch.istft:
//   * 'replicate' padding of n_fft/2 samples instead of reflect;
//   * the periodic Hann(win) sits at the START of the n_fft frame (zero padded at the end, stft.py:39-46), not centred;
//   * magnitude = sqrt(re^2 + im^2 + 1e-14) and the outputs are re/mag, im/mag (stft.py:131-139);
//   * the inverse sums the bins ONE-SIDED (no doubling of the inner bins, stft.py:73-77), scales by 1/n_fft, windows again and
//     overlap-adds with NO window-envelope normalisation, then trims n_fft/2 samples on both sides.
// Utterance u has F_u = frame_off[u+1] - frame_off[u] frames and (F_u - 1) * hop samples starting at hop * (frame_off[u] - u).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv_stft_kernel(const float* __restrict__ wave, const int* __restrict__ frame_off, int hop,
                                                        const float* __restrict__ hann, const double2* __restrict__ twiddle, float* __restrict__ mag,
                                                        float* __restrict__ xo, float* __restrict__ yo, int ld) {
  constexpr int H = kNfft / 2;
  __shared__ double2 A[H], Bf[H];
  const int u = blockIdx.y, f = blockIdx.x;
  const int lo = frame_off[u], nfr = frame_off[u + 1] - lo;
  if (f >= nfr) return;
  const long L = (long)(nfr - 1) * hop;
  const float* x = wave + (long)hop * (lo - u);
  for (int n = threadIdx.x; n < H; n += 256) {
    double v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int p = 2 * n + e;
      double val = 0.0;
      if (p < kWin) {
        long m = (long)f * hop - kNfft / 2 + p;
        m = m < 0 ? 0 : (m >= L ? L - 1 : m);  // replicate
        val = (double)(x[m] * hann[p]);
      }
      v[e] = val;
    }
    A[n] = make_double2(v[0], v[1]);
  }
  __syncthreads();
  const double2* Z = fft_lds<double2, H, false>(A, Bf, twiddle, 2);
  const long row = (long)(lo + f) * ld;
  for (int k = threadIdx.x; k < ld; k += 256) {
    float m = 0.f, cx = 0.f, cy = 0.f;
    if (k < kBins) {
      double re, im;
      if (k == 0 || k == H) {
        re = k == 0 ? Z[0].x + Z[0].y : Z[0].x - Z[0].y;
        im = 0.0;
      } else {
        const double2 a = Z[k], b = Z[H - k];
        const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);
        const double orr = 0.5 * (a.y + b.y), oi = -0.5 * (a.x - b.x);
        const double2 w = twiddle[k];
        re = er + orr * w.x - oi * w.y;
        im = ei + orr * w.y + oi * w.x;
      }
      const float fr = (float)re, fi = (float)im;
      m = sqrtf(fr * fr + fi * fi + 1e-14f);
      cx = fr / m;
      cy = fi / m;
    }
    mag[row + k] = m;
    xo[row + k] = cx;
    yo[row + k] = cy;
  }
}

// inverse, step 1: frame f -> its 1200 windowed samples  w[n] / N * Re sum_{k=0}^{N/2} X_k e^{+2 pi i k n / N},  X = mag (x + i y).
// one-sided sum = (Hermitian inverse + (Re X_0 + (-1)^n Re X_{N/2}) / N) / 2, so the half-size transform of istft_frames_kernel serves.
__global__ void __launch_bounds__(256) conv_istft_frames_kernel(const float* __restrict__ mag, const float* __restrict__ xi, const float* __restrict__ yi,
                                                                int ld, const int* __restrict__ frame_off, const float* __restrict__ hann,
                                                                const float2* __restrict__ twiddle, float* __restrict__ yw) {
  constexpr int H = kNfft / 2;
  __shared__ float2 Xs[H + 1], A[H], Bf[H], tw[H];
  const int u = blockIdx.y, f = blockIdx.x;
  const int lo = frame_off[u], nfr = frame_off[u + 1] - lo;
  if (f >= nfr) return;
  const long row = (long)(lo + f) * ld;
  for (int i = threadIdx.x; i < H; i += 256) tw[i] = twiddle[i];
  for (int k = threadIdx.x; k < kBins; k += 256) {
    const float m = mag[row + k];
    Xs[k] = make_float2(m * xi[row + k], (k == 0 || k == H) ? 0.f : m * yi[row + k]);
  }
  __syncthreads();
  const float re0 = Xs[0].x, reN = Xs[H].x;
  for (int k = threadIdx.x; k < H; k += 256) {
    const float2 x = Xs[k], y = Xs[H - k];
    const float er = 0.5f * (x.x + y.x), ei = 0.5f * (x.y - y.y);
    const float dr = 0.5f * (x.x - y.x), di = 0.5f * (x.y + y.y);
    const float2 w = tw[k];
    const float orr = dr * w.x + di * w.y, oi = di * w.x - dr * w.y;
    A[k] = make_float2(er - oi, ei + orr);
  }
  __syncthreads();
  const float2* z = fft_lds<float2, H, true>(A, Bf, tw, 2);
  float* o = yw + (long)(lo + f) * kWin;
  for (int n = threadIdx.x; n < kWin; n += 256) {
    const float2 v = z[n >> 1];
    const float full = ((n & 1) ? v.y : v.x) * (1.0f / H);
    o[n] = (0.5f * full + 0.5f * (re0 + ((n & 1) ? -reN : reN)) * (1.0f / kNfft)) * hann[n];
  }
}

// inverse, step 2: overlap-add (gather form, deterministic) and centre trim: out sample s of utterance u is position t = s + n_fft/2
// of the untrimmed signal; frame f contributes yw[f][t - hop f] while that index is in [0, 1200).
__global__ void __launch_bounds__(256) conv_istft_ola_kernel(const float* __restrict__ yw, const int* __restrict__ frame_off, int hop, float* __restrict__ out) {
  const int u = blockIdx.y;
  const int lo = frame_off[u], nfr = frame_off[u + 1] - lo;
  const long nsamp = (long)(nfr - 1) * hop;
  const float* y = yw + (long)lo * kWin;
  float* o = out + (long)hop * (lo - u);
  for (long s = (long)blockIdx.x * 256 + threadIdx.x; s < nsamp; s += (long)gridDim.x * 256) {
    const long t = s + kNfft / 2;
    long f_hi = t / hop;
    if (f_hi > nfr - 1) f_hi = nfr - 1;
    long f_lo = (t - (kWin - 1) + hop - 1) / hop;
    if (t - (kWin - 1) < 0) f_lo = 0;
    float acc = 0.f;
    for (long f = f_lo; f <= f_hi; ++f) acc += y[f * kWin + (t - f * hop)];
    o[s] = acc;
  }
}

}  // namespace stts
