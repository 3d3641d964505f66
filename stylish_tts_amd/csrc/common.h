// Shared helpers for the gfx950 kernels of the Stylish-TTS inference hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>

namespace stts {

constexpr int kWave = 64;  // CDNA wavefront

// last error text (per thread); the C-ABI never throws
inline std::string& last_error() {
  static thread_local std::string e;
  return e;
}
inline int fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
inline int fail(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return 1;
}

#define STTS_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) return stts::fail("%s:%d %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)

#define STTS_CHECK(cond, ...)                    \
  do {                                           \
    if (!(cond)) return stts::fail(__VA_ARGS__); \
  } while (0)

#define STTS_TRY(expr)      \
  do {                      \
    int _r = (expr);        \
    if (_r != 0) return _r; \
  } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace stts
