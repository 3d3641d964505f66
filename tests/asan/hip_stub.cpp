// Host-only stand-in for the HIP runtime, used ONLY by the CPU sanitizer build of the library's host side
// (tests/test_asan_host.py; SURVEY.md section 5: sanitizers run on the CPU build).  "Device" memory is host memory
// (64 bytes of red zone checked by AddressSanitizer like any other heap block), copies are memcpy, kernel launches
// are counted and otherwise ignored: what runs under the sanitizers is the library's own host code - weight folding and
// packing, Winograd / fragment packing, workspace carving, launch planning - over the shapes of BASELINE's configs.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

static long g_launches = 0;
extern "C" long stts_stub_launch_count() { return g_launches; }

extern "C" {
char stts_stub_fatbin[16] = {0};

hipError_t hipMalloc(void** p, size_t n) {
  *p = malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  memcpy(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  memcpy(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) {
  memset(d, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  memset(d, v, n);
  return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
  p->multiProcessorCount = 256;
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) {
  *e = (hipEvent_t)malloc(8);
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  free(e);
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = (hipStream_t)malloc(8);
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  free(s);
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* status) {
  *status = hipStreamCaptureStatusNone;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
  *ms = 0.001f;
  return hipSuccess;
}
hipError_t hipLaunchKernel(const void*, dim3 grid, dim3 block, void**, size_t, hipStream_t) {
  // what a real launch would reject
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024 || grid.y > 65535 || grid.z > 65535) {
    fprintf(stderr, "hip_stub: invalid launch configuration grid (%u,%u,%u) block (%u,%u,%u)\n", grid.x, grid.y, grid.z, block.x, block.y, block.z);
    abort();
  }
  ++g_launches;
  return hipSuccess;
}
hipError_t hipExtLaunchKernel(const void* f, dim3 grid, dim3 block, void** a, size_t s, hipStream_t st, hipEvent_t, hipEvent_t, int) {
  return hipLaunchKernel(f, grid, block, a, s, st);
}
void** __hipRegisterFatBinary(const void*) {
  static void* h = nullptr;
  return &h;
}
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
static thread_local struct {
  dim3 g, b;
  size_t s;
  hipStream_t st;
} g_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t s, hipStream_t st) {
  g_cfg.g = g;
  g_cfg.b = b;
  g_cfg.s = s;
  g_cfg.st = st;
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* s, hipStream_t* st) {
  *g = g_cfg.g;
  *b = g_cfg.b;
  *s = g_cfg.s;
  *st = g_cfg.st;
  return hipSuccess;
}
}
