// Host stand-ins for the few HIP runtime calls the HOST-side code of the library makes (geneset.cpp's uploads, the
// context helpers): "device" memory is plain heap memory, so AddressSanitizer sees every byte the planners upload.
// Used only by `make -C plaid_amd/csrc host-asan` (g++ -fsanitize=address,undefined; no GPU, no hipcc): the sanitizers
// are not available for device code on this pool, and the index planning, the GMT parser and the statistics tails are
// host code anyway.  Never linked into libplaidhip.so.
#include <hip/hip_runtime.h>
#include <malloc.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../plaid_amd/csrc/common.h"

static std::atomic<int64_t> g_live{0}, g_peak{0}, g_total{0};   // "device" bytes (tools/plan_probe reports them)
extern "C" {
void plaidhip_stub_alloc_stats(int64_t* live, int64_t* peak, int64_t* total) { *live = g_live; *peak = g_peak; *total = g_total; }
hipError_t hipMalloc(void** p, size_t n) {
  *p = malloc(n ? n : 1);
  if (!*p) return hipErrorOutOfMemory;
  const int64_t now = (g_live += (int64_t)malloc_usable_size(*p));
  g_total += (int64_t)n;
  if (now > g_peak) g_peak = now;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (p) g_live -= (int64_t)malloc_usable_size(p);
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
}

namespace plaidhip {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
int hip_fail(hipError_t, const char* what, const char* file, int line) {
  set_error("%s failed at %s:%d", what, file, line);
  return PLAIDHIP_EHIP;
}
const char* last_error_cstr() { return g_err.c_str(); }
}  // namespace plaidhip
