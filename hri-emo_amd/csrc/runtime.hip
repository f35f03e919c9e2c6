// Host-side runtime bits of the C-ABI: last-error string, and optional per-kernel-class HIP-event
// timing on the stream the kernels are launched on (used by bench.py's roofline leg only).
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void hriemo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* hriemo_last_error(void) { return g_err; }
extern "C" int hriemo_abi_version(void) { return 1; }

struct ProfClass {
  std::vector<hipEvent_t> ev;  // start/end pairs
  size_t used = 0;
  double work = 0.0;
  long launches = 0;
};
static bool g_prof_on = false;
static ProfClass g_prof[HP_NCLASS];

static hipEvent_t next_event(ProfClass& c) {
  if (c.used == c.ev.size()) {
    hipEvent_t e;
    hipEventCreate(&e);
    c.ev.push_back(e);
  }
  return c.ev[c.used++];
}

void hriemo_prof_begin(int cls, hipStream_t s) {
  if (!g_prof_on) return;
  hipEventRecord(next_event(g_prof[cls]), s);
}
void hriemo_prof_end(int cls, hipStream_t s, double work) {
  if (!g_prof_on) return;
  hipEventRecord(next_event(g_prof[cls]), s);
  g_prof[cls].work += work;
  g_prof[cls].launches += 1;
}

extern "C" int hriemo_prof_enable(int on) {
  for (int c = 0; c < HP_NCLASS; ++c) {
    g_prof[c].used = 0;
    g_prof[c].work = 0.0;
    g_prof[c].launches = 0;
  }
  g_prof_on = on != 0;
  return 0;
}

extern "C" int hriemo_prof_nclass(void) { return HP_NCLASS; }

extern "C" const char* hriemo_prof_name(int cls) {
  static const char* names[HP_NCLASS] = {"gemm_bf16_nt", "gemm_bf16_nn", "gemm_bf16_tn", "attn_fwd",
                                         "attn_bwd_dq",  "attn_bwd_dkv", "rowops", "gemm_mx8_nt"};
  return (cls >= 0 && cls < HP_NCLASS) ? names[cls] : "?";
}

// Synchronises the device; returns total ms, launch count and total work (flops or bytes) of a class.
extern "C" int hriemo_prof_collect(int cls, double* ms_total, long* launches, double* work) {
  HRIEMO_CHECK(cls >= 0 && cls < HP_NCLASS, "prof: bad class %d", cls);
  hipDeviceSynchronize();
  ProfClass& c = g_prof[cls];
  double ms = 0.0;
  for (size_t i = 0; i + 1 < c.used; i += 2) {
    float t = 0.f;
    hipEventElapsedTime(&t, c.ev[i], c.ev[i + 1]);
    ms += t;
  }
  *ms_total = ms;
  *launches = c.launches;
  *work = c.work;
  return 0;
}


// Tuning hook: occupy `blocks` CUs for about `micros` microseconds (one 100 KB-LDS block each, so nothing else fits
// beside it) -- stands in for a collective holding CUs while the persistent kernels run (scripts_dev/bench_hog.py).
__global__ __launch_bounds__(64) void hog_kernel(long ticks, float* sink) {
  extern __shared__ char hog_lds[];
  hog_lds[threadIdx.x] = 1;
  const long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (hog_lds[threadIdx.x] == 77) sink[0] = 1.f;
}
extern "C" int hriemo_debug_hog(int blocks, int micros, float* sink, hipStream_t st) {
  HRIEMO_CHECK(blocks > 0 && blocks <= 256 && micros > 0 && micros <= 100000 && sink != nullptr, "debug_hog: bad arguments");
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); attr = true; }
  hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(64), 100 * 1024, st, (long)micros * 100, sink);   // 100 MHz clock
  HRIEMO_LAUNCH_CHECK("hog_kernel");
  return 0;
}
