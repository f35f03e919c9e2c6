// Matrix-pipe ceilings on this chip, register operands only (no LDS, no memory): 8 waves per CU x 256 CUs, independent
// accumulator chains.  Build: hipcc --offload-arch=gfx950 -O3 scripts_dev/mfma_rate.hip -o scripts_dev/mfma_rate
// Planning aid for the fp8 operand path: what the fp8 instructions give over bf16 16x16x32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int MODE>
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  bf16x8 a16, b16;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a16[j] = (__bf16)(float)(lane + j); b16[j] = (__bf16)(float)(lane - j); }
  long a8 = 0x3838383838383838L + lane, b8 = 0x3838383838383838L - lane;     // 8 x fp8 e4m3
  i32x8 a32, b32;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a32[j] = 0x38383838 + lane + j; b32[j] = 0x38383838 - lane - j; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16, b16, acc[i], 0, 0, 0);
      if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, b8, acc[i], 0, 0, 0);
      if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a32, b32, acc[i], 0, 0, 0, 127, 0, 127);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name, double flop_per_mfma) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 4000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(512), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * 8 /*waves*/ * iters * 8;
  printf("%-34s %8.3f ms  %8.1f TFLOP/s\n", name, ms, mfmas * flop_per_mfma / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<0>("mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32);
  run<1>("mfma_f32_16x16x32_fp8_fp8", 2.0 * 16 * 16 * 32);
  run<2>("mfma_scale_f32_16x16x128_f8f6f4", 2.0 * 16 * 16 * 128);
  return 0;
}
