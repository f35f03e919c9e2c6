// Micro-probe: what does each ingredient of the GEMM K-step cost on MI355X?  (tuning aid, not product code)
// Each block = 8 waves (2 per SIMD); per iteration every wave does 32 MFMA 16x16x32 on 128 accumulators
// (the 128x64 wave tile of the 256x256 block) plus, optionally, the LDS-DMA issues, fragment reads and
// barriers of one 32-deep K-step.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(T) __attribute__((address_space(3))) T*

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE bits: 1 = fragment reads (12 ds_read_b128 / iter), 2 = LDS-DMA (4 x 1 KB per wave / iter),
//            4 = one barrier / iter, 8 = two barriers + stagger, 16 = DMA source walks a large buffer (HBM/L2 misses)
//            32 = 32x32x16 MFMAs instead of 16x16x32
//            64 = DMA source follows the real NT GEMM pattern (M=25600, N=3072, K=768, 256x256 tiles, XCD remap,
//                 24 K-steps per tile, 5 tiles per block);  128 = plus a direct bf16 C-tile store after each tile
template <int MODE>
__global__ __launch_bounds__(512) void probe(char* __restrict__ src, long src_bytes, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 af[8], bfr[4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) af[i][e] = (__bf16)(float)(lane + i + e);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) bfr[i][e] = (__bf16)(float)(lane - i + e);
  // zero LDS so reads are defined
  for (int i = tid; i < 131072 / 16; i += 512) *(LDS_PTR(f32x4))(smem + i * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long blk_base = (MODE & 16) ? ((long)blockIdx.x * 786432) % (src_bytes - 4194304) : (long)(blockIdx.x & 7) * 32768;
  const bool g1 = (MODE & 8) && wave >= 4;
  if (g1) __builtin_amdgcn_s_barrier();
  int slot = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 64) {
      const int round = it / 24, ks = it - round * 24;
      const int bid = round * 256 + blockIdx.x;
      const int wg = min((bid & 7) * 150 + (bid >> 3), 1199);
      const int tm = wg / 12, tn = wg - tm * 12;
      const char* ka = src + (long)tm * 256 * 1536 + ks * 64;
      const char* kbb = src + 25600L * 1536 + (long)tn * 256 * 1536 + ks * 64;
      __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)ka, 0, 0x80000000, 0x00020000);
      __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)kbb, 0, 0x80000000, 0x00020000);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int j = wave * 2 + t;
        const unsigned off = (unsigned)((j * 16 + (lane >> 2)) * 1536 + (lane & 3) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_PTR(void))(smem + ((slot + 3) & 3) * 32768 + j * 1024), 16, (int)off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_PTR(void))(smem + ((slot + 3) & 3) * 32768 + 16384 + j * 1024), 16, (int)off, 0, 0, 0);
      }
    } else if (MODE & 2) {
      const char* kb = src + blk_base + ((MODE & 16) ? (long)it * 64 : (long)(it & 7) * 4096);
      __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, 0x80000000, 0x00020000);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = wave * 4 + t;
        const unsigned off = (MODE & 16) ? (unsigned)((j * 16 + (lane >> 2)) * 1536 + (lane & 3) * 16) : (unsigned)(j * 1024 + lane * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_PTR(void))(smem + ((slot + 3) & 3) * 32768 + j * 1024), 16, (int)off, 0, 0, 0);
      }
    }
    if (MODE & 1) {
      const char* sa = smem + slot * 32768;
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = *(LDS_PTR(const bf16x8))(sa + ((wave >> 2) * 128 + i * 16 + (lane & 15)) * 64 + (((lane >> 4) ^ ((lane >> 2) & 3)) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) bfr[i] = *(LDS_PTR(const bf16x8))(sa + 16384 + ((wave & 3) * 64 + i * 16 + (lane & 15)) * 64 + (((lane >> 4) ^ ((lane >> 2) & 3)) << 4));
    }
    if (MODE & 2) wait_vmcnt<8>();
    if (MODE & 1) __builtin_amdgcn_s_waitcnt(0xC07F);
    if (MODE & (4 | 8)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if (MODE & 32) {
      typedef __attribute__((ext_vector_type(16))) float f32x16;
      f32x16* a16 = (f32x16*)acc;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) a16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[i & 3], af[(i + r) & 7], a16[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi * 4 + ni], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & 8) __builtin_amdgcn_s_barrier();
    if ((MODE & 128) && (it % 24) == 23) {
      const int round = it / 24;
      const int bid = round * 256 + blockIdx.x;
      const int wg = min((bid & 7) * 150 + (bid >> 3), 1199);
      const int tm = wg / 12, tn = wg - tm * 12;
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      __bf16* C = (__bf16*)(src + (64L << 20));
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const long m = tm * 256 + (wave >> 2) * 128 + mi * 16 + (lane & 15), n = tn * 256 + (wave & 3) * 64 + ni * 16 + 4 * (lane >> 4);
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[mi * 4 + ni][e];
          *(bf16x4*)(C + m * 3072 + n) = o;
        }
      wait_vmcnt<0>();
    }
    slot = (slot + 1) & 3;
  }
  if ((MODE & 8) && !g1) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 32; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[blockIdx.x * 512 + tid] = s[0];
}

template <int MODE>
static void run(const char* name, char* src, long src_bytes, float* out, int blocks_per_cu) {
  const int iters = (MODE & 64) ? 120 : 2000, grid = 256 * blocks_per_cu;
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<grid, 512, 131072>>>(src, src_bytes, out, (MODE & 64) ? 120 : 200);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<grid, 512, 131072>>>(src, src_bytes, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 8 * 32 * iters * 16384.0;
  // cycles per iteration at an assumed 2.4 GHz, for 2 waves per SIMD (ideal: 2 * 32 * 16 = 1024)
  printf("%-52s %8.3f ms  %7.1f TFLOP/s   %6.0f clk/iter@2.4GHz\n", name, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / iters / blocks_per_cu);
  fflush(stdout);
}

int main() {
  const long src_bytes = 1L << 30;
  char* src; float* out;
  hipMalloc(&src, src_bytes); hipMemset(src, 0, src_bytes);
  hipMalloc(&out, 256 * 4 * 512 * 4);
  run<0>("mfma 16x16x32 only", src, src_bytes, out, 1);
  run<32>("mfma 32x32x16 only", src, src_bytes, out, 1);
  run<1>("+ 12 frag reads", src, src_bytes, out, 1);
  run<4>("+ barrier", src, src_bytes, out, 1);
  run<1 | 4>("+ frag reads + barrier", src, src_bytes, out, 1);
  run<2>("+ DMA (L2-resident source)", src, src_bytes, out, 1);
  run<2 | 4>("+ DMA(L2) + barrier", src, src_bytes, out, 1);
  run<1 | 2 | 4>("+ frag + DMA(L2) + barrier", src, src_bytes, out, 1);
  run<1 | 2 | 8>("+ frag + DMA(L2) + 2 barriers staggered", src, src_bytes, out, 1);
  run<1 | 8>("+ frag + 2 barriers staggered (no DMA)", src, src_bytes, out, 1);
  run<2 | 16>("+ DMA (strided big source)", src, src_bytes, out, 1);
  run<1 | 2 | 4 | 16>("+ frag + DMA(big) + barrier", src, src_bytes, out, 1);
  run<1 | 2 | 8 | 16>("+ frag + DMA(big) + 2 barriers staggered", src, src_bytes, out, 1);
  run<2 | 64>("GEMM-pattern DMA only", src, src_bytes, out, 1);
  run<1 | 2 | 4 | 64>("GEMM-pattern: frag + DMA + barrier", src, src_bytes, out, 1);
  run<1 | 2 | 8 | 64>("GEMM-pattern: frag + DMA + 2 barriers staggered", src, src_bytes, out, 1);
  run<1 | 2 | 8 | 64 | 128>("GEMM-pattern staggered + direct C store per tile", src, src_bytes, out, 1);
  run<1 | 2 | 4 | 64 | 128>("GEMM-pattern 1 barrier + direct C store per tile", src, src_bytes, out, 1);
  return 0;
}
