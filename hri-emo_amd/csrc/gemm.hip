// bf16 MFMA GEMM for gfx950 with the three operand layouts the fusion path needs:
//   NT  C[M,N] = A[M,K]  . B[N,K]^T   forward projections / FFN (weights are [out,in])
//   NN  C[M,N] = A[M,K]  . B[K,N]     dX = dY . W
//   TN  C[M,N] = A[K,M]^T. B[K,N]     dW = dY^T . X   (reduction over the B*L rows, split-K)
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles.
// Operand tiles are staged HBM -> LDS with 16-byte global_load_lds (no VGPR round trip), two
// LDS stages (64 KiB), one barrier per K-step.  LDS images are lane-linear (LDS-DMA rule) and
// XOR-swizzled through the per-lane SOURCE address + the same XOR on the fragment read:
//   K-contiguous operand: [128 rows][64 k] bf16, 128-B rows, chunk ^= row&7     -> ds_read_b128
//   K-strided operand:    [64 k][128 cols] bf16, 256-B rows, 32-B pair ^= key(k) -> ds_read_b64_tr_b16
// Accumulators are produced transposed (mfma(Bfrag, Afrag)) so a lane owns 4 consecutive
// columns of one row: 8-byte bf16 / 16-byte fp32 stores.
#include "common.h"

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];

struct GemmArgs {
  int M, N, K;
  const bf16_t* A; long lda;
  const bf16_t* B; long ldb;
  void* C; long ldc;
  const float* bias;
  const bf16_t* aux; long ldaux;
  int epi;  // 0 none, 1 relu, 2 multiply by (aux > 0), 3 add aux
  int tiles_m, tiles_n, splitk, k_per_split;
  float* ws;
  int accumulate;
};

__device__ __forceinline__ void glds16(const bf16_t* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((GLB_PTR(const void))g, (LDS_PTR(void))lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 lds_row_frag(const char* tile, int sub0, int ks, int lane) {
  const int row = sub0 + (lane & 15);
  const int c = ks * 4 + (lane >> 4);
  return *(LDS_PTR(const bf16x8))(tile + row * 128 + ((c ^ (row & 7)) << 4));
}

__device__ __forceinline__ bf16x8 lds_tr_frag(const char* tile, int sub0, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int kr = ks * 32 + 8 * g + qq;
  const int key = qq | ((g & 1) << 2);
  const int c = ((sub0 >> 3) ^ (key << 1)) | (pp >> 1);
  const char* a0 = tile + kr * 256 + c * 16 + (pp & 1) * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(a0 + 4 * 256));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

template <int TA, int TB, int OUTF32>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware, bijective block remap: blocks b and b+8 share an XCD (L2); give each XCD a
  // contiguous range of tiles so neighbouring tiles (same A row-panel) hit the same L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
  const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int tiles = p.tiles_m * p.tiles_n;
  const int slice = wg / tiles;
  const int t = wg - slice * tiles;
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int kbeg = slice * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + 63) >> 6;
  const bf16_t* zero = (const bf16_t*)g_zero16;

  auto stage = [&](int s, int k0) {
    char* sa = smem + s * 32768;
    char* sb = sa + 16384;
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int j = wave * 4 + t4;
      const bf16_t* src;
      if (TA == 0) {
        const int row = j * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        const int gr = m0 + row, gk = k0 + c * 8;
        src = (gr < p.M && gk < kend) ? p.A + (long)gr * p.lda + gk : zero;
      } else {
        const int kr = j * 4 + (lane >> 4);
        const int key = (kr & 3) | (((kr >> 3) & 1) << 2);
        const int c = (lane & 15) ^ (key << 1);
        const int gk = k0 + kr, gc = m0 + c * 8;
        src = (gk < kend && gc < p.M) ? p.A + (long)gk * p.lda + gc : zero;
      }
      glds16(src, sa + j * 1024);
      if (TB == 0) {
        const int row = j * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        const int gr = n0 + row, gk = k0 + c * 8;
        src = (gr < p.N && gk < kend) ? p.B + (long)gr * p.ldb + gk : zero;
      } else {
        const int kr = j * 4 + (lane >> 4);
        const int key = (kr & 3) | (((kr >> 3) & 1) << 2);
        const int c = (lane & 15) ^ (key << 1);
        const int gk = k0 + kr, gc = n0 + c * 8;
        src = (gk < kend && gc < p.N) ? p.B + (long)gk * p.ldb + gc : zero;
      }
      glds16(src, sb + j * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    stage(0, kbeg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    if (it + 1 < nk) stage(cur ^ 1, kbeg + (it + 1) * 64);
    const char* sa = smem + cur * 32768;
    const char* sb = sa + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        af[mi] = TA == 0 ? lds_row_frag(sa, wm * 64 + mi * 16, ks, lane) : lds_tr_frag(sa, wm * 64 + mi * 16, ks, lane);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        bfr[ni] = TB == 0 ? lds_row_frag(sb, wn * 64 + ni * 16, ks, lane) : lds_tr_frag(sb, wn * 64 + ni * 16, ks, lane);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: lane owns C[m = .. + (lane&15)][n = .. + 4*(lane>>4) .. +3]
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + 4 * g;
      if (n >= p.N) continue;
      f32x4 v = acc[mi][ni];
      if (p.bias != nullptr && slice == 0) {
        const f32x4 b4 = *(const f32x4*)(p.bias + n);
        v += b4;
      }
      if (OUTF32) {
        float* dst = p.splitk > 1 ? p.ws + ((long)slice * p.M + m) * p.N + n : (float*)p.C + (long)m * p.ldc + n;
        if (p.splitk == 1 && p.accumulate) v += *(const f32x4*)dst;
        *(f32x4*)dst = v;
      } else {
        if (p.epi == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.epi == 2) {
          const bf16x4 a4 = *(const bf16x4*)(p.aux + (long)m * p.ldaux + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ((float)a4[e] > 0.f) ? v[e] : 0.f;
        } else if (p.epi == 3) {
          const bf16x4 a4 = *(const bf16x4*)(p.aux + (long)m * p.ldaux + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)a4[e];
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
        *(bf16x4*)((bf16_t*)p.C + (long)m * p.ldc + n) = o;
      }
    }
  }
}

// out[m][n] (+)= sum_s ws[s][m][n]
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, long ldc, int M, int N,
                                     int splitk, int accumulate) {
  const long nv = (long)M * N / 4;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
    const long e = v * 4;
    const int m = (int)(e / N), n = (int)(e - (long)m * N);
    f32x4 s = *(const f32x4*)(ws + e);
    for (int k = 1; k < splitk; ++k) s += *(const f32x4*)(ws + (long)k * M * N + e);
    float* dst = C + (long)m * ldc + n;
    if (accumulate) s += *(const f32x4*)dst;
    *(f32x4*)dst = s;
  }
}

template <int TA, int TB, int OUTF32>
static int launch_gemm(const GemmArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_kernel<TA, TB, OUTF32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr_set = true;
  }
  const int grid = a.tiles_m * a.tiles_n * a.splitk;
  hipLaunchKernelGGL((gemm_kernel<TA, TB, OUTF32>), dim3(grid), dim3(256), 65536, st, a);
  return 0;
}

extern "C" int hriemo_gemm_bf16(int ta, int tb, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                                void* C, long ldc, int c_is_f32, const float* bias, int epilogue, const void* aux,
                                long ldaux, int accumulate, float* workspace, long workspace_bytes, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  HRIEMO_CHECK(!(ta == 1 && tb == 0), "gemm: layout (ta=1,tb=0) is not used by this path and not built");
  HRIEMO_CHECK(N % 8 == 0, "gemm: N=%d must be a multiple of 8", N);
  HRIEMO_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm: leading dims must keep 16-byte alignment");
  HRIEMO_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 8) == 0, "gemm: unaligned operand");
  if (ta == 0) HRIEMO_CHECK(K % 8 == 0, "gemm: K=%d must be a multiple of 8 for a K-contiguous operand", K);
  if (ta == 1) HRIEMO_CHECK(M % 8 == 0, "gemm: M=%d must be a multiple of 8 for a transposed A", M);
  HRIEMO_CHECK(c_is_f32 || (epilogue >= 0 && epilogue <= 3), "gemm: bad epilogue");
  HRIEMO_CHECK(epilogue < 2 || (aux != nullptr && ldaux % 4 == 0), "gemm: epilogue 2/3 needs aux");
  HRIEMO_CHECK(!(c_is_f32 && epilogue != 0), "gemm: fp32 output has no activation epilogue");
  HRIEMO_CHECK(c_is_f32 || !accumulate, "gemm: accumulate needs fp32 output");

  GemmArgs a;
  a.M = M; a.N = N; a.K = K;
  a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb;
  a.C = C; a.ldc = ldc; a.bias = bias; a.aux = (const bf16_t*)aux; a.ldaux = ldaux; a.epi = epilogue;
  a.tiles_m = (M + 127) / 128; a.tiles_n = (N + 127) / 128;
  a.ws = workspace; a.accumulate = accumulate;
  int splitk = 1;
  if (c_is_f32) {
    const long tiles = (long)a.tiles_m * a.tiles_n;
    const int ksteps = (K + 63) / 64;
    long want = (768 + tiles - 1) / tiles;           // ~3 blocks per CU in flight
    if (want > ksteps / 4) want = ksteps / 4;        // >= 4 K-steps (256 of K) per slice
    const long fit = workspace ? workspace_bytes / ((long)M * N * 4) : 0;
    if (want > fit) want = fit;
    if (want > 1) splitk = (int)want;
  }
  int kper = ((K + splitk - 1) / splitk + 63) / 64 * 64;
  splitk = (K + kper - 1) / kper;
  a.splitk = splitk; a.k_per_split = kper;

  const int cls = ta ? HP_GEMM_TN : (tb ? HP_GEMM_NN : HP_GEMM_NT);
  hriemo_prof_begin(cls, st);
  if (ta == 0 && tb == 0) {
    if (c_is_f32) launch_gemm<0, 0, 1>(a, st); else launch_gemm<0, 0, 0>(a, st);
  } else if (ta == 0 && tb == 1) {
    if (c_is_f32) launch_gemm<0, 1, 1>(a, st); else launch_gemm<0, 1, 0>(a, st);
  } else {
    if (c_is_f32) launch_gemm<1, 1, 1>(a, st); else launch_gemm<1, 1, 0>(a, st);
  }
  HRIEMO_LAUNCH_CHECK("gemm_kernel");
  if (splitk > 1) {
    const long nv = (long)M * N / 4;
    int grid = (int)((nv + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, workspace, (float*)C, ldc, M, N, splitk,
                       accumulate);
    HRIEMO_LAUNCH_CHECK("splitk_reduce_kernel");
  }
  hriemo_prof_end(cls, st, 2.0 * M * N * K);
  return 0;
}
