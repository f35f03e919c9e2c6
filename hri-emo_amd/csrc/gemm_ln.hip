// Linear + bias + dropout + residual + LayerNorm in ONE kernel (north_star's "fused bias + LayerNorm + residual epilogue"):
//   g = A[M,K] . W[D,K]^T + bias          (rounded to bf16: what the separate GEMM stores and the backward reads)
//   y = LayerNorm(x + drop(g)) * gamma + beta
// for the six post-LN sites of a fusion layer (cross_modal_block_tacfn.py:81,92,105,106,118,119).  LayerNorm needs whole rows,
// so a workgroup owns a FULL-ROW tile: 64 rows x D columns, 8 waves side by side (64 x D/8 each), K in 32-deep steps through a
// 3-slot LDS ring filled by LDS-DMA (the 256x128 kernel's staging, gemm_common.h).  A D-wide weight stage is D x 64 B (48 KB at
// D = 768): three of them are all the LDS a CU has, and every workgroup streams the whole weight matrix for its 64 rows -- the
// two structural costs of this tile.  The epilogue keeps the 64 x D sums in the accumulator registers, reduces
// the row statistics across the eight waves through LDS (two-pass variance, as rowops.hip), and writes g (bf16, for the
// backward), y (bf16) and its fp32 twin straight from registers.  Dropout and row keys are those of hriemo_add_ln_fwd(_rows).
#include "gemm_common.h"

struct GemmLnArgs {
  int M, K;
  const bf16_t* A; long lda;
  const bf16_t* W; long ldw;
  const float* bias;
  const bf16_t* X16; const float* X32;     // residual [M,D]: the fp32 twin when given, else the bf16 tensor
  const float* gamma; const float* beta;
  bf16_t* G; bf16_t* Y16; float* Y32; float* mean; float* rstd;
  const long long* rows;                    // packed sequences: the padded row that keys the dropout hash (or NULL)
  float eps;
  uint32_t thr16; float inv_keep; uint64_t seed; const unsigned long long* seed_dev; uint32_t site; long row_off;
};

template <int D, bool XF32>
__global__ __launch_bounds__(512) void gemm_ln_fwd_kernel(const GemmLnArgs p) {
  constexpr int BM = 64, BK = 32, NS = 3, NW = 8;
  constexpr int WCOLS = D / NW, MT = BM / 16, NTL = WCOLS / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = D * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int B_PW = B_BYTES / 1024 / NW;              // 1 KiB LDS-DMA pieces of the weight stage per wave
  static_assert(WCOLS % 16 == 0 && B_PW * NW * 1024 == B_BYTES && A_BYTES == 4 * 1024, "tile must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const int nk = (p.K + BK - 1) / BK;
  const LaneOff aoff = operand_lane<0, BM, BK>(p.lda, lane), boff = operand_lane<0, D, BK>(p.ldw, lane);
  const bf16_t* abase = p.A + (long)m0 * p.lda;
  const bool loads_a = wave < 4;                         // the 4 KB activation stage is four pieces: waves 0..3 carry one each
  auto stage = [&](int s, int kstep) {
    char* sa = smem + s * STAGE;
    const int krem = p.K - kstep * BK;
    if (loads_a) stage_operand<0, BM, 1, BK>(sa, abase + kstep * BK, aoff, p.lda, p.M - m0, krem, wave, lane);
    stage_operand<0, D, B_PW, BK>(sa + A_BYTES, p.W + kstep * BK, boff, p.ldw, D, krem, wave, lane);
  };
  f32x4 acc[MT][NTL];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NTL; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Ring of three slots, fragments one K-step ahead in a second register set: while the 24 MFMAs of step `it` run from one
  // set, the 10 fragment reads of step it+1 fill the other and the LDS-DMA of step it+3 refills the slot step `it` was read
  // from.  One barrier per K-step; counted vmcnt (waves 0..3 carry one more load per stage: the activation piece).
  auto load_frags = [&](int slot, bf16x8 (&af)[MT], bf16x8 (&bfr)[NTL]) {
    const char* sa = smem + slot * STAGE;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) af[mi] = lds_row_frag<BK>(sa, mi * 16, 0, lane);
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni) bfr[ni] = lds_row_frag<BK>(sb, wave * WCOLS + ni * 16, 0, lane);
  };
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  if (nk > 2) stage(2, 2);
  if (nk > 2) { if (loads_a) wait_vmcnt<2 * B_PW + 2>(); else wait_vmcnt<2 * B_PW>(); }
  else if (nk > 1) { if (loads_a) wait_vmcnt<B_PW + 1>(); else wait_vmcnt<B_PW>(); }
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  bf16x8 afA[MT], bfA[NTL], afB[MT], bfB[NTL];
  load_frags(0, afA, bfA);
  int cur = 0;                           // slot of stage `it`
  auto kstep = [&](int it, bf16x8 (&af)[MT], bf16x8 (&bfr)[NTL], bf16x8 (&afn)[MT], bf16x8 (&bfn)[NTL]) {
    const int nslot = (cur + 1 == NS) ? 0 : cur + 1;
    if (it + 1 < nk) {
      // stage it+1 must have landed; stage it+2 (if any) may still be in flight
      if (it + 2 < nk) { if (loads_a) wait_vmcnt<B_PW + 1>(); else wait_vmcnt<B_PW>(); }
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of stage `it` are in registers -> its slot may be refilled
    __builtin_amdgcn_s_barrier();
    if (it + 3 < nk) stage(cur, it + 3);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(nslot, afn, bfn);             // (unconditional: past the last stage it reads a stale slot and nobody uses it)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {     // the reads trickle between the MFMAs instead of bursting in front of them
      __builtin_amdgcn_sched_group_barrier(0x008, MT * NTL / 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, (MT + NTL + 3) / 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = nslot;
  };
  for (int it = 0; it < nk; it += 2) {
    kstep(it, afA, bfA, afB, bfB);
    if (it + 1 < nk) kstep(it + 1, afB, bfB, afA, bfA);
  }

  // ---------------------------------------------------------------- epilogue: the 64 x D sums stay in acc[][]
  // lane (i = lane & 15, g = lane >> 4) owns row mi*16 + i, columns wave*WCOLS + ni*16 + 4g .. +3 of accumulator tile (mi, ni).
  // Straight-line code: every global access is a raw-buffer access whose per-lane offset is out of range for rows >= M (loads
  // return zeros, stores are dropped) and whose descriptor has zero records for an absent operand -- no EXEC branches, so the
  // 24 residual loads of a lane are all in flight before the first is consumed (hipcc otherwise waits vmcnt(0) per branch).
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4e;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2e;
  const int g = lane >> 4, i = lane & 15;
  const int colw = wave * WCOLS + 4 * g;                 // + ni*16
  const unsigned total = (unsigned)p.M * (unsigned)D;    // elements of an [M, D] operand (host: M*D*4 < 2^31)
  auto rsrc_of = [&](const void* ptr, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, ptr != nullptr ? (int)bytes : 0, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rx = XF32 ? rsrc_of(p.X32, total * 4u) : rsrc_of(p.X16, total * 2u);
  const __amdgpu_buffer_rsrc_t rg = rsrc_of(p.G, total * 2u), ry = rsrc_of(p.Y16, total * 2u), ry32 = rsrc_of(p.Y32, total * 4u);
  const __amdgpu_buffer_rsrc_t rbias = rsrc_of(p.bias, (unsigned)D * 4u);
  unsigned eo[MT];                                        // element offset of (row, colw); rows >= M: out of every range
  uint32_t rk[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const long m = (long)m0 + mi * 16 + i;
    const bool rv = m < p.M;
    eo[mi] = rv ? (unsigned)(m * D + colw) : 0x20000000u;    // x2 and x4 stay beyond every operand (host: M*D*4 < 2^30), no 32-bit wrap
    const long mc = rv ? m : 0;
    rk[mi] = (uint32_t)(p.row_off + (p.rows != nullptr ? (long)p.rows[mc] : mc));
  }
  f32x4 xv[MT][NTL];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni) {
      if (XF32) {
        xv[mi][ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(eo[mi] * 4u), ni * 64, 0));
      } else {
        const bf16x4 xb = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rx, (int)(eo[mi] * 2u), ni * 32, 0));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) xv[mi][ni][jj] = (float)xb[jj];
      }
    }
  f32x4 bs[NTL];
#pragma unroll
  for (int ni = 0; ni < NTL; ++ni) bs[ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, colw * 4, ni * 64, 0));
  __syncthreads();                       // the ring is free: its first 2 KB become the cross-wave reduction scratch
  float* red = (float*)smem;             // [64 rows][8 waves]
  const uint32_t dkey = p.thr16 != 0 ? site_key(eff_seed(p.seed, p.seed_dev), p.site, 0u) : 0u;
  float mu[MT], rs[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    float psum = 0.f;
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni) {
      f32x4 v = acc[mi][ni] + bs[ni];
      bf16x4 gb;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) { gb[jj] = (bf16_t)v[jj]; v[jj] = (float)gb[jj]; }
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2e, gb), rg, (int)(eo[mi] * 2u), ni * 32, 0);
      if (p.thr16 != 0) {
        const uint32_t cp = (uint32_t)((colw + ni * 16) >> 1);
        const uint32_t x0 = mix24(drop_base(dkey, rk[mi], cp));
        const uint32_t x1 = mix24(drop_base(dkey, rk[mi], cp + 1u));
        v[0] = keep_lo(x0, p.thr16) ? v[0] * p.inv_keep : 0.f;
        v[1] = keep_hi(x0, p.thr16) ? v[1] * p.inv_keep : 0.f;
        v[2] = keep_lo(x1, p.thr16) ? v[2] * p.inv_keep : 0.f;
        v[3] = keep_hi(x1, p.thr16) ? v[3] * p.inv_keep : 0.f;
      }
      v += xv[mi][ni];
      acc[mi][ni] = v;
      psum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    if (g == 0) red[(mi * 16 + i) * NW + wave] = psum;
  }
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int r = mi * 16 + i;
    const f32x4 a = *(const f32x4*)(red + r * NW), b = *(const f32x4*)(red + r * NW + 4);
    mu[mi] = (((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]))) * (1.f / (float)D);
  }
  __syncthreads();                       // everyone has its means before the scratch is reused for the squares
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    float q = 0.f;
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) { const float t = acc[mi][ni][jj] - mu[mi]; q += t * t; }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (g == 0) red[(mi * 16 + i) * NW + wave] = q;
  }
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int r = mi * 16 + i;
    const f32x4 a = *(const f32x4*)(red + r * NW), b = *(const f32x4*)(red + r * NW + 4);
    const float var = (((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]))) * (1.f / (float)D);
    rs[mi] = 1.f / sqrtf(var + p.eps);
    const long m = (long)m0 + r;
    if (wave == 0 && g == 0 && m < p.M) { p.mean[m] = mu[mi]; p.rstd[m] = rs[mi]; }
  }
#pragma unroll
  for (int ni = 0; ni < NTL; ++ni) {
    const f32x4 gm = *(const f32x4*)(p.gamma + colw + ni * 16), bt = *(const f32x4*)(p.beta + colw + ni * 16);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      f32x4 y;
      bf16x4 yb;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) { y[jj] = (acc[mi][ni][jj] - mu[mi]) * rs[mi] * gm[jj] + bt[jj]; yb[jj] = (bf16_t)y[jj]; }
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2e, yb), ry, (int)(eo[mi] * 2u), ni * 32, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4e, y), ry32, (int)(eo[mi] * 4u), ni * 64, 0);
    }
  }
}

template <int D, bool XF32>
static int launch_gemm_ln(const GemmLnArgs& a, hipStream_t st) {
  constexpr int LDS = 3 * (64 * 32 * 2 + D * 32 * 2);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_ln_fwd_kernel<D, XF32>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      hriemo_set_error("gemm_ln_fwd: cannot reserve %d bytes of LDS", LDS);
      return 1;
    }
    attr = true;
  }
  hipLaunchKernelGGL((gemm_ln_fwd_kernel<D, XF32>), dim3((a.M + 63) / 64), dim3(512), LDS, st, a);
  return 0;
}

// 1 when hriemo_gemm_ln_fwd is built for this width
extern "C" int hriemo_gemm_ln_supported(int d) { return d == 256 || d == 512 || d == 768; }

extern "C" int hriemo_gemm_ln_fwd(int M, int d, int K, const void* A, long lda, const void* W, long ldw, const float* bias, const void* X16,
                                  const float* X32, const float* gamma, const float* beta, void* G, void* Y16, float* Y32, float* mean,
                                  float* rstd, float eps, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                                  unsigned site, long row_offset, const long long* row_index, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && K > 0 && hriemo_gemm_ln_supported(d), "gemm_ln_fwd: d=%d is not built (256, 512, 768) or empty problem", d);
  HRIEMO_CHECK(K % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && lda < (1L << 21) && ldw < (1L << 21), "gemm_ln_fwd: K and leading dimensions must be multiples of 8 (16-byte rows)");
  HRIEMO_CHECK(A != nullptr && W != nullptr && gamma != nullptr && beta != nullptr && Y16 != nullptr && mean != nullptr && rstd != nullptr &&
                   (X16 != nullptr || X32 != nullptr), "gemm_ln_fwd: missing operand");
  HRIEMO_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)X32 % 16) == 0 && ((uintptr_t)X16 % 8) == 0 &&
                   ((uintptr_t)G % 8) == 0 && ((uintptr_t)Y16 % 8) == 0 && ((uintptr_t)Y32 % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                   ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0, "gemm_ln_fwd: unaligned operand");
  HRIEMO_CHECK(p_drop >= 0.f && p_drop < 1.f, "gemm_ln_fwd: dropout p=%f", (double)p_drop);
  HRIEMO_CHECK((long)M * d * 4 < (1L << 30), "gemm_ln_fwd: M*d = %ld exceeds the 32-bit offsets of the epilogue", (long)M * d);
  const DropCfg dc = make_drop(p_drop, seed, site);
  GemmLnArgs a;
  a.M = M; a.K = K; a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw; a.bias = bias;
  a.X16 = (const bf16_t*)X16; a.X32 = X32; a.gamma = gamma; a.beta = beta; a.G = (bf16_t*)G; a.Y16 = (bf16_t*)Y16; a.Y32 = Y32;
  a.mean = mean; a.rstd = rstd; a.rows = row_index; a.eps = eps;
  a.thr16 = dc.thr16; a.inv_keep = dc.inv_keep; a.seed = seed; a.seed_dev = seed_dev; a.site = site; a.row_off = row_offset;
  hriemo_prof_begin(HP_GEMM_NT, st);
  int rc;
  const bool xf = X32 != nullptr;
  switch (d) {
    case 256: rc = xf ? launch_gemm_ln<256, true>(a, st) : launch_gemm_ln<256, false>(a, st); break;
    case 512: rc = xf ? launch_gemm_ln<512, true>(a, st) : launch_gemm_ln<512, false>(a, st); break;
    default: rc = xf ? launch_gemm_ln<768, true>(a, st) : launch_gemm_ln<768, false>(a, st); break;
  }
  if (rc != 0) return rc;
  HRIEMO_LAUNCH_CHECK("gemm_ln_fwd_kernel");
  hriemo_prof_end(HP_GEMM_NT, st, 2.0 * M * (double)d * K);
  return 0;
}
