// bf16 MFMA GEMM for gfx950 with the three operand layouts the fusion path needs:
//   NT  C[M,N] = A[M,K]  . B[N,K]^T   forward projections / FFN (weights are [out,in])
//   NN  C[M,N] = A[M,K]  . B[K,N]     dX = dY . W
//   TN  C[M,N] = A[K,M]^T. B[K,N]     dW = dY^T . X   (reduction over the B*L rows, split-K)
// One kernel template, parameterised by block tile BMxBN (BK = 64), wave grid WMxWN (each wave owns
// (BM/WM)x(BN/WN) as MFMA 16x16x32 tiles) and NS LDS stages:
//  * operand tiles go HBM -> LDS with 16-byte buffer_load ... lds (LDS-DMA, no VGPR round trip; per-lane
//    offsets are loop-invariant, out-of-matrix lanes are zero-filled by the buffer range check) into an
//    NS-deep ring; the prefetch runs NS-1 K-steps ahead and is retired with a COUNTED s_waitcnt vmcnt(N)
//    + raw s_barrier (never vmcnt(0) in the steady state), one barrier per K-step;
//  * LDS images are lane-linear (LDS-DMA rule) and XOR-swizzled through the per-lane SOURCE address +
//    the same XOR on the fragment read:
//      K-contiguous operand: [rows][64 k] bf16, 128-B rows, 16-B chunk ^= row&7      -> ds_read_b128
//      K-strided operand:    [64 k][cols] bf16, 256/512-B rows, 32-B pair ^= key(k)  -> ds_read_b64_tr_b16
//    both conflict-free for the lane groups of those instructions;
//  * accumulators are produced transposed (mfma(Bfrag, Afrag)) so a lane owns 4 consecutive columns of one
//    row; bias / ReLU / ReLU-mask / residual-add are fused in the epilogue; bf16 tiles leave through LDS as
//    full-line 16-byte stores, fp32 (weight-gradient) tiles as 16-byte stores from registers;
//  * blockIdx is remapped (bijectively) so each XCD's L2 sees a contiguous range of tiles.
#include "common.h"

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

__device__ __forceinline__ bf16x8 lds_row_frag(const char* tile, int sub0, int ks, int lane) {
  const int row = sub0 + (lane & 15);
  const int c = ks * 4 + (lane >> 4);
  return *(LDS_PTR(const bf16x8))(tile + row * 128 + ((c ^ (row & 7)) << 4));
}

template <int ROWB>   // bytes per k-row of the strided image (256 or 512)
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* tile, int sub0, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int kr = ks * 32 + 8 * g + qq;
  const int key = qq | ((g & 1) << 2);
  const int c = ((sub0 >> 3) ^ (key << 1)) | (pp >> 1);
  const char* a0 = tile + kr * ROWB + c * 16 + (pp & 1) * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(a0 + 4 * ROWB));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// ---- operand staging through buffer_load ... lds -------------------------------------------------
// Per lane and wave-instruction the byte offset inside the block's operand window is CONSTANT over the
// K loop; only the (wave-uniform) descriptor base advances by one K-step.  So the steady-state loop
// spends no VALU on addresses.  Lanes outside the matrix carry offset 0x80000000 >= num_records, which
// the hardware range check turns into zeros (no branches, no clamping).
#define OOB_OFF 0x80000000u

template <int T, int ROWS, int PW>
__device__ __forceinline__ void operand_offsets(unsigned (&off)[PW], long ld, int r0, int rmax, int wave, int lane) {
#pragma unroll
  for (int t = 0; t < PW; ++t) {
    const int j = wave * PW + t;
    if (T == 0) {
      const int row = j * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      off[t] = (r0 + row < rmax) ? (unsigned)((row * ld + c * 8) * 2) : OOB_OFF;
    } else {
      constexpr int CPR = ROWS / 8;        // 16-B chunks per k-row
      constexpr int RPI = 64 / CPR;        // k-rows per wave-instruction
      const int kr = j * RPI + lane / CPR;
      const int key = (kr & 3) | (((kr >> 3) & 1) << 2);
      const int c = (lane % CPR) ^ (key << 1);
      off[t] = (r0 + c * 8 < rmax) ? (unsigned)((kr * ld + c * 8) * 2) : OOB_OFF;
    }
  }
}

// krem = valid k extent of this K-step (>= 64 in the steady state)
template <int T, int ROWS, int PW>
__device__ __forceinline__ void stage_operand(char* tile, const bf16_t* kbase, const unsigned (&off)[PW], int krem, int wave, int lane) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (int)OOB_OFF, 0x00020000);
#pragma unroll
  for (int t = 0; t < PW; ++t) {
    const int j = wave * PW + t;
    unsigned o = off[t];
    if (krem < 64) {                       // ragged last K-step only
      if (T == 0) {
        const int row = j * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        if (c * 8 >= krem) o = OOB_OFF;
      } else {
        constexpr int CPR = ROWS / 8;
        constexpr int RPI = 64 / CPR;
        if (j * RPI + lane / CPR >= krem) o = OOB_OFF;
      }
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_PTR(void))(tile + j * 1024), 16, (int)o, 0, 0, 0);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TA, int TB, int OUTF32, int BM, int BN, int WM, int WN, int NS, int STAG>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(const GemmArgs p) {
  constexpr int NWAVE = WM * WN, NTHR = NWAVE * 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PW = A_BYTES / 1024 / NWAVE, B_PW = B_BYTES / 1024 / NWAVE, LPT = A_PW + B_PW;
  constexpr int MT = BM / WM / 16, NTL = BN / WN / 16;
  static_assert(A_PW * NWAVE * 1024 == A_BYTES && B_PW * NWAVE * 1024 == B_BYTES, "tile must split evenly over waves");
  static_assert(NS >= 2 && NS <= 4 && (NS - 1) * LPT < 64, "vmcnt range");
  static_assert(OUTF32 || BM * BN * 2 <= NS * STAGE, "C staging must fit the operand ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases stay scalar
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware, bijective block remap: blocks b and b+8 share an XCD (L2); give each XCD a
  // contiguous range of tiles so neighbouring tiles (same A row-panel) hit the same L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
  const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int tiles = p.tiles_m * p.tiles_n;
  const int slice = wg / tiles;
  const int t = wg - slice * tiles;
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = slice * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + 63) >> 6;

  // block-local operand windows (offsets stay small whatever the tensor size)
  const bf16_t* abase = TA == 0 ? p.A + (long)m0 * p.lda + kbeg : p.A + (long)kbeg * p.lda + m0;
  const bf16_t* bbase = TB == 0 ? p.B + (long)n0 * p.ldb + kbeg : p.B + (long)kbeg * p.ldb + n0;
  const long astep = TA == 0 ? 64 : 64 * p.lda, bstep = TB == 0 ? 64 : 64 * p.ldb;
  unsigned aoff[A_PW], boff[B_PW];
  operand_offsets<TA, BM, A_PW>(aoff, p.lda, m0, p.M, wave, lane);
  operand_offsets<TB, BN, B_PW>(boff, p.ldb, n0, p.N, wave, lane);

  auto stage = [&](int s, int kstep) {
#if defined(HRIEMO_GEMM_ABL) && (HRIEMO_GEMM_ABL & 1)
    if (kstep >= NS - 1) return;        // timing-only build: prologue stages only
#endif
    char* sa = smem + s * STAGE;
    const int krem = kend - kbeg - kstep * 64;
    stage_operand<TA, BM, A_PW>(sa, abase + kstep * astep, aoff, krem, wave, lane);
    stage_operand<TB, BN, B_PW>(sa + A_BYTES, bbase + kstep * bstep, boff, krem, wave, lane);
  };

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NTL; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#if defined(HRIEMO_GEMM_ABL)
  int abl_nfrag = 0;
#endif
  auto load_frags = [&](const char* sa, int ks, bf16x8 (&af)[MT], bf16x8 (&bfr)[NTL]) {
#if defined(HRIEMO_GEMM_ABL) && (HRIEMO_GEMM_ABL & 2)
    if (abl_nfrag >= 2) {               // timing-only build: keep the first two fragment sets, keep them live
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) asm volatile("" : "+v"(af[mi]));
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) asm volatile("" : "+v"(bfr[ni]));
      return;
    }
    ++abl_nfrag;
#endif
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
      af[mi] = TA == 0 ? lds_row_frag(sa, wm * MT * 16 + mi * 16, ks, lane) : lds_tr_frag<BM * 2>(sa, wm * MT * 16 + mi * 16, ks, lane);
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni)
      bfr[ni] = TB == 0 ? lds_row_frag(sb, wn * NTL * 16 + ni * 16, ks, lane) : lds_tr_frag<BN * 2>(sb, wn * NTL * 16 + ni * 16, ks, lane);
  };
  auto mma_rows = [&](const bf16x8 (&af)[MT], const bf16x8 (&bfr)[NTL], int lo, int hi) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
      if (mi >= lo && mi < hi) {
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
      }
  };
  auto mma = [&](const bf16x8 (&af)[MT], const bf16x8 (&bfr)[NTL]) { mma_rows(af, bfr, 0, MT); };

  // prologue: put NS-1 stages in flight, retire the first
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) stage(s, s);
  if (nk > NS - 2) wait_vmcnt<(NS - 2) * LPT>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  bf16x8 afA[MT], bfA[NTL], afB[MT], bfB[NTL];
  int cur = 0, nxt = NS - 1;          // ring slots of the stage being computed / being filled
  if (STAG) {
    // Staggered two-group schedule.  A SIMD hosts wave w (group 0) and wave w + NWAVE/2 (group 1).  Every
    // K-step is split into an S-phase (LDS-DMA issue for stage it+NS-1, all fragment reads of stage it) and
    // an M-phase (MFMAs only), each closed by a barrier; group 1 runs one phase behind, so on every SIMD one
    // wave feeds the matrix pipe while its partner pays the DMA-issue / LDS-read time.  Needs NS >= 3:
    // stage it is read at slots 2it (g0) / 2it+1 (g1); every wave confirms its share of stage it at the end
    // of S(it-1) (counted vmcnt, then barrier), and the slot is only restaged in S(it+1), one barrier after
    // the last group-1 read was retired (lgkmcnt(0) before the barrier).
    static_assert(!STAG || NS >= 3, "stagger needs a 3-deep ring");
    const bool g1 = wave >= NWAVE / 2;
    if (g1) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < nk; ++it) {
      if (it + NS - 1 < nk) stage(nxt, it + NS - 1);
      load_frags(smem + cur * STAGE, 0, afA, bfA);
      load_frags(smem + cur * STAGE, 1, afB, bfB);
      if (it + NS - 1 < nk) wait_vmcnt<(NS - 2) * LPT>();
      else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(afA, bfA);
      mma(afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
    if (!g1) __builtin_amdgcn_s_barrier();
  } else {
    // Software pipeline across the barrier: while the MFMAs of one 32-deep half-step run, the fragments of
    // the next half-step are already being read (two register sets), and the retire-wait + barrier for
    // stage it+1 sits in the MIDDLE of iteration it, so its first fragments load under the second cluster.
    // Order inside an iteration (pinned with sched_barrier(0): hipcc otherwise moves the register-only MFMA
    // clusters across the waits and the barrier, rule 18):
    //   DMA(it+NS-1) | MFMA(A, rows 0..MT/2) | read B-set | MFMA(A, rest) | retire stage it+1 + barrier |
    //   read next A-set | MFMA(B-set)
    // so every fragment read has >= half a cluster of MFMAs to land, and the LDS-read retire before the
    // barrier is a builtin s_waitcnt (lgkmcnt(0) only) the compiler can see, so it adds no wait of its own
    // in front of the B cluster.
    // fragment reads of the NEXT half-step are interleaved with the MFMAs of the current one in four
    // equal groups (sched_group_barrier), so the LDS sees a steady trickle instead of 8 waves bursting
    // 12 reads each right after the barrier.
    constexpr int RD_A = (TA == 0 ? 1 : 2) * MT, RD_B = (TB == 0 ? 1 : 2) * NTL;     // ds_read instrs per set
    auto interleave = [&]() {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NTL / 4, 0);               // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, (RD_A + RD_B + 3) / 4, 0);      // DS read
      }
    };
    // 64x64 wave tiles have the registers for it (<= 16 accumulator tiles); the 128x64 wave tile of the
    // 256x256 block is at the 256-VGPR limit and keeps the coarser split below.
    constexpr bool ILV = (MT * NTL <= 16);
    if (nk > 0) load_frags(smem, 0, afA, bfA);
    for (int it = 0; it < nk; ++it) {
      if (it + NS - 1 < nk) stage(nxt, it + NS - 1);
      __builtin_amdgcn_sched_barrier(0);
      if (ILV) {
        load_frags(smem + cur * STAGE, 1, afB, bfB);
        mma(afA, bfA);
        interleave();
      } else {
        __builtin_amdgcn_s_setprio(1);
        mma_rows(afA, bfA, 0, MT / 2);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(smem + cur * STAGE, 1, afB, bfB);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma_rows(afA, bfA, MT / 2, MT);
        __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
      const bool more = it + 1 < nk;
      if (more) {
        // stage it+1 must have landed for every wave; our own reads of stage `it` must be retired before any
        // wave may restage that slot (WAR).
        if (it + NS - 1 < nk) wait_vmcnt<(NS - 2) * LPT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0) alone
        __builtin_amdgcn_s_barrier();
        if (ILV) {                                // reads + MFMAs in ONE scheduling region
          __builtin_amdgcn_sched_barrier(0);
          load_frags(smem + cur * STAGE, 0, afA, bfA);
          mma(afB, bfB);
          interleave();
          __builtin_amdgcn_sched_barrier(0);
        } else {
          load_frags(smem + cur * STAGE, 0, afA, bfA);
        }
      } else {
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (ILV) {
          __builtin_amdgcn_sched_barrier(0);
          mma(afB, bfB);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!ILV) {                                 // big tile: one merged copy of the cluster (register budget)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma(afB, bfB);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();                    // everyone is done with the ring before it becomes the C staging area

  // ---- epilogue.  A lane owns C[m = .. + (lane&15)][n = .. + 4*(lane>>4) .. +3] of each 16x16 tile.
  const int g = lane >> 4, i = lane & 15;
  if (OUTF32) {
    // weight gradients / split-K slabs: 16-byte fp32 stores straight from the accumulators
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int m = m0 + wm * MT * 16 + mi * 16 + i;
      if (m >= p.M) continue;
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) {
        const int n = n0 + wn * NTL * 16 + ni * 16 + 4 * g;
        if (n >= p.N) continue;
        f32x4 v = acc[mi][ni];
        if (p.bias != nullptr && slice == 0) v += *(const f32x4*)(p.bias + n);
        float* dst = p.splitk > 1 ? p.ws + ((long)slice * p.M + m) * p.N + n : (float*)p.C + (long)m * p.ldc + n;
        if (p.splitk == 1 && p.accumulate) v += *(const f32x4*)dst;
        *(f32x4*)dst = v;
      }
    }
  } else {
    // bf16: bias / activation in registers, then through LDS (the operand ring is free after the last
    // barrier) so every global store is 16 bytes of a full 128-B line instead of 8-B row fragments.
    // Image: [BM][BN] bf16, 16-B chunk index ^= row & 15.
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int r = wm * MT * 16 + mi * 16 + i;
      const int m = m0 + r;
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) {
        const int cn = wn * NTL * 16 + ni * 16 + 4 * g;
        const int n = n0 + cn;
        f32x4 v = acc[mi][ni];
        if (m < p.M && n < p.N) {
          if (p.bias != nullptr) v += *(const f32x4*)(p.bias + n);
          if (p.epi == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (p.epi == 3) {
            const bf16x4 a4 = *(const bf16x4*)(p.aux + (long)m * p.ldaux + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)a4[e];
          }   // epi == 2 (ReLU mask) commutes with the bf16 rounding: applied below on full 16-B lines
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
        *(LDS_PTR(bf16x4))(smem + r * (BN * 2) + ((((cn >> 3) ^ (r & 15))) << 4) + ((cn >> 2) & 1) * 8) = o;
      }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
#pragma unroll
    for (int k = 0; k < BM * CPR / NTHR; ++k) {
      const int id = tid + k * NTHR;
      const int r = id / CPR, cc = id % CPR;
      const int m = m0 + r, n = n0 + cc * 8;
      if (m < p.M && n < p.N) {
        bf16x8 v = *(LDS_PTR(const bf16x8))(smem + r * (BN * 2) + ((cc ^ (r & 15)) << 4));
        if (p.epi == 2) {
          const bf16x8 a8 = *(const bf16x8*)(p.aux + (long)m * p.ldaux + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ((float)a8[e] > 0.f) ? v[e] : (bf16_t)0.f;
        }
        *(bf16x8*)((bf16_t*)p.C + (long)m * p.ldc + n) = v;
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

// ---------------------------------------------------------------------------------------------- host
struct TileCfg { int bm, bn, threads, lds; };
static const TileCfg kCfg[] = {
    {128, 128, 256, 2 * 32768},   // 0: 128x128, 2x2 waves, 2 stages  (2 blocks/CU)
    {128, 128, 256, 3 * 32768},   // 1: 128x128, 2x2 waves, 3 stages
    {256, 128, 512, 3 * 49152},   // 2: 256x128, 4x2 waves, 3 stages
    {256, 256, 512, 2 * 65536},   // 3: 256x256, 2x4 waves (128x64 per wave), 2 stages
    {256, 128, 512, 2 * 49152},   // 4: 256x128, 4x2 waves, 2 stages
    {128, 256, 512, 3 * 49152},   // 5: 128x256, 2x4 waves, 3 stages
    {256, 128, 512, 3 * 49152},   // 6: 256x128, 4x2 waves, 3 stages, staggered S/M phases
    {128, 256, 512, 3 * 49152},   // 7: 128x256, 2x4 waves, 3 stages, staggered S/M phases
    {64, 128, 256, 2 * 24576},    // 8: 64x128, 2x2 waves (32x64 per wave), 2 stages: small-M problems
};
static const int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);
static int g_force_cfg = -1;
extern "C" int hriemo_gemm_force_config(int cfg) {   // tuning hook (scripts_dev/bench_gemm.py); -1 = heuristic
  g_force_cfg = (cfg >= 0 && cfg < kNumCfg) ? cfg : -1;
  return 0;
}

template <int TA, int TB, int OUTF32, int BM, int BN, int WM, int WN, int NS, int STAG = 0>
static void launch_one(const GemmArgs& a, int lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_kernel<TA, TB, OUTF32, BM, BN, WM, WN, NS, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int grid = a.tiles_m * a.tiles_n * a.splitk;
  hipLaunchKernelGGL((gemm_kernel<TA, TB, OUTF32, BM, BN, WM, WN, NS, STAG>), dim3(grid), dim3(WM * WN * 64), lds, st, a);
}

template <int TA, int TB, int OUTF32>
static void launch_gemm(const GemmArgs& a, int cfg, hipStream_t st) {
  const int lds = kCfg[cfg].lds;
  switch (cfg) {
    case 0: launch_one<TA, TB, OUTF32, 128, 128, 2, 2, 2>(a, lds, st); break;
    case 1: launch_one<TA, TB, OUTF32, 128, 128, 2, 2, 3>(a, lds, st); break;
    case 2: launch_one<TA, TB, OUTF32, 256, 128, 4, 2, 3>(a, lds, st); break;
    case 3: launch_one<TA, TB, OUTF32, 256, 256, 2, 4, 2>(a, lds, st); break;
    case 4: launch_one<TA, TB, OUTF32, 256, 128, 4, 2, 2>(a, lds, st); break;
    case 5: launch_one<TA, TB, OUTF32, 128, 256, 2, 4, 3>(a, lds, st); break;
    case 6: launch_one<TA, TB, OUTF32, 256, 128, 4, 2, 3, 1>(a, lds, st); break;
    case 7: launch_one<TA, TB, OUTF32, 128, 256, 2, 4, 3, 1>(a, lds, st); break;
    default: launch_one<TA, TB, OUTF32, 64, 128, 2, 2, 2>(a, lds, st); break;
  }
}

// Tile choice from the measured sweep on MI355X (scripts_dev/bench_gemm.py, profiles/): the 256x256 tile
// halves LDS-DMA issues and fragment reads per MFMA and wins whenever it still yields >= ~1 block per CU;
// narrow outputs (N = d_model) keep the 128x128 tile at 2 blocks/CU.
static int pick_config(int ta, int tb, int M, int N, int K) {
  if (g_force_cfg >= 0) return g_force_cfg;
  if (ta == 1) return (M >= 512 && N >= 512 && (long)K >= 4096) ? 3 : 0;     // dW: split-K fills the chip
  if (M < 1024 || N < 256) return (M <= 512 && N >= 256) ? 8 : 0;             // decoder / gate sized problems
  if (tb == 0) return (N >= 2048 && M >= 16384) ? 3 : 7;                      // NT: 256x256, else staggered 128x256
  return M >= 16384 ? 3 : 4;                                                  // NN
}

extern "C" int hriemo_gemm_bf16(int ta, int tb, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                                void* C, long ldc, int c_is_f32, const float* bias, int epilogue, const void* aux,
                                long ldaux, int accumulate, float* workspace, long workspace_bytes, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  HRIEMO_CHECK(!(ta == 1 && tb == 0), "gemm: layout (ta=1,tb=0) is not used by this path and not built");
  HRIEMO_CHECK(N % 8 == 0, "gemm: N=%d must be a multiple of 8", N);
  HRIEMO_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ldc % (c_is_f32 ? 4 : 8) == 0, "gemm: leading dims must keep 16-byte alignment");
  HRIEMO_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0, "gemm: unaligned operand");
  HRIEMO_CHECK(lda < (1L << 21) && ldb < (1L << 21), "gemm: leading dimension too large for 32-bit tile offsets");
  if (ta == 0) HRIEMO_CHECK(K % 8 == 0, "gemm: K=%d must be a multiple of 8 for a K-contiguous operand", K);
  if (ta == 1) HRIEMO_CHECK(M % 8 == 0, "gemm: M=%d must be a multiple of 8 for a transposed A", M);
  HRIEMO_CHECK(c_is_f32 || (epilogue >= 0 && epilogue <= 3), "gemm: bad epilogue");
  HRIEMO_CHECK(epilogue < 2 || (aux != nullptr && ldaux % 8 == 0 && ((uintptr_t)aux % 16) == 0), "gemm: epilogue 2/3 needs a 16-byte aligned aux");
  HRIEMO_CHECK(!(c_is_f32 && epilogue != 0), "gemm: fp32 output has no activation epilogue");
  HRIEMO_CHECK(c_is_f32 || !accumulate, "gemm: accumulate needs fp32 output");

  int cfg = pick_config(ta, tb, M, N, K);
  if (cfg == 8 && ta == 1) cfg = 0;            // the 64-row tile has no K-strided A image (128-B rows cannot hold the swizzle)
  GemmArgs a;
  a.M = M; a.N = N; a.K = K;
  a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb;
  a.C = C; a.ldc = ldc; a.bias = bias; a.aux = (const bf16_t*)aux; a.ldaux = ldaux; a.epi = epilogue;
  a.tiles_m = (M + kCfg[cfg].bm - 1) / kCfg[cfg].bm; a.tiles_n = (N + kCfg[cfg].bn - 1) / kCfg[cfg].bn;
  a.ws = workspace; a.accumulate = accumulate;
  int splitk = 1;
  if (c_is_f32) {
    const long tiles = (long)a.tiles_m * a.tiles_n;
    const int ksteps = (K + 63) / 64;
    const long slots = 256L * (kCfg[cfg].lds <= 65536 ? 2 : 1);
    long want = (slots * 3 / 2 + tiles - 1) / tiles;   // ~1.5 rounds of resident blocks
    if (want > ksteps / 4) want = ksteps / 4;          // >= 4 K-steps (256 of K) per slice
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
    if (c_is_f32) launch_gemm<0, 0, 1>(a, cfg, st); else launch_gemm<0, 0, 0>(a, cfg, st);
  } else if (ta == 0 && tb == 1) {
    if (c_is_f32) launch_gemm<0, 1, 1>(a, cfg, st); else launch_gemm<0, 1, 0>(a, cfg, st);
  } else {
    if (c_is_f32) launch_gemm<1, 1, 1>(a, cfg, st); else launch_gemm<1, 1, 0>(a, cfg, st);
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
