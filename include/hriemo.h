/* hriemo.h -- C ABI of libhriemo.so: the MI355X (gfx950) kernels behind HRI-EMO's cross-modal fusion +
 * beta-gate + emotion-decoder forward/backward path.
 *
 * The reference (Makiato1999/HRI-EMO) has no FFI layer: every FLOP of this path is a stock torch.nn
 * call inside models/ (SURVEY.md section 8b).  Each entry point below therefore names the reference
 * nn.Module arithmetic it replaces (file:line under the reference tree); the Python mirror of the
 * reference's nn.Module API (hri-emo_amd/models/) binds these symbols with ctypes.
 *
 * Conventions
 *  - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller (outputs and
 *    workspaces included); nothing is allocated, freed or synchronised inside (graph-capturable);
 *  - activations are bf16 (uint16 storage) row-major with an explicit leading dimension in ELEMENTS;
 *    statistics, parameters' masters, gradients of parameters are fp32;
 *  - masks are uint8 [B, L], 1 = PAD (the reference's key_padding_mask convention);
 *  - dropout is replayed from (seed, site, row/col) -- the backward takes the same triple, no mask is
 *    stored; p_drop = 0 disables it; the effective seed is seed + *seed_dev (device word, may be NULL) so a
 *    captured hipGraph can draw fresh masks per replay by bumping that word inside the graph;  b_offset / row_offset = global index of the first utterance/row
 *    of this shard so masks do not depend on how the batch is sharded over GPUs;
 *  - `stream` is a hipStream_t; return 0 = ok, otherwise hriemo_last_error() explains.
 */
#ifndef HRIEMO_H_
#define HRIEMO_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hriemo_stream_t;

const char* hriemo_last_error(void);
int hriemo_abi_version(void);

/* ---- GEMM: every nn.Linear / packed in-projection / out-projection on the path and their backward.
 * C[M,N] = opA(A) . opB(B) (+bias) with fp32 accumulation on MFMA 16x16x32 bf16.
 *   ta=0: A is [M,K] (K contiguous)      ta=1: A is [K,M] (dW = dY^T . X)
 *   tb=0: B is [N,K] (weight layout)     tb=1: B is [K,N] (dX = dY . W)
 * c_is_f32: 0 -> bf16 C, 1 -> fp32 C (weight gradients; split-K through `workspace`).
 * epilogue (bf16 C only): 0 none, 1 ReLU, 2 multiply by (aux > 0) (ReLU backward), 3 add aux
 *   (fuses the residual-branch gradient into the dX GEMM).
 * Replaces: F.linear in nn.MultiheadAttention in-proj/out-proj (models/cross_modal_block_tacfn.py:24-40,
 * 74-117; models/emotion_decoder.py:14,20,42-54), ffn_a/ffn_t (cross_modal_block_tacfn.py:43-52,106,119),
 * linear1/linear2 (emotion_decoder.py:25-27,58), gate MLP (models/beta_gate_tacfn.py:62-66,92). */
int hriemo_gemm_bf16(int ta, int tb, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                     void* C, long ldc, int c_is_f32, const float* bias, int epilogue, const void* aux,
                     long ldaux, int accumulate, float* workspace, long workspace_bytes, hriemo_stream_t stream);

/* njobs independent weight-gradient GEMMs C_j[M_j, N_j] (+)= A_j[K_j, M_j]^T . B_j[K_j, N_j] (layout ta = 1, tb = 1 of
 * hriemo_gemm_bf16, fp32 results, no split-K) in ONE launch per 16 problems: jobs_host = HOST array of njobs x 9 int64
 * {M, N, K, A, lda, B, ldb, C, ldc}, passed through kernel arguments (capture-safe).  The decoder's and the gate's weight gradients
 * (models/emotion_decoder.py:14-27, models/beta_gate_tacfn.py:62-66: M = B*N_e or B reduction rows), 16 latency-bound launches
 * per step otherwise. */
int hriemo_gemm_bf16_group_tn(const void* jobs_host, int njobs, int accumulate, hriemo_stream_t stream);
/* hriemo_gemm_bf16 with an fp32 result written to TWO matrices: rows [0, split_m) of the [M, N] result to C, rows [split_m, M) to
 * C2 (from its row 0).  The weight gradient of a projection whose weight rows belong to two parameters -- one N = 3d GEMM per
 * shared input: rows [0, d) are the Q rows of one nn.MultiheadAttention.in_proj_weight, rows [d, 3d) the K | V rows of another
 * (models/cross_modal_block_tacfn.py:98-104,111-117) -- in one launch.  accumulate as in hriemo_gemm_bf16. */
int hriemo_gemm_bf16_split(int ta, int tb, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                           void* C2, long ldc2, int split_m, int accumulate, float* workspace, long workspace_bytes,
                           hriemo_stream_t stream);
/* C[M,N] (bf16) = (A . B) * (aux > 0) as hriemo_gemm_bf16 with epilogue 2, plus the column sums of the stored (masked, rounded)
 * C as per-row-block partials [hriemo_gemm_colsum_rows(ta,tb,M,N,K)][N] fp32 -- summed over rows (hriemo_colreduce_batch) they
 * are the bias gradient of the FIRST Linear of a feed-forward block (dh = (dy . W2) * relu'(h), db1 = colsum(dh);
 * cross_modal_block_tacfn.py:43-52,106,119) without a second pass over dh. */
int hriemo_gemm_colsum_rows(int ta, int tb, int M, int N, int K);
int hriemo_gemm_bf16_colsum(int ta, int tb, int M, int N, int K, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                            const void* aux, long ldaux, float* colsum_partials, hriemo_stream_t stream);
/* tuning hook: force one of the built tile configurations (-1 = built-in heuristic); 0-8: work-queue kernels (128x128 ... 32x64
 * tiles), 9: the loader / consumer kernel (256x128 tile, 8 MFMA waves + 4 waves that only stage operands; round 4, the default for
 * encoder-sized projections and long weight-gradient reductions) */
int hriemo_gemm_force_config(int cfg);
/* Tuning word of every following GEMM launch (default 9); returns the previous value.  Bit 0: in the 3-stage 256x128 kernel the
 * first K-step after an epilogue counts that epilogue's stores in its retire wait instead of waiting for their acknowledgement;
 * bit 1: never pick configuration 9; bit 3: configuration 9 (and the MX-fp8 kernel of the same form) walk their tiles statically instead of drawing
 * them from the per-XCD work queue -- faster on a chip the launch has to itself; hri_emo_amd.dp clears it while collectives run
 * beside backward (a block whose CU is held late then draws fewer tiles). */
int hriemo_gemm_debug_flags(int flags);

/* ---- MX-fp8 operand path (BASELINE.json configs[4], "fp8 MFMA path"): the same nn.Linear / in-projection / out-projection
 * sites as hriemo_gemm_bf16 in the FORWARD direction (models/cross_modal_block_tacfn.py:24-52, models/emotion_decoder.py:14-27),
 * with both operands as OCP e4m3 bytes + one E8M0 scale byte per 32 k-elements (OCP microscaling, block 32), multiplied on
 * v_mfma_scale_f32_16x16x128_f8f6f4 with fp32 accumulation.  Attention cores, backward GEMMs, masters and gradients keep
 * their bf16 / fp32 formats.
 *   hriemo_quant_mx8: X[M,K] (bf16, or fp32 when src_is_f32: weight masters) -> Xq[M,K] bytes (row stride ldq) and scales
 *     S[K/32][lds] (k-block major; lds = hriemo_mx8_scale_ld(M), a multiple of 256 >= M).  Block scale = 2^ceil(log2(amax/448)):
 *     nothing saturates; elements round to nearest even.
 *   hriemo_gemm_mx8: C[M,N] = A[M,K] . B[N,K]^T (+bias) with the epilogues of hriemo_gemm_bf16; K % 128 == 0. */
long hriemo_mx8_scale_ld(int rows);
int hriemo_quant_mx8(const void* X, long ldx, int src_is_f32, int M, int K, void* Xq, long ldq, void* S, long lds,
                     hriemo_stream_t stream);
int hriemo_gemm_mx8(int M, int N, int K, const void* Aq, long lda, const void* SA, long ldsa, const void* Bq, long ldb,
                    const void* SB, long ldsb, void* C, long ldc, int c_is_f32, const float* bias, int epilogue,
                    const void* aux, long ldaux, hriemo_stream_t stream);
/* hriemo_gemm_mx8 with bf16 output (epilogue 0 / 1) whose epilogue ALSO writes the MX-fp8 form of the values it stores: bytes
 * CQ[M][ldcq] + E8M0 scales SC[N/32][ldsc] (ldsc >= M, a multiple of 256), bit-identical to hriemo_quant_mx8 of C -- the operand
 * of the next GEMM (FFN1 -> FFN2, models/cross_modal_block_tacfn.py:43-52) without a quantisation pass of its own.  N % 32 == 0. */
int hriemo_gemm_mx8_q(int M, int N, int K, const void* Aq, long lda, const void* SA, long ldsa, const void* Bq, long ldb,
                      const void* SB, long ldsb, void* C, long ldc, const float* bias, int epilogue, void* CQ, long ldcq,
                      void* SC, long ldsc, hriemo_stream_t stream);
int hriemo_gemm_mx8_force_config(int cfg);

/* ---- attention core: softmax(QK^T/sqrt(hd) + mask) -> dropout -> .V per (batch, head), flash style.
 * Q/K/V/O/dX are read/written in place inside the projection buffers: element (b, l, h, e) of X is
 * X[(b*L + l)*ldx + h*head_dim + e].  lse [B,H,Lq] (natural log) is written by fwd, read by bwd/probs;
 * delta [B,H,Lq] is bwd scratch.  Replaces the body of nn.MultiheadAttention.forward between the
 * projections (cross_modal_block_tacfn.py:74-80,85-91,98-104,111-117; emotion_decoder.py:42,48-54). */
int hriemo_attn_fwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O, long ldo,
                    const unsigned char* key_padding_mask, float* lse, int B, int H, int Lq, int Lk, int head_dim,
                    float p_drop, unsigned long long seed, const unsigned long long* seed_dev, unsigned site, int b_offset,
                    void* drop_mask_bits, hriemo_stream_t stream);
int hriemo_attn_bwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, const void* O, long ldo,
                    const void* dO, long lddo, void* dQ, long lddq, void* dK, long lddk, void* dV, long lddv,
                    const unsigned char* key_padding_mask, const float* lse, float* delta, int B, int H, int Lq,
                    int Lk, int head_dim, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                    unsigned site, int b_offset, float* dq_colsum_partials, float* dkv_colsum_partials,
                    const void* drop_mask_bits, hriemo_stream_t stream);
/* drop_mask_bits (optional, hriemo_attn_mask_bytes(B,H,Lq,Lk) bytes, 8-byte aligned): the dropout keep-mask as bits, one
 * 64-bit word per (batch, head, query, 64-key tile), written by hriemo_attn_fwd when p_drop > 0 and read by hriemo_attn_bwd
 * instead of replaying the hash (NULL on either side: the hash is replayed; both give the same mask).
 * For 16 < Lk <= 128 hriemo_attn_bwd is ONE kernel (dQ, dK, dV together; HRIEMO_ATTN_FUSED_BWD=0 in the environment selects
 * the two-kernel path). */
long hriemo_attn_mask_bytes(int B, int H, int Lq, int Lk);
/* 1 when hriemo_attn_bwd runs as the single kernel for this problem (the bit words pay there; the two-kernel path is as fast
 * replaying the hash) -- what the host side asks before it requests drop_mask_bits from the forward */
int hriemo_attn_bwd_single_pass(int B, int H, int Lk, int head_dim);
/* Both lengths known: 1 if the backward of this shape is ONE kernel -- the key-resident form above (16 < L_k <= 128) or the
 * query-resident one (16 < L_q <= 128 < L_k: all queries of a (batch, head) in one block, dQ in registers, complete dK / dV tiles
 * per key tile) -- and the rows of the dK | dV column-sum partials hriemo_attn_bwd then leaves behind. */
int hriemo_attn_bwd_single_pass_q(int B, int H, int Lq, int Lk, int head_dim);
int hriemo_attn_bwd_kv_colsum_rows(int B, int H, int Lq, int Lk, int head_dim);
/* Packed (varlen) sequences, SURVEY 8(f) rank 4: the reference pads every sample to the batch maximum
 * (scripts/fusion/train_fusion_seq_level_decoder.py:191-232) and computes the PAD rows; here Q / O / dO / dQ hold the valid rows of
 * all samples back to back (sample b = rows cu_seqlens_q[b] .. cu_seqlens_q[b+1]-1) and K / V / dK / dV likewise with
 * cu_seqlens_k (int32 [B+1], device memory, every length >= 1).  max_len_q / max_len_k are the longest sequences: they size the
 * grid and are the row stride of lse / delta / drop_mask_bits and of the column-sum partials, which keep the padded indexing of
 * the functions above (so hriemo_attn_mask_bytes / *_colsum_rows are asked with the maxima).  Same arithmetic, same dropout
 * mask (it is keyed by position within the sequence), results on the valid rows identical to the padded call with a
 * key_padding_mask. */
/* hriemo_attn_fwd whose epilogue also writes the MX-fp8 form of O -- bytes Oq[B*Lq][ldoq], E8M0 scales So[H*hd/32][ldso] (ldso >=
 * B*Lq, a multiple of 256), bit-identical to hriemo_quant_mx8 of O -- for the out-projection GEMM of the fp8 mode.  head_dim % 32 == 0. */
int hriemo_attn_fwd_q(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O, long ldo,
                      const unsigned char* key_padding_mask, float* lse, int B, int H, int Lq, int Lk, int head_dim, float p_drop,
                      unsigned long long seed, const unsigned long long* seed_dev, unsigned site, int b_offset, void* drop_mask_bits,
                      void* Oq, long ldoq, void* So, long ldso, hriemo_stream_t stream);
int hriemo_attn_fwd_varlen(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O, long ldo,
                           const int* cu_seqlens_q, const int* cu_seqlens_k, float* lse, int B, int H, int max_len_q,
                           int max_len_k, int head_dim, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                           unsigned site, int b_offset, void* drop_mask_bits, hriemo_stream_t stream);
int hriemo_attn_bwd_varlen(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, const void* O, long ldo,
                           const void* dO, long lddo, void* dQ, long lddq, void* dK, long lddk, void* dV, long lddv,
                           const int* cu_seqlens_q, const int* cu_seqlens_k, const float* lse, float* delta, int B, int H,
                           int max_len_q, int max_len_k, int head_dim, float p_drop, unsigned long long seed,
                           const unsigned long long* seed_dev, unsigned site, int b_offset, float* dq_colsum_partials,
                           float* dkv_colsum_partials, const void* drop_mask_bits, hriemo_stream_t stream);
/* Optional by-product of hriemo_attn_bwd (either pointer may be NULL): per-block column sums of the dQ tiles,
 * [hriemo_attn_bwd_dq_colsum_rows(B, H, Lq, Lk, head_dim), H*head_dim] fp32, and of the dK | dV tiles,
 * [hriemo_attn_bwd_colsum_rows(B, H, Lk, head_dim), 2*H*head_dim] fp32 (values before their bf16 rounding).  Summed over rows
 * (hriemo_colreduce_batch) they are the gradient of the packed in-projection bias (in_proj_bias of nn.MultiheadAttention,
 * cross_modal_block_tacfn.py:24-40) without re-reading dQ/dK/dV. */
int hriemo_attn_bwd_colsum_rows(int B, int H, int L, int head_dim);
int hriemo_attn_bwd_dq_colsum_rows(int B, int H, int Lq, int Lk, int head_dim);
/* head-averaged attention probabilities [B,Lq,Lk] fp32 (need_weights=True; return_attention path,
 * cross_modal_block_tacfn.py:70-125, emotion_decoder.py:48-64) */
int hriemo_attn_probs(const void* Q, long ldq, const void* K, long ldk, const unsigned char* key_padding_mask,
                      const float* lse, float* probs, int B, int H, int Lq, int Lk, int head_dim, float p_drop,
                      unsigned long long seed, const unsigned long long* seed_dev, unsigned site, int b_offset,
                      hriemo_stream_t stream);

/* ---- y = LayerNorm(x + dropout(g)), eps, affine (X may be NULL: plain LayerNorm of g).
 * The residual stream has an optional fp32 twin: X32 (read instead of the bf16 X when non-NULL) and Y32
 * (written next to the bf16 Y when non-NULL).  bf16 rounding of the LayerNorm outputs is ~85 % of the
 * path's end-to-end error (scripts_dev/precision_study.py); GEMM operands stay bf16.
 * Replaces norm(h + self.dropout(sub(h))) (cross_modal_block_tacfn.py:81,92,105,106,118,119;
 * emotion_decoder.py:43,55,59).  bwd writes dX (residual branch), dG (sub-layer branch, dropout mask
 * applied) and the column sums dgamma, dbeta, dbias (= colsum dG, the producing Linear's bias grad);
 * accumulate=1 adds them into the destination (fused accumulation into existing .grad buffers). */
int hriemo_add_ln_fwd(const void* G, const void* X, const float* X32, const float* gamma, const float* beta, void* Y,
                      float* Y32, float* mean, float* rstd, int M, int d, float eps, float p_drop, unsigned long long seed,
                      const unsigned long long* seed_dev, unsigned site, long row_offset, hriemo_stream_t stream);
/* same, plus the MX-fp8 copy of Y (e4m3 bytes [M][d], E8M0 scales [d/32][ldsy], d % 32 == 0): the quantiser of hriemo_quant_mx8
 * fused into the LayerNorm that produces the next GEMM's operand (bit-identical to quantising the bf16 Y afterwards) */
int hriemo_add_ln_fwd_mx8(const void* G, const void* X, const float* X32, const float* gamma, const float* beta, void* Y,
                          float* Y32, float* mean, float* rstd, int M, int d, float eps, float p_drop, unsigned long long seed,
                          const unsigned long long* seed_dev, unsigned site, long row_offset, void* Yq, void* SY, long ldsy,
                          hriemo_stream_t stream);
/* hriemo_add_ln_fwd / _bwd for rows gathered from a larger layout (packed varlen sequences): the dropout hash is keyed by
 * row_index[row] (int64 [M], e.g. the row of the padded [B*L] layout) instead of row, so the packed launch drops exactly the
 * elements the padded one drops.  Yq / SY (MX-fp8 copy) may be NULL. */
int hriemo_add_ln_fwd_rows(const void* G, const void* X, const float* X32, const float* gamma, const float* beta, void* Y,
                           float* Y32, float* mean, float* rstd, int M, int d, float eps, float p_drop, unsigned long long seed,
                           const unsigned long long* seed_dev, unsigned site, long row_offset, void* Yq, void* SY, long ldsy,
                           const long long* row_index, hriemo_stream_t stream);
int hriemo_add_ln_bwd_rows(const void* dY, const void* G, const void* X, const float* X32, const float* gamma, const float* mean,
                           const float* rstd, void* dX, void* dG, float* dgamma, float* dbeta, float* dbias, int accumulate,
                           int M, int d, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                           unsigned site, long row_offset, float* workspace, const long long* row_index, hriemo_stream_t stream);
/* Packed (varlen) sequences <-> padded layout, lengths read from DEVICE memory (one captured graph serves every batch whose valid
 * rows fit n_rows; replaces the reference's pad-to-the-batch-maximum, train_fusion_seq_level_decoder.py:191-232).  cu_seqlens:
 * int32 [B+1], sequence b = packed rows cu[b]..cu[b+1]-1 = its first positions of the padded [B, L, d] layout.  pack: packed rows
 * cu[B]..n_rows-1 (bucket padding) are written as zeros; row_index[r] (int64 [n_rows], may be NULL) = padded row of packed row r
 * (the row key of hriemo_add_ln_*_rows).  unpack: PAD positions are written as zeros.  X16/P16 (bf16) and X32/P32 (fp32 twin):
 * either pair may be NULL. */
int hriemo_pack_rows(const void* X16, const float* X32, const int* cu_seqlens, int B, int L, int d, int n_rows, void* P16, float* P32,
                     long long* row_index, hriemo_stream_t stream);
int hriemo_unpack_rows(const void* P16, const float* P32, const int* cu_seqlens, int B, int L, int d, void* Y16, float* Y32,
                       hriemo_stream_t stream);
long hriemo_add_ln_bwd_workspace_bytes(int M, int d);
int hriemo_add_ln_bwd(const void* dY, const void* G, const void* X, const float* X32, const float* gamma, const float* mean,
                      const float* rstd, void* dX, void* dG, float* dgamma, float* dbeta, float* dbias, int accumulate,
                      int M, int d, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                      unsigned site, long row_offset, float* workspace, hriemo_stream_t stream);

/* ---- small glue on the path */
long hriemo_colsum_workspace_bytes(int M, int N);
int hriemo_colsum_bf16(const void* X, long ldx, int M, int N, float* out, int accumulate, float* workspace,
                       hriemo_stream_t stream);                                   /* bias gradients */
int hriemo_cast_f32_to_bf16(const float* src, void* dst, long n, hriemo_stream_t stream);   /* bf16 shadows */
int hriemo_cast_bf16_to_f32(const void* src, float* dst, long n, hriemo_stream_t stream);
/* njobs casts in one launch (64 per launch): jobs_host = HOST array of njobs x 3 int64 {src fp32, dst bf16, n elements}, 16-byte
 * aligned pointers; the table is passed through kernel arguments (capture-safe).  The bf16 shadows of all weights of a step. */
int hriemo_cast_f32_to_bf16_batch(const void* jobs_host, int njobs, hriemo_stream_t stream);
/* the same with a kind per job: jobs_host = njobs x 4 int64 {src fp32, dst, n elements, kind}; kind 0 = fp32 -> bf16 (dst bf16),
 * 1 = fp32 -> fp32 copy.  One launch refreshes a shared projection's concatenated weight shadow and bias vector
 * (nn.MultiheadAttention.in_proj_weight / in_proj_bias slices of two modules, cross_modal_block_tacfn.py:98-117). */
int hriemo_cast_copy_batch(const void* jobs_host, int njobs, hriemo_stream_t stream);
int hriemo_dropout_bf16(const void* X, void* Y, long M, int N, float p_drop, unsigned long long seed,
                        const unsigned long long* seed_dev, unsigned site, long row_offset, hriemo_stream_t stream);                  /* emotion_decoder.py:58 */
/* emotion_decoder.py:127: out[b] = q for b < B, as bf16 and (out32 != NULL) as the fp32 twin of the residual stream */
int hriemo_expand_rows(const float* q, void* out, float* out32, int B, long n, hriemo_stream_t stream);
/* the device-resident dropout seed word of a captured step += 0x9E3779B97F4A7C15 (one launch per replay, inside the graph) */
int hriemo_seed_bump(unsigned long long* seed_dev, hriemo_stream_t stream);
int hriemo_rowdot_fwd(const void* Z, const float* Z32, const float* w, const float* b, float* out, int M, int d,
                      hriemo_stream_t stream);                                     /* emotion_decoder.py:155 */
/* dw[d], db[1]: overwritten, or added to when accumulate != 0 (gradients accumulated straight into .grad) */
int hriemo_rowdot_bwd(const float* dl, const void* Z, const float* Z32, const float* w, void* dZ, float* dw, float* db, int accumulate,
                      int M, int d, hriemo_stream_t stream);

/* ---- beta gate (models/beta_gate_tacfn.py:68-118): LayerNorm + masked mean-pool (:6-24,79-84),
 * gate input [a,t,|a-t|,a*t] (:87-89), w = sigmoid(MLP), beta = mean(w) (:92-95), fuse over the first
 * L positions (:98-116).  partials buffers are [B, hriemo_pool_chunks(L), d] fp32. */
int hriemo_pool_chunks(int L);
int hriemo_ln_pool_fwd(const void* X, const float* X32, const unsigned char* mask, const float* gamma, const float* beta, void* Yn,
                       float* mean, float* rstd, float* partials, int B, int L, int Lkeep, int d, float eps,
                       hriemo_stream_t stream);
int hriemo_gate_input(const float* partials_a, const float* partials_t, const unsigned char* mask_a,
                      const unsigned char* mask_t, int B, int La, int Lt, int d, void* gate_in, float* a_pool,
                      float* t_pool, float* cnt, hriemo_stream_t stream);
int hriemo_sigmoid_beta(const float* pre, float* w, float* beta, int B, int d, hriemo_stream_t stream);
int hriemo_fuse_fwd(const float* w, const void* A, const void* T, void* H, int B, int L, int d, hriemo_stream_t stream);
int hriemo_fuse_bwd_dw(const void* dH, const void* A, const void* T, float* partials, int B, int L, int d,
                       hriemo_stream_t stream);
int hriemo_gate_dpre(const float* partials, int L, const float* dbeta, const float* w, void* dpre, int B, int d,
                     hriemo_stream_t stream);
int hriemo_gate_input_bwd(const void* dgin, const float* a_pool, const float* t_pool, const float* cnt, float* da,
                          float* dt, int B, int d, hriemo_stream_t stream);
/* Legacy scalar gate, models/beta_gate.py:6-32,60-114 (what the reference's tests/test_beta_gate.py exercises):
 * masked mean pooling of the raw [B,L,d] features (pooled[B,d], cnt[B] = max(#valid,1)); the [B,4d]->h->1 MLP and
 * its sigmoid are [B]-sized host-side plumbing; h_fusion = beta*h_a[:, :L] + (1-beta)*h_t[:, :L] reuses
 * hriemo_fuse_fwd with beta broadcast over d, hriemo_fuse_bwd_dw + hriemo_rowsum_f32 give dbeta, and
 * hriemo_scalar_gate_dx writes dX = coef*dH (first Lf rows) + valid*dpool/cnt in one pass. */
int hriemo_masked_mean_fwd(const void* X, const unsigned char* mask, float* pooled, float* cnt, int B, int L, int d,
                           hriemo_stream_t stream);
int hriemo_rowsum_f32(const float* x, float* out, int B, long n, hriemo_stream_t stream);
/* gate input [a, t, |a-t|, a*t] (bf16 [B,4d]) from already pooled means: the legacy gate's MLP then runs on hriemo_gemm_bf16 +
 * hriemo_rowdot_* like the vector gate's (models/beta_gate.py:82-93) */
int hriemo_gate_input_pooled(const float* a_pool, const float* t_pool, void* gate_in, int B, int d, hriemo_stream_t stream);
/* Trainer losses, value and gradients in one launch ([B,N_e]-sized): mean BCEWithLogits(logits, targets; pos_weight (may be NULL))
 * + reg(beta) (reg_mode 0 none; 1: -coef*mean(beta(1-beta)), scripts/fusion/train_fusion_seq_level_decoder.py:318-326; 2: +coef*mean
 * binary entropy of clamp(beta,1e-8,1-1e-8), train_mosei_fusion_seq_level_decoder.py:340-347,385-386,569), everything times `scale`
 * (1/grad_accum).  Writes loss[1], dlogits[B,N_e], dbeta[B] (dbeta may be NULL). */
int hriemo_fusion_loss(const float* logits, const float* targets, const float* pos_weight, const float* beta, int B, int Ne,
                       int reg_mode, float reg_coef, float scale, float* loss, float* dlogits, float* dbeta, hriemo_stream_t stream);
/* Single-label variant, torch.nn.CrossEntropyLoss() as scripts/fusion/train_fusion_seq_level_decoder.py:413-414 builds it (mean
 * over the labelled samples, no class weights, no label smoothing, ignore_index = -100), plus the same beta regulariser (:325-326,
 * a mean over all B): labels[B] are int64 class indices in [0, C), or -100 for a sample the criterion skips (no loss term, zero
 * gradient, not counted in the mean; all of them skipped: NaN, as torch); any other index outside [0, C) poisons the loss with NaN
 * instead of reading out of bounds.  Writes loss[1], dlogits[B,C] = (softmax - onehot)/n_labelled * scale, dbeta[B] (may be NULL). */
int hriemo_fusion_loss_ce(const float* logits, const long long* labels, const float* beta, int B, int C, int reg_mode,
                          float reg_coef, float scale, float* loss, float* dlogits, float* dbeta, hriemo_stream_t stream);
int hriemo_scalar_gate_dx(const void* dH, int Lf, const float* beta, int is_a, const float* dpool, const float* cnt,
                          const unsigned char* mask, void* dX, int B, int L, int d, hriemo_stream_t stream);
/* Trainer step off the timed path, scripts/fusion/train_fusion_seq_level_decoder.py:332-334: clip_grad_norm_(5.0) +
 * AdamW(lr 1e-4, weight_decay 1e-2) as two passes over flat fp32 buffers that share one layout (parameters,
 * gradients, first and second moments; hri-emo_amd/optim.py lays them out): hriemo_sumsq_f32 writes nblocks partial
 * sums of squares (reduce them with hriemo_rowsum_f32), hriemo_adamw_flat reads the squared norm from device memory
 * (coef = min(1, max_norm/(norm+1e-6)); max_norm <= 0: no clipping) and applies torch.optim.AdamW's update. */
int hriemo_sumsq_f32(const float* x, long n, float* partial, int nblocks, hriemo_stream_t stream);
int hriemo_adamw_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int step, float max_norm, const float* norm2, hriemo_stream_t stream);
/* Launch-boundary reduce.  hriemo_add_ln_bwd with dgamma == NULL and hriemo_colsum_bf16 with out == NULL stop after
 * their per-block partial sums ([hriemo_add_ln_bwd_partial_rows(M,d)][3*d] resp. [hriemo_colsum_partial_rows(M,N)][N]
 * fp32 at the start of the caller's workspace); hriemo_colreduce_batch finishes any number of such jobs in one launch
 * at the end of backward.  jobs_host: HOST array of njobs records (copied into the caller's device buffer jobs_dev,
 * njobs*64 bytes, through kernel arguments: capture-safe) of 8 x int64 = {partials, pstride (floats), np, w,
 * nseg | accumulate << 8 | first_block << 32, out0, out1, out2} with first_block the running sum of
 * nseg * ceil(w/32) over the preceding jobs and nblocks the total.  Deterministic (fixed summation order). */
int hriemo_add_ln_bwd_partial_rows(int M, int d);
/* Tuning / test hook: 1 (default) = LayerNorm(x + dropout(g)) forward and backward run the chunk-mapped kernels; 0 = the quad-mapped,
 * software-pipelined kernels where they are built (d = 256, 512, 768 or 1024, fp32 twin in and out, no MX copy).  Both mappings
 * draw the same dropout masks and differ by the summation order of the row statistics only.  Set it before any workspace is sized
 * (hriemo_add_ln_bwd_workspace_bytes / _partial_rows follow the variant). */
int hriemo_rowops_force_variant(int variant);
int hriemo_colsum_partial_rows(int M, int N);
int hriemo_colreduce_batch(const void* jobs_host, int njobs, void* jobs_dev, int nblocks, hriemo_stream_t stream);
/* tuning hook: `blocks` CUs made unavailable for ~`micros` us (stands in for a collective running beside the step) */
int hriemo_debug_hog(int blocks, int micros, float* sink, hriemo_stream_t stream);
/* The gate's two modalities from ONE launch (d <= 1024: hriemo_ln_pool_pair_supported): side a = audio (gate coefficient w), side t =
 * text (1 - w); the blocks are those of the single calls, so are the results.  Replaces two launches on two streams and the fork /
 * join around them (models/beta_gate_tacfn.py:79-84 forward, its autograd backward). */
int hriemo_ln_pool_pair_supported(int d);
int hriemo_ln_pool_fwd_pair(const void* Xa, const float* Xa32, const unsigned char* mask_a, const float* gamma_a, const float* beta_a,
                            void* Yna, float* mean_a, float* rstd_a, float* partials_a, int La,
                            const void* Xt, const float* Xt32, const unsigned char* mask_t, const float* gamma_t, const float* beta_t,
                            void* Ynt, float* mean_t, float* rstd_t, float* partials_t, int Lt,
                            int B, int Lkeep, int d, float eps, hriemo_stream_t stream);
int hriemo_ln_pool_bwd_pair(const void* dH, int Lf, const float* w,
                            const float* dpool_a, const unsigned char* mask_a, const void* Xa, const float* Xa32, const float* gamma_a,
                            const float* mean_a, const float* rstd_a, void* dXa, float* dgamma_a, float* dbeta_a, int La, float* workspace_a,
                            const float* dpool_t, const unsigned char* mask_t, const void* Xt, const float* Xt32, const float* gamma_t,
                            const float* mean_t, const float* rstd_t, void* dXt, float* dgamma_t, float* dbeta_t, int Lt, float* workspace_t,
                            int accumulate, int B, int d, hriemo_stream_t stream);
/* row chunks per sample of hriemo_ln_pool_bwd: its partial sums are [B * chunks][2 d] floats at the head of `workspace` */
int hriemo_ln_pool_bwd_chunks(int L);
long hriemo_ln_pool_bwd_workspace_bytes(int B, int L, int d);
int hriemo_ln_pool_bwd(const void* dH, int Lf, const float* w, int is_a, const float* dpool, const unsigned char* mask,
                       const void* X, const float* X32, const float* gamma, const float* mean, const float* rstd, void* dX,
                       float* dgamma, float* dbeta, int accumulate, int B, int L, int d, float* workspace, hriemo_stream_t stream);
/* (dgamma / dbeta: overwritten, or added to when accumulate != 0; both NULL: the per-block partial sums stay in `workspace`, rows
 * [B * ceil(L/32)] of [dgamma | dbeta] (2d floats each), for hriemo_colreduce_batch at the end of backward) */

/* ---- fp32-tolerance mode, forward (csrc/fp32mode.hip; host side hri-emo_amd/_fp32.py, HRIEMO_PRECISION=fp32).
 * The reference's modules are fp32 nn.Modules throughout (models/cross_modal_block_tacfn.py:70-125, beta_gate_tacfn.py:68-118,
 * emotion_decoder.py:30-64,116-162); this mode reproduces them to 1e-3 (no dropout; the backward follows below):
 *  - hriemo_split_bf16x3: X fp32 [M,K] (row stride ldx) -> Y bf16 [M,3K], x = hi + mid + ..: layout 0 (activations) [hi|mid|hi],
 *    layout 1 (weights) [hi|hi|mid]; relu != 0 applies max(x,0) first.  hriemo_gemm_bf16 over the 3K-long contraction with fp32
 *    output then gives hi.hi + mid.hi + hi.mid, the fp32 product to 2^-16 relative (nn.Linear, nn.MultiheadAttention in/out-proj);
 *  - hriemo_attn_fwd_f32 / hriemo_attn_probs_f32: softmax(Q K^T / sqrt(hd) + key padding) V on the fp32 MFMA
 *    (v_mfma_f32_16x16x4_f32), operands and outputs fp32 with row strides; lse = log-sum-exp of the scaled scores per
 *    (batch, head, query), probs = head-averaged probabilities [B,Lq,Lk] (need_weights=True);
 *  - hriemo_add_ln_f32: Y32 (and the bf16 copy Y16 when non-NULL) = LayerNorm(drop(G) + X) (X may be NULL), fp32 in and out;
 *  - dropout (round 4): every fp32 kernel that sits where the reference drops takes (p_drop, seed, seed_dev, site, offset) like its
 *    bf16 counterpart and draws the SAME mask from the same counter hash (hriemo_add_ln_fwd / hriemo_attn_fwd / hriemo_dropout_bf16):
 *    attention weights after the softmax (the exported probabilities too, as nn.MultiheadAttention returns them in training mode),
 *    the sub-layer output before the residual add, and hriemo_dropout_f32 for the FFN's hidden layer: Y = drop(relu ? max(X,0) : X)
 *    [* (gate > 0) when gate is non-NULL: the backward form, X = gradient, gate = pre-activations]; p_drop = 0 drops nothing;
 *  - gate pieces of models/beta_gate_tacfn.py in fp32: masked mean (:6-24), gate input [a,t,|a-t|,a*t] (:87-89),
 *    w = sigmoid(pre) / beta = mean(w) (:92-95), h = w*a + (1-w)*t over the first L positions (:98-116; A, T are [B,La,d], [B,Lt,d]). */
int hriemo_split_bf16x3(const float* X, long ldx, int M, int K, void* Y, int layout, int relu, hriemo_stream_t stream);
int hriemo_attn_fwd_f32(const float* Q, long ldq, const float* K, long ldk, const float* V, long ldv, float* O, long ldo,
                        const unsigned char* key_padding_mask, float* lse, int B, int H, int Lq, int Lk, int head_dim,
                        float p_drop, unsigned long long seed, const unsigned long long* seed_dev, unsigned site, int b_offset,
                        hriemo_stream_t stream);
int hriemo_attn_probs_f32(const float* Q, long ldq, const float* K, long ldk, const unsigned char* key_padding_mask,
                          const float* lse, float* probs, int B, int H, int Lq, int Lk, int head_dim, float p_drop,
                          unsigned long long seed, const unsigned long long* seed_dev, unsigned site, int b_offset,
                          hriemo_stream_t stream);
int hriemo_add_ln_f32(const float* G, const float* X, const float* gamma, const float* beta, float* Y32, void* Y16, int M, int d,
                      float eps, float p_drop, unsigned long long seed, const unsigned long long* seed_dev, unsigned site,
                      long row_offset, hriemo_stream_t stream);
int hriemo_dropout_f32(const float* X, float* Y, long M, int N, int relu, const float* gate, float p_drop, unsigned long long seed,
                       const unsigned long long* seed_dev, unsigned site, long row_offset, hriemo_stream_t stream);
int hriemo_masked_mean_f32(const float* X, const unsigned char* mask, float* pooled, int B, int L, int d, hriemo_stream_t stream);
int hriemo_gate_input_f32(const float* a_pool, const float* t_pool, float* gate_in, int B, int d, hriemo_stream_t stream);
int hriemo_sigmoid_beta_f32(const float* pre, float* w, float* beta, int B, int d, hriemo_stream_t stream);
int hriemo_fuse_f32(const float* w, const float* A, int La, const float* T, int Lt, float* H32, void* H16, int B, int L, int d,
                    hriemo_stream_t stream);

/* ---- fp32-tolerance TRAINING step: the backward of the pieces above in the same arithmetic (round 4).  The IEMOCAP trainer runs
 * the reference in fp32 without autocast (scripts/fusion/train_fusion_seq_level_decoder.py:310-334); with HRIEMO_PRECISION=fp32 the
 * modules of hri-emo_amd/models keep every activation and every gradient in fp32 (host side hri-emo_amd/_fp32.py).  Dropout 0 only.
 *  - hriemo_split3_f32: the operand split in four forms.  X fp32 [M,K] (row stride ldx), optionally times (mask > 0) (mask fp32
 *    [M, ldmask]: ReLU's derivative) and / or clamped at 0 (relu) -> bf16
 *      form 0: Y[M][3K] = [hi | mid | hi]      form 1: Y[M][3K] = [hi | hi | mid]       contraction along the columns of X
 *      form 2: Y[3M][K] = [hi ; mid ; hi]      form 3: Y[3M][K] = [hi ; hi ; mid]       contraction along the rows of X
 *      form 4: Y[M][6K] = [hi|mid|lo|hi|mid|hi]  form 5: Y[M][6K] = [hi|hi|hi|mid|mid|lo]  (x = hi + mid + lo exactly: six products,
 *      form 6: Y[6M][K] = [hi;mid;lo;hi;mid;hi]  form 7: Y[6M][K] = [hi;hi;hi;mid;mid;lo]   2^-24 relative; what _fp32.py uses: no
 *              ReLU pre-activation changes sign against fp32, ill-conditioned weights keep every gradient within 1e-3)
 *    dX = dY . W is hriemo_gemm_bf16(ta 0, tb 1) on form 0 (4) of dY and form 3 (7) of W (K = 3 (6) N_out); dW = dY^T . X is
 *    (ta 1, tb 1) on form 2 (6) of dY and form 3 (7) of X (K = 3 (6) M); fp32 output.
 *  - hriemo_colsum_f32: out[n] (+)= sum_m X[m][n] * (mask[m][n] > 0) (bias gradients; mask fp32 [M, ldmask] or NULL), fixed
 *    summation order; workspace >= hriemo_colsum_f32_workspace_bytes.
 *  - hriemo_add_ln_bwd_f32: backward of Y = LayerNorm(drop(G) + X) * gamma + beta (X may be NULL): dS = d loss / d (drop(G) + X)
 *    [M,d] (= dX; = dG as well when p_drop = 0, dG may then be NULL; with dropout dG = dS * keep / (1 - p) is written to its own
 *    matrix), dgamma / dbeta / dbias (= column sums of dG; may be NULL) overwritten or added to (accumulate); the row statistics
 *    are recomputed from drop(G) + X.  d <= 1024.  workspace >= hriemo_add_ln_bwd_f32_workspace_bytes(M, d).
 *  - hriemo_attn_bwd_f32: dQ, dK, dV of O = softmax(Q K^T / sqrt(hd) + key padding) V from Q, K, V, O, dO and the forward's lse, on
 *    v_mfma_f32_16x16x4_f32; two kernels (dQ per 64 queries, dK / dV per 64 keys), no atomics, fixed order; delta: scratch
 *    [B, H, Lq] floats (rowsum(dO * O), written by the first kernel for the second).  nn.MultiheadAttention, cross_modal_block_
 *    tacfn.py:74-80,98-117, emotion_decoder.py:42,48-54.
 *  - gate backward (beta_gate_tacfn.py:79-116): hriemo_gate_dpre_f32: dpre[B,d] = (sum_{l<L} dH (A - T) + dbeta / d) w (1 - w)
 *    (dbeta [B] may be NULL); hriemo_gate_input_bwd_f32: gradients of the pooled means from d [a, t, |a-t|, a*t];
 *    hriemo_gate_dy_f32: the gradient that reaches LayerNorm_x's output of one modality, dY[B,Lx,d] = (l < L ? wsel dH : 0) +
 *    (valid ? dpool / max(#valid, 1) : 0), wsel = w (is_a) or 1 - w; the LayerNorm itself goes through hriemo_add_ln_bwd_f32.
 *  - hriemo_rowdot_bwd_f32: logits = z . w + b (emotion_decoder.py:155): dZ[M,d] = dl w, dw[d] (+)= sum_m dl z, db[1] (+)= sum dl. */
int hriemo_split3_f32(const float* X, long ldx, int M, int K, void* Y, int form, int relu, const float* mask, long ldmask,
                      hriemo_stream_t stream);
long hriemo_colsum_f32_workspace_bytes(int M, int N);
int hriemo_colsum_f32(const float* X, long ldx, int M, int N, const float* mask, long ldmask, float* out, int accumulate,
                      float* workspace, hriemo_stream_t stream);
long hriemo_add_ln_bwd_f32_workspace_bytes(int M, int d);
int hriemo_add_ln_bwd_f32(const float* dY, const float* G, const float* X, const float* gamma, float* dS, float* dG, float* dgamma,
                          float* dbeta, float* dbias, int accumulate, int M, int d, float eps, float p_drop, unsigned long long seed,
                          const unsigned long long* seed_dev, unsigned site, long row_offset, float* workspace, hriemo_stream_t stream);
int hriemo_attn_bwd_f32(const float* Q, long ldq, const float* K, long ldk, const float* V, long ldv, const float* O, long ldo,
                        const float* dO, long lddo, const unsigned char* key_padding_mask, const float* lse, float* dQ, long lddq,
                        float* dK, long lddk, float* dV, long lddv, float* delta, int B, int H, int Lq, int Lk, int head_dim,
                        float p_drop, unsigned long long seed, const unsigned long long* seed_dev, unsigned site, int b_offset,
                        hriemo_stream_t stream);
int hriemo_gate_dpre_f32(const float* dH, const float* A, int La, const float* T, int Lt, const float* w, const float* dbeta, float* dpre,
                         int B, int L, int d, hriemo_stream_t stream);
int hriemo_gate_input_bwd_f32(const float* dgin, const float* a_pool, const float* t_pool, float* da, float* dt, int B, int d,
                              hriemo_stream_t stream);
int hriemo_gate_dy_f32(const float* dH, const float* w, int is_a, const float* dpool, const unsigned char* mask, float* dY, int B, int L,
                       int Lx, int d, hriemo_stream_t stream);
int hriemo_rowdot_bwd_f32(const float* dl, const float* Z, const float* w, float* dZ, float* dw, float* db, int accumulate, int M, int d,
                          hriemo_stream_t stream);

/* ---- per-kernel-class HIP-event timing on the launch stream (bench.py roofline leg) */
int hriemo_prof_enable(int on);
int hriemo_prof_nclass(void);
const char* hriemo_prof_name(int cls);
int hriemo_prof_collect(int cls, double* ms_total, long* launches, double* work);

/* ---- Linear + bias + dropout + residual + LayerNorm in one kernel (round 4, csrc/gemm_ln.hip) ---------------------------------
 * north_star's "fused bias + LayerNorm + residual epilogue" for the post-LN sites of a fusion layer (cross_modal_block_tacfn.py:
 * 81,92,105,106,118,119):  G = A[M,K] . W[d,K]^T + bias (bf16; kept because the backward reads it; may be NULL),
 * Y16 / Y32 (may be NULL) = LayerNorm(X + drop(G)) * gamma + beta, mean / rstd [M] as hriemo_add_ln_fwd leaves them.  X = X32 (fp32
 * twin) when non-NULL, else X16 (bf16).  A workgroup owns 64 FULL rows (LayerNorm needs them), d in {256, 512, 768}
 * (hriemo_gemm_ln_supported).  Dropout mask and keys = hriemo_add_ln_fwd(_rows); row_index as there (packed sequences) or NULL.
 * G is bit-identical to hriemo_gemm_bf16's output, Y to hriemo_add_ln_fwd's within the summation order of the row statistics. */
int hriemo_gemm_ln_supported(int d);
int hriemo_gemm_ln_fwd(int M, int d, int K, const void* A, long lda, const void* W, long ldw, const float* bias, const void* X16,
                       const float* X32, const float* gamma, const float* beta, void* G, void* Y16, float* Y32, float* mean, float* rstd,
                       float eps, float p_drop, unsigned long long seed, const unsigned long long* seed_dev, unsigned site,
                       long row_offset, const long long* row_index, hriemo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HRIEMO_H_ */
