// Trainer losses of the reference's training scripts, value + gradients in ONE launch each ([B, N_e]-sized problems).
//   hriemo_fusion_loss      multi-label: BCEWithLogits (+ pos_weight) + beta regulariser
//                           scripts/fusion/train_fusion_seq_level_decoder.py:312-326 (multi_label branch), :415-416
//                           scripts/fusion/train_mosei_fusion_seq_level_decoder.py:340-347,383-388,569
//   hriemo_fusion_loss_ce   single-label: CrossEntropyLoss(logits, class index) + the same regulariser
//                           scripts/fusion/train_fusion_seq_level_decoder.py:312-314,325-326,413-414
// Built with -fno-slp-vectorize (Makefile): the scalar sums of these tiny kernels are where hipcc once produced the packed
// fp32 operand-select form `make check-isa` forbids.
#include "common.h"

// ------------------------------------------------------------------ trainer losses, value + gradient in one launch
// loss = mean_{b,e} BCEWithLogits(x, y; pos_weight) + reg(beta), written with its gradients d loss / d logits, d loss / d beta:
//   BCE (torch.nn.BCEWithLogitsLoss, scripts/fusion/train_mosei_fusion_seq_level_decoder.py:569):
//        l = (1 - y) x + (1 + (pw - 1) y) softplus(-x)
//   reg mode 1 (train_fusion_seq_level_decoder.py:318-326):  - coef * mean_b beta (1 - beta)
//   reg mode 2 (train_mosei_fusion_seq_level_decoder.py:340-347, 385-386):  + coef * mean_b H(clamp(beta, eps, 1 - eps)), H = binary entropy
// One block; [B, N_e] is a few hundred values.
__global__ __launch_bounds__(256) void fusion_loss_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ pw,
                                                          const float* __restrict__ beta, int B, int Ne, int mode, float coef, float scale,
                                                          float* __restrict__ loss, float* __restrict__ dx, float* __restrict__ dbeta) {
  __shared__ float red[8];
  const int n = B * Ne;
  float acc = 0.f;
  const float inv_n = 1.f / (float)n, inv_b = 1.f / (float)B;
  for (int t = threadIdx.x; t < n; t += 256) {
    const float xv = x[t], yv = y[t];
    const float lw = pw != nullptr ? 1.f + (pw[t % Ne] - 1.f) * yv : 1.f;
    const float sp = log1pf(__expf(-fabsf(xv))) + fmaxf(-xv, 0.f);            // softplus(-x)
    acc += ((1.f - yv) * xv + lw * sp) * inv_n;
    const float sig = 1.f / (1.f + __expf(-xv));
    dx[t] = ((1.f - yv) - lw * (1.f - sig)) * inv_n * scale;
  }
  for (int b = threadIdx.x; b < B; b += 256) {
    float g = 0.f;
    if (beta != nullptr && mode == 1) {
      const float bv = beta[b];
      acc += -coef * bv * (1.f - bv) * inv_b;
      g = -coef * (1.f - 2.f * bv) * inv_b;
    } else if (beta != nullptr && mode == 2) {
      const float eps = 1e-8f, raw = beta[b];
      const float bv = fminf(fmaxf(raw, eps), 1.f - eps);
      acc += -coef * (bv * __logf(bv) + (1.f - bv) * __logf(1.f - bv)) * inv_b;
      g = (raw > eps && raw < 1.f - eps) ? coef * (__logf(1.f - bv) - __logf(bv)) * inv_b : 0.f;
    }
    if (dbeta != nullptr) dbeta[b] = g * scale;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}


// beta regulariser shared by both losses: value contribution and d/d beta of sample b (mode 0: none)
__device__ __forceinline__ void beta_reg(const float* __restrict__ beta, int b, int mode, float coef, float inv_b, float& acc, float& g) {
  g = 0.f;
  if (beta == nullptr) return;
  if (mode == 1) {
    const float bv = beta[b];
    acc += -coef * bv * (1.f - bv) * inv_b;
    g = -coef * (1.f - 2.f * bv) * inv_b;
  } else if (mode == 2) {
    const float eps = 1e-8f, raw = beta[b];
    const float bv = fminf(fmaxf(raw, eps), 1.f - eps);
    acc += -coef * (bv * __logf(bv) + (1.f - bv) * __logf(1.f - bv)) * inv_b;
    g = (raw > eps && raw < 1.f - eps) ? coef * (__logf(1.f - bv) - __logf(bv)) * inv_b : 0.f;
  }
}

// loss = mean over the labelled samples of [logsumexp_c x[b, c] - x[b, label_b]] + reg(beta): torch.nn.CrossEntropyLoss() (mean
// reduction, no label smoothing, no class weights, ignore_index = -100) as the IEMOCAP trainer builds it (:413-414);
// d loss / d x = (softmax(x) - onehot(label)) / n_labelled.  A sample whose label is -100 is skipped like the reference skips it:
// no loss term, zero gradient, not counted in the mean (every sample ignored: 0 / 0 = NaN, as torch); the beta regulariser stays a
// mean over all B samples.  One thread per sample (C = number of emotion classes, a handful); any OTHER label outside [0, C)
// poisons the loss with NaN instead of reading out of bounds (torch raises for them on the host; a kernel cannot).
#define CE_IGNORE_INDEX (-100LL)
__global__ __launch_bounds__(256) void fusion_loss_ce_kernel(const float* __restrict__ x, const long long* __restrict__ label,
                                                             const float* __restrict__ beta, int B, int C, int mode, float coef,
                                                             float scale, float* __restrict__ loss, float* __restrict__ dx,
                                                             float* __restrict__ dbeta) {
  __shared__ float red[4];
  __shared__ float cnt[4];
  float nl = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) nl += label[b] != CE_IGNORE_INDEX ? 1.f : 0.f;
  nl = wave_sum(nl);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = nl;
  __syncthreads();
  const float n_lab = (cnt[0] + cnt[1]) + (cnt[2] + cnt[3]);
  const float inv_n = 1.f / n_lab;                 // n_lab = 0: inf, and 0 * inf below is the NaN torch returns
  float acc = 0.f, ce = 0.f;
  const float inv_b = 1.f / (float)B;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* xr = x + (long)b * C;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, xr[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(xr[c] - m);
    const long long lb = label[b];
    const bool ign = lb == CE_IGNORE_INDEX;
    const bool ok = lb >= 0 && lb < C;
    const float lse = m + __logf(se);
    if (!ign) ce += ok ? (lse - xr[ok ? lb : 0]) : __builtin_nanf("");
    const float inv_se = 1.f / se;
    for (int c = 0; c < C; ++c)
      dx[(long)b * C + c] = ign ? 0.f : (__expf(xr[c] - m) * inv_se - ((long long)c == lb ? 1.f : 0.f)) * inv_n * scale;
    float g;
    beta_reg(beta, b, mode, coef, inv_b, acc, g);
    if (dbeta != nullptr) dbeta[b] = g * scale;
  }
  ce = wave_sum(ce);
  acc = wave_sum(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = acc; cnt[threadIdx.x >> 6] = ce; }
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (((cnt[0] + cnt[1]) + (cnt[2] + cnt[3])) * inv_n + (red[0] + red[1]) + (red[2] + red[3])) * scale;
}

extern "C" int hriemo_fusion_loss(const float* logits, const float* targets, const float* pos_weight, const float* beta, int B, int Ne,
                                  int reg_mode, float reg_coef, float scale, float* loss, float* dlogits, float* dbeta, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && Ne > 0 && loss != nullptr && dlogits != nullptr, "fusion_loss: bad arguments");
  HRIEMO_CHECK(reg_mode >= 0 && reg_mode <= 2, "fusion_loss: reg_mode %d (0 none, 1 -c*mean(b(1-b)), 2 +c*entropy)", reg_mode);
  hipLaunchKernelGGL(fusion_loss_kernel, dim3(1), dim3(256), 0, st, logits, targets, pos_weight, beta, B, Ne, reg_mode, reg_coef, scale, loss,
                     dlogits, dbeta);
  HRIEMO_LAUNCH_CHECK("fusion_loss_kernel");
  return 0;
}


extern "C" int hriemo_fusion_loss_ce(const float* logits, const long long* labels, const float* beta, int B, int C, int reg_mode,
                                     float reg_coef, float scale, float* loss, float* dlogits, float* dbeta, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && C > 0 && logits != nullptr && labels != nullptr && loss != nullptr && dlogits != nullptr, "fusion_loss_ce: bad arguments");
  HRIEMO_CHECK(reg_mode >= 0 && reg_mode <= 2, "fusion_loss_ce: reg_mode %d (0 none, 1 -c*mean(b(1-b)), 2 +c*entropy)", reg_mode);
  hipLaunchKernelGGL(fusion_loss_ce_kernel, dim3(1), dim3(256), 0, st, logits, labels, beta, B, C, reg_mode, reg_coef, scale, loss, dlogits, dbeta);
  HRIEMO_LAUNCH_CHECK("fusion_loss_ce_kernel");
  return 0;
}
