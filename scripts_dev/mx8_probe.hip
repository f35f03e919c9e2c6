// Layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (the MX-fp8 GEMM path, csrc/gemm_mx8.hip):
// checks on hardware, with exactly representable integers and power-of-two block scales, that
//   * lane l = (i = l & 15, g = l >> 4) of the A (B) operand holds row (column) i, k = 16g .. 16g+15 in registers 0-3 and
//     k = 64+16g .. 64+16g+15 in registers 4-7 (found with the one-hot sweeps of mx8_probe2/3.hip);
//   * the scale VGPR's byte selected by opsel is an E8M0 exponent (127 = 1.0); lane (i, g) scales row i, k = 32g .. 32g+31;
//   * C/D is the standard 16x16 map (col = l & 15 from the B operand, row = 4*(l >> 4) + reg from the A operand);
//   * __builtin_amdgcn_cvt_pk_fp8_f32 produces OCP e4m3fn bytes (checked against a host encoder for exact values).
// Build: hipcc --offload-arch=gfx950 -O2 scripts_dev/mx8_probe.hip -o scripts_dev/mx8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* SA, const uint8_t* SB, float* D, int opsel) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  i32x8 a, b;
  const int* ap = (const int*)(A + i * 128 + 16 * g);
  const int* bp = (const int*)(B + i * 128 + 16 * g);
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = ap[j]; b[j] = bp[j]; a[4 + j] = ap[16 + j]; b[4 + j] = bp[16 + j]; }
  // scale byte of (row i, k-block g) placed in byte `opsel` of the VGPR, other bytes poisoned
  const unsigned sa = 0x55555555u ^ (0x55u << (8 * opsel)) | ((unsigned)SA[i * 4 + g] << (8 * opsel));
  const unsigned sb = 0x33333333u ^ (0x33u << (8 * opsel)) | ((unsigned)SB[i * 4 + g] << (8 * opsel));
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)sa, 0, (int)sb);
  else if (opsel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, (int)sa, 1, (int)sb);
  else if (opsel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, (int)sa, 2, (int)sb);
  else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, (int)sa, 3, (int)sb);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];      // row from A, column from B
}

__global__ void cvt(const float* x, uint8_t* out, int n) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  if (2 * t + 1 < n) {
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * t], x[2 * t + 1], 0, false);
    out[2 * t] = (uint8_t)(w & 0xff);
    out[2 * t + 1] = (uint8_t)((w >> 8) & 0xff);
  }
}

// OCP e4m3fn decode
static float dec(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 15 && m == 7) return NAN;
  if (e == 0) f = ldexpf((float)m, -9); else f = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}
static uint8_t enc_int(int x) {           // exact for |x| <= 16
  for (int v = 0; v < 256; ++v) if (dec((uint8_t)v) == (float)x && !(v == 0x80)) return (uint8_t)v;
  abort();
}

int main() {
  uint8_t hA[16 * 128], hB[16 * 128], hSA[64], hSB[64];
  srand(7);
  for (int i = 0; i < 16 * 128; ++i) { hA[i] = enc_int(rand() % 9 - 4); hB[i] = enc_int(rand() % 7 - 3); }
  for (int i = 0; i < 64; ++i) { hSA[i] = (uint8_t)(127 + rand() % 5 - 2); hSB[i] = (uint8_t)(127 + rand() % 7 - 3); }
  double ref[16][16];
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int k = 0; k < 128; ++k)
        s += (double)dec(hA[i * 128 + k]) * ldexp(1.0, hSA[i * 4 + k / 32] - 127) * (double)dec(hB[j * 128 + k]) * ldexp(1.0, hSB[j * 4 + k / 32] - 127);
      ref[i][j] = s;
    }
  uint8_t *dA, *dB, *dSA, *dSB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dSA, hSA, 64, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 64, hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int opsel = 0; opsel < 4; ++opsel) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dD, opsel);
    float hD[256];
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if ((double)hD[i * 16 + j] != ref[i][j]) ++bad;
    printf("opsel %d: %d / 256 mismatches (D[0][0] %g ref %g, D[3][7] %g ref %g)\n", opsel, bad, hD[0], ref[0][0], hD[3 * 16 + 7], ref[3][7]);
    bad_total += bad;
  }
  // conversion builtin vs host: every e4m3 value itself, midpoints (round to nearest even), and saturation behaviour
  float hx[512]; int n = 0;
  for (int v = 0; v < 256; ++v) { float f = dec((uint8_t)v); if (!std::isnan(f)) hx[n++] = f; }
  hx[n++] = 17.f; hx[n++] = 18.f; hx[n++] = 19.f; hx[n++] = 21.f; hx[n++] = 448.f; hx[n++] = 449.f; hx[n++] = 480.f; hx[n++] = 1000.f;
  hx[n++] = 0.3f; hx[n++] = -0.7f; hx[n++] = 1e-4f; hx[n++] = 0.0019f;
  if (n & 1) hx[n++] = 0.f;
  float* dx; uint8_t* dout; hipMalloc(&dx, n * 4); hipMalloc(&dout, n);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt, dim3(2), dim3(256), 0, 0, dx, dout, n);
  uint8_t ho[512]; hipMemcpy(ho, dout, n, hipMemcpyDeviceToHost);
  int cbad = 0;
  for (int k = 0; k < n; ++k) {
    // nearest e4m3 value (ties to even mantissa), saturating at 448
    float best = 0; double bd = 1e30; int bv = 0;
    for (int v = 0; v < 256; ++v) { float f = dec((uint8_t)v); if (std::isnan(f)) continue; double d = fabs((double)f - (double)hx[k]);
      if (d < bd || (d == bd && !(v & 1))) { bd = d; best = f; bv = v; } }
    const float got = dec(ho[k]);
    const bool exact_in = (double)best == (double)hx[k];
    if (got != best && !(std::isnan(got) && fabsf(hx[k]) > 448.f)) { if (cbad < 12) printf("  cvt %g -> 0x%02x = %g, nearest %g (0x%02x)\n", hx[k], ho[k], got, best, bv); ++cbad; }
    if (fabsf(hx[k]) > 448.f) printf("  overflow input %g -> 0x%02x = %g\n", hx[k], ho[k], got);
    (void)exact_in;
  }
  printf("cvt_pk_fp8_f32: %d / %d differ from round-to-nearest-even e4m3fn\n", cbad, n);
  printf(bad_total == 0 && cbad == 0 ? "MX8 PROBE PASS\n" : "MX8 PROBE FAIL\n");
  return bad_total != 0;
}
