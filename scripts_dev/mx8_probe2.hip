// Diagnostic companion of mx8_probe.hip: maps, by one-hot experiments, (1) which operand byte of the B side each operand byte
// of the A side is multiplied with, (2) which lane's scale byte applies to which (row, k-block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// a_bytes/b_bytes: [64 lanes][32 bytes]; sa/sb: [64 lanes] scale byte (in byte 0)
__global__ void run(const uint8_t* ab, const uint8_t* bb, const uint8_t* sa, const uint8_t* sb, float* D) {
  const int l = threadIdx.x;
  i32x8 a, b;
  const int* ap = (const int*)(ab + l * 32);
  const int* bp = (const int*)(bb + l * 32);
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = ap[j]; b[j] = bp[j]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, (int)sa[l], 0, (int)sb[l]);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
static float dec(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 15 && m == 7) return NAN;
  if (e == 0) f = ldexpf((float)m, -9); else f = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}
int main() {
  uint8_t ha[2048], hb[2048], hsa[64], hsb[64]; float hD[256];
  uint8_t *da, *db, *dsa, *dsb; float* dD;
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 1024);
  auto go = [&]() {
    hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(run, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  };
  // (1) data pairing: B lanes hold code (g*32 + j) + 8 (distinct positive values), A one-hot 1.0 (0x38) at (lane, byte)
  for (int i = 0; i < 64; ++i) { hsa[i] = 127; hsb[i] = 127; }
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) hb[l * 32 + j] = (uint8_t)(((l >> 4) * 32 + j) % 112 + 8);
  printf("pairing: A operand (g, byte j) row 5 -> D[row?][col 0..] shows dec(code of the B byte it met)\n");
  for (int g = 0; g < 4; ++g) {
    for (int j = 0; j < 32; ++j) {
      for (int k = 0; k < 2048; ++k) ha[k] = 0;
      ha[(g * 16 + 5) * 32 + j] = 0x38;           // 1.0 in lane (i=5, g)
      go();
      // find the nonzero row, report value at col 0 and the (g', j') it corresponds to
      int row = -1; float v = 0;
      for (int r = 0; r < 16; ++r) if (hD[r * 16 + 0] != 0.f) { row = r; v = hD[r * 16]; }
      int code = -1;
      for (int c = 0; c < 128; ++c) if (dec((uint8_t)(c % 112 + 8)) == v) { code = c; break; }
      printf(" (g%d,j%2d)->row%2d B(g%d,j%2d)%s", g, j, row, code / 32, code % 32, (j % 4 == 3) ? "\n" : "");
    }
  }
  // (2) scale mapping: data all 1.0; scale B = 127 everywhere; scale A one-hot: lane ls gets 131 (x16), others 127.
  for (int k = 0; k < 2048; ++k) { ha[k] = 0x38; hb[k] = 0x38; }
  printf("scale-A lane -> which D rows change (D = 128 when all scales are 1; a x16 block adds 32*15 = 480)\n");
  for (int ls = 0; ls < 64; ++ls) {
    for (int i = 0; i < 64; ++i) { hsa[i] = 127; hsb[i] = 127; }
    hsa[ls] = 131;
    go();
    printf(" lane %2d:", ls);
    for (int r = 0; r < 16; ++r) if (hD[r * 16 + 3] != 128.f) printf(" row%d=%g", r, hD[r * 16 + 3]);
    for (int c = 0; c < 16; ++c) if (hD[2 * 16 + c] != 128.f && hD[2 * 16 + 3] == 128.f) printf(" [col%d of row2=%g]", c, hD[2 * 16 + c]);
    printf("%s", (ls % 4 == 3) ? "\n" : "");
  }
  printf("scale-B lane -> which D cols change\n");
  for (int ls = 0; ls < 64; ++ls) {
    for (int i = 0; i < 64; ++i) { hsa[i] = 127; hsb[i] = 127; }
    hsb[ls] = 131;
    go();
    printf(" lane %2d:", ls);
    for (int c = 0; c < 16; ++c) if (hD[3 * 16 + c] != 128.f) printf(" col%d=%g", c, hD[3 * 16 + c]);
    printf("%s", (ls % 4 == 3) ? "\n" : "");
  }
  return 0;
}
