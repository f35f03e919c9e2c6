// Which operand bytes does lane (i, gs)'s scale byte apply to?  A: one 1.0 at (lane (0,g), byte j), B all ones; scale of
// lane (0, gs) on the A side = 16, all others 1: D[0][*] = 16 iff byte (g, j) belongs to the block that lane scales.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
__global__ void run(const uint8_t* ab, const uint8_t* bb, const unsigned* sa, const unsigned* sb, float* D) {
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
int main() {
  uint8_t ha[2048], hb[2048]; unsigned hsa[64], hsb[64]; float hD[256];
  uint8_t *da, *db; unsigned *dsa, *dsb; float* dD;
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 1024);
  for (int k = 0; k < 2048; ++k) hb[k] = 0x38;
  hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
  for (int side = 0; side < 2; ++side)
  for (int gs = 0; gs < 4; ++gs) {
    for (int i = 0; i < 64; ++i) { hsa[i] = 127; hsb[i] = 127; }
    (side ? hsb : hsa)[gs * 16 + 0] = 131;
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    printf("scale-%c lane (0,g%d) x16 -> bytes of lane-block g scaled (X) :\n", side ? 'B' : 'A', gs);
    for (int g = 0; g < 4; ++g) {
      printf("   g%d ", g);
      for (int j = 0; j < 32; ++j) {
        for (int k = 0; k < 2048; ++k) ha[k] = 0;
        ha[(g * 16 + 0) * 32 + j] = 0x38;
        hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(run, dim3(1), dim3(64), 0, 0, side ? db : da, side ? da : db, dsa, dsb, dD);
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        const float v = hD[0];
        printf("%c", v == 16.f ? 'X' : (v == 1.f ? '.' : '?'));
      }
      printf("\n");
    }
  }
  return 0;
}
