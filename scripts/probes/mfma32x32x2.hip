// probe: operand / result layout of v_mfma_f32_32x32x2f32 on gfx950 (round 6: the re-decomposition of the backward).
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma32x32x2.hip -o /tmp/mfma32 && /tmp/mfma32
// D[32 x 32] += A[32 x 2] * B[2 x 32].  Hypothesis: A operand of lane l = A[row l % 32][k = l / 32], B operand = B[k = l / 32][col l % 32],
// D register i of lane l = D[row 8 (i / 4) + 4 (l / 32) + i % 4][col l % 32].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* out) {  // A [32][2], B [2][32] row-major
  const int l = threadIdx.x;
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const float a = A[(l % 32) * 2 + l / 32], b = B[(l / 32) * 32 + l % 32];
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out[(8 * (i / 4) + 4 * (l / 32) + i % 4) * 32 + l % 32] = c[i];
}
int main() {
  float hA[64], hB[64], hD[1024], *dA, *dB, *dD;
  for (int i = 0; i < 64; ++i) { hA[i] = (float)(1 + (i * 7) % 13); hB[i] = (float)(2 + (i * 5) % 11); }
  if (hipMalloc(&dA, 256) != hipSuccess || hipMalloc(&dB, 256) != hipSuccess || hipMalloc(&dD, 4096) != hipSuccess) return 1;
  hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  if (hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  int bad = 0;
  for (int r = 0; r < 32; ++r)
    for (int c = 0; c < 32; ++c) {
      const float ref = hA[r * 2] * hB[c] + hA[r * 2 + 1] * hB[32 + c];
      if (hD[r * 32 + c] != ref) { if (bad < 5) printf("D[%d][%d] = %g, expected %g\n", r, c, hD[r * 32 + c], ref); ++bad; }
    }
  printf("hypothesis (A[l%%32][l/32], B[l/32][l%%32], D reg i -> row 8(i/4)+4(l/32)+i%%4, col l%%32): %s (%d mismatches)\n", bad ? "FAILS" : "HOLDS", bad);
  return 0;
}
