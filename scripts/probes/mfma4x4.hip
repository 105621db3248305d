// probe: operand / result layout of v_mfma_f32_4x4x1_16B_f32 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma4x4.hip -o /tmp/mfma4x4 && /tmp/mfma4x4
// pass 0: A[lane] = 1 + lane, B = 1 -> D[lane][reg] names the A lane; pass 1: A = 1, B[lane] = 1 + lane -> the B lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int mode) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  const float a = mode == 0 ? (float)(1 + l) : 1.f, b = mode == 1 ? (float)(1 + l) : 1.f;
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d;
  if (hipMalloc(&d, 256 * 4) != hipSuccess) return 1;
  float h[2][256];
  for (int m = 0; m < 2; ++m) {
    k<<<1, 64>>>(d, m);
    if (hipMemcpy(h[m], d, sizeof h[m], hipMemcpyDeviceToHost) != hipSuccess) return 1;
  }
  int ok = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int fa = (int)h[0][l * 4 + r] - 1, fb = (int)h[1][l * 4 + r] - 1;
      if (l < 8 || l >= 60) printf("lane %2d reg %d: A lane %2d  B lane %2d\n", l, r, fa, fb);
      // hypothesis: D[lane 4b + j][reg i] = A[lane 4b + i] * B[lane 4b + j]
      if (!(fa == (l / 4) * 4 + r && fb == l)) ok = 0;
    }
  printf("hypothesis D[4b+j][i] = A[4b+i]*B[4b+j]: %s\n", ok ? "HOLDS" : "FAILS");
  return 0;
}
