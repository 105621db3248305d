// probe (round 6): how fast one SIMD of gfx950 issues fp32 MFMAs -- v_mfma_f32_16x16x4 (8 passes) against v_mfma_f32_32x32x2 (16 passes),
// dependent chains against independent ones, one or two waves per SIMD, bare or with the LDS reads and VALU work of bwd32_kernel's
// slots between the MFMAs.  Prints cycles per MFMA per SIMD (s_memrealtime would need its clock; wall time x 2.4 GHz is what the
// bench uses too) and the share of the fp32 matrix peak (64 FLOP / cycle / SIMD).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 2000;
// CH independent chains of 32x32x2; per MFMA slot, by the bits of F: 1 = two ds_read_b32 (the MFMA operands, requested one slot ahead),
// 2 = four FMAs, 4 = one ds_read_b128 (requested one slot ahead; the FMAs' operand when both are on), 8 = one ds_write_b32,
// 16 = eight more FMAs (registers only), 32 = one v_mfma_f32_4x4x1 (registers only), 64 = its a operand from LDS (ds_read_b32, one slot ahead)
template <int CH, int F>
__global__ __launch_bounds__(256) void k32(float* out, int n) {
  __shared__ float lds[8192];
  const int l = threadIdx.x;
  for (int i = l; i < 8192; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = (float)(l & 3), b = 1.f, an = a, bn = b;
  f32x4 d = {0.f, 0.f, 0.f, 0.f}, e = {1.f, 2.f, 3.f, 4.f}, g = {1.f, 1.f, 1.f, 1.f}, gn = g;
  const float* p = lds + (l & 63) * 33 + (l >> 6) * 2048;
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int sl = u * CH + c;
        if (F & 1) { a = an; b = bn; an = p[sl * 2]; bn = p[sl * 2 + 1]; }
        if (F & 4) { g = gn; gn = *(const f32x4*)(lds + 4096 + (sl & 15) * 64 + (l >> 5 & 1) * 4); }
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        if (F & 2) { d.x = fmaf(g.x, e.x, d.x); d.y = fmaf(g.y, e.y, d.y); d.z = fmaf(g.z, e.z, d.z); d.w = fmaf(g.w, e.w, d.w); }
        if (F & 16) {
          e.x = fmaf(d.x, 0.5f, e.x); e.y = fmaf(d.y, 0.5f, e.y); e.z = fmaf(d.z, 0.5f, e.z); e.w = fmaf(d.w, 0.5f, e.w);
          d.x = fmaf(e.y, 0.25f, d.x); d.y = fmaf(e.z, 0.25f, d.y); d.z = fmaf(e.w, 0.25f, d.z); d.w = fmaf(e.x, 0.25f, d.w);
        }
        if (F & 32) { if (F & 64) { g.x = gn.x; gn.x = lds[4096 + sl * 64 + (l & 3)]; } d = __builtin_amdgcn_mfma_f32_4x4x1f32(g.x, e.x, d, 0, 0, 0); }
        if (F & 8) lds[6144 + (l & 255) + (sl & 3) * 256] = d.x;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = d.x + d.y + d.z + d.w + e.x + an + bn + gn.x;
  for (int c = 0; c < CH; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 12345.f) out[l] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void k16(float* out, int n) {
  const int l = threadIdx.x;
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float a = (float)(l & 3), b = 1.f;
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
  if (s == 12345.f) out[l] = s;
}
template <class K>
static void run(const char* name, K kern, int blocks_per_cu, int mfma_per_iter, double flop_per_mfma, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  kern<<<grid, 256>>>(out, 10);
  hipEventRecord(e0);
  kern<<<grid, 256>>>(out, ITER);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)ITER * mfma_per_iter * blocks_per_cu;  // a block = 4 waves = one per SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
  printf("%-58s %d wave(s)/SIMD: %7.1f cycles / MFMA / SIMD at 2.4 GHz = %.3f of the fp32 matrix peak\n", name, blocks_per_cu, cyc,
         flop_per_mfma / cyc / 64.0);
}
int main() {
  float* out;
  if (hipMalloc(&out, 4096) != hipSuccess) return 1;
  for (int w = 1; w <= 2; ++w) {
    run("16x16x4, 1 chain", k16<1>, w, 8, 2048, out);
    run("16x16x4, 4 chains", k16<4>, w, 32, 2048, out);
    run("32x32x2, 1 chain, bare", k32<1, 0>, w, 8, 4096, out);
    run("32x32x2, 2 chains, bare", k32<2, 0>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands from LDS one slot ahead", k32<2, 1>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands + 4 FMA", k32<2, 3>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands + 12 FMA", k32<2, 19>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands + b128 read", k32<2, 5>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands + b128 read + 4 FMA", k32<2, 7>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands + ds_write_b32", k32<2, 9>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands + b128 + 4 FMA + ds_write", k32<2, 15>, w, 16, 4096, out);
    run("32x32x2, 1 chain, operands + b128 + 4 FMA + ds_write", k32<1, 15>, w, 8, 4096, out);
    run("32x32x2, 2 chains, no LDS, one 4x4x1 MFMA per slot", k32<2, 32>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands + one 4x4x1 MFMA per slot", k32<2, 33>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands + 4x4x1 (a from LDS) ", k32<2, 97>, w, 16, 4096, out);
    run("32x32x2, 2 chains, operands + 4x4x1 (a from LDS) + ds_write", k32<2, 105>, w, 16, 4096, out);
    run("32x32x2, 2 chains, no LDS, 4 FMA", k32<2, 2>, w, 16, 4096, out);
    run("32x32x2, 2 chains, no LDS, 12 FMA", k32<2, 18>, w, 16, 4096, out);
  }
  return 0;
}
