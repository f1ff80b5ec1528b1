// mfma_probe.hip -- can the fp64 matrix pipe take work off a lone wavefront's fp64 VALU stream?  
// One wavefront per SIMD (1024 workgroups of 64), s_memtime around an unrolled loop body of
//   A: 64 v_fma_f64 on 8 independent chains                                  (the GN kernel's regime)
//   B: 16 v_mfma_f64_4x4x4_4b_f64 on 4 independent accumulators              (256 FMAs each = the work of 4 v_fma_f64)
//   C: A and B interleaved 4 : 1                                              (do the two pipes overlap inside ONE wavefront?)
//   D: what feeding ONE such MFMA from lane-owned 4x4 blocks costs: 2 operand gathers through ds_bpermute (4 x 32-bit) + 1 result scatter
// prints shader cycles per loop body (median over wavefronts).   hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o /tmp/mp && /tmp/mp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(64) probe(unsigned long long* out, double* sink, double seed) {
  double c[8], m[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = seed + i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) m[i] = seed * i;
  const double a = seed * 1.0000001, b = seed * 0.9999999;
  int perm = (threadIdx.x * 5) & 63;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          c[i] = __builtin_fma(c[i], a, b);
          if (MODE == 2 && (i & 3) == 3) m[(r * 2 + i / 4) & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m[(r * 2 + i / 4) & 3], 0, 0, 0);
        }
    }
    if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) m[r & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m[r & 3], 0, 0, 0);
    }
    if (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // gather A and B elements from the owning lane (two doubles = four 32-bit bpermutes), multiply, scatter the result back
        int alo = __builtin_amdgcn_ds_bpermute(perm << 2, __double2loint(c[r])), ahi = __builtin_amdgcn_ds_bpermute(perm << 2, __double2hiint(c[r]));
        int blo = __builtin_amdgcn_ds_bpermute(perm << 2, __double2loint(c[r + 4])), bhi = __builtin_amdgcn_ds_bpermute(perm << 2, __double2hiint(c[r + 4]));
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(__hiloint2double(ahi, alo), __hiloint2double(bhi, blo), m[r], 0, 0, 0);
        int dlo = __builtin_amdgcn_ds_bpermute(perm << 2, __double2loint(d)), dhi = __builtin_amdgcn_ds_bpermute(perm << 2, __double2hiint(d));
        m[r] = __hiloint2double(dhi, dlo);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += m[i];
  if (s == 1.2345) sink[0] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE> double run(unsigned long long* d_out, double* d_sink, int waves) {
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE>), dim3(waves), dim3(64), 0, 0, d_out, d_sink, 1.0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(waves);
  hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[waves / 2] / 64.0;
}

int main() {
  unsigned long long* d_out; double* d_sink;
  hipMalloc(&d_out, 4096 * sizeof(unsigned long long)); hipMalloc(&d_sink, 64);
  for (int waves : {1024, 2048}) {
    const double A = run<0>(d_out, d_sink, waves), B = run<1>(d_out, d_sink, waves), Cc = run<2>(d_out, d_sink, waves), Dd = run<3>(d_out, d_sink, waves);
    printf("%d wavefronts (%d per SIMD), cycles per loop body (s_memtime ticks, median wavefront):\n", waves, waves / 1024);
    printf("  A  64 v_fma_f64                         : %7.1f  (%.2f per instruction)\n", A, A / 64);
    printf("  B  16 v_mfma_f64_4x4x4_4b               : %7.1f  (%.2f per instruction, = %.2f per 64-lane-FMA-instruction equivalent)\n", B, B / 16, B / 64);
    printf("  C  64 v_fma_f64 + 16 mfma interleaved   : %7.1f  (A + B = %.1f: %s)\n", Cc, A + B, Cc < 0.8 * (A + B) ? "the pipes overlap" : "no overlap worth having");
    printf("  D  4 x (gather A,B + mfma + scatter D)  : %7.1f  (%.1f per MFMA fed from lane-owned blocks; the 4 v_fma_f64 it replaces cost %.1f)\n", Dd, Dd / 4, A / 16);
  }
  return 0;
}
