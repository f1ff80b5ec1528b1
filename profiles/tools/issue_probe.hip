// Issue-rate / latency probe for the instructions the GN kernel is made of (MI355X, one wavefront per SIMD unless noted):
//   dependent vs independent v_fma_f64 chains, v_mov_b32 DPP row shifts feeding FMAs, v_accvgpr moves, v_rcp_f64.
// Prints cycles per instruction from s_memtime (constant 100 MHz) and wall-clock.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define N_ITER 2000

template <int CHAINS>
__global__ void __launch_bounds__(64) k_fma(double* out, double a, double b) {
  double x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / CHAINS; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

// 8 DPP moves + 8 fmas per iteration (the PCR pattern: fetch a double = 2 dpp movs, use it in an FMA)
__global__ void __launch_bounds__(64) k_dpp(double* out, double a, double b) {
  double x[4];
  int lo[4], hi[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) x[c] = threadIdx.x * 1e-3 + c;
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      lo[c] = __builtin_amdgcn_update_dpp(0, __double2loint(x[c]), 0x111, 0xf, 0xf, true);
      hi[c] = __builtin_amdgcn_update_dpp(0, __double2hiint(x[c]), 0x111, 0xf, 0xf, true);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double f = __hiloint2double(hi[c], lo[c]);
      asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[c]) : "v"(f), "v"(a));
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

__global__ void __launch_bounds__(64) k_acc(double* out, double a, double b) {
  int x[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] = threadIdx.x + c;
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int t;
      asm volatile("v_accvgpr_write_b32 a[%1], %2\n v_accvgpr_read_b32 %0, a[%1]" : "=v"(t) : "n"(0), "v"(x[c]) : "a0");
      x[c] = t + 1;
    }
  }
  int s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(64) k_rcp(double* out, double a, double b) {
  double x[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) x[c] = threadIdx.x * 1e-3 + c + 1.0;
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[c]));
  }
  out[blockIdx.x * 64 + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}

__global__ void __launch_bounds__(64) k_empty(double* out, double a, double b) {
  if (a == 12345.0) out[threadIdx.x] = b;
}

template <typename K>
static void run(const char* name, K kern, int blocks, int instr_per_iter, double* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  const int reps = 200;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double n_instr = (double)instr_per_iter * N_ITER;
  printf("%-28s blocks=%5d  %9.2f us/launch  %7.3f ns/instr  (= %.2f cycles @2.4GHz)\n", name, blocks, us,
         n_instr > 0 ? us * 1e3 / n_instr : 0.0, n_instr > 0 ? us * 1e3 / n_instr * 2.4 : 0.0);
}

int main() {
  double* d;
  hipMalloc(&d, 64 * 65536 * sizeof(double));
  // warm the clocks up (~0.5 s of fp64 work) before anything is timed
  for (int i = 0; i < 4000; ++i) hipLaunchKernelGGL(k_fma<8>, dim3(4096), dim3(64), 0, 0, d, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  for (int pass = 0; pass < 2; ++pass)
  for (int blocks : {1024, 2048, 4096}) {
    run("empty", k_empty, blocks, 0, d);
    run("fma_f64 1 chain (dependent)", k_fma<1>, blocks, 8, d);
    run("fma_f64 2 chains", k_fma<2>, blocks, 8, d);
    run("fma_f64 4 chains", k_fma<4>, blocks, 8, d);
    run("fma_f64 8 chains", k_fma<8>, blocks, 8, d);
    run("8 dpp mov + 8 fma", k_dpp, blocks, 16, d);
    run("8 x (acc write + read + add)", k_acc, blocks, 24, d);
    run("rcp_f64 4 chains", k_rcp, blocks, 8, d);
  }
  return 0;
}
