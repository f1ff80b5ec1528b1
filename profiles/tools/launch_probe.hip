// Where does the fixed ~6 us of a gn_kernel launch go?  A kernel with the same launch geometry (1024 x 64), the same
// by-value 1.2 KB argument block and the same th / dtheta access pattern stamps s_memrealtime (100 MHz) at its phase
// boundaries; the host prints, over all wavefronts of one launch in the middle of a back-to-back series:
//   start skew (dispatch), kernarg latency, th load latency, store + drain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

struct Args { const float* th; float* out; unsigned long long* stamps; double pad[150]; };   // ~1.2 KB like dgp::GnParams

template <bool COALESCED>
__global__ void __launch_bounds__(64) probe(const Args a) {
  const unsigned long long t0 = wall_clock64();
  const double k = a.pad[149];                               // last word of the argument block (forces the scalar loads)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  typedef float __attribute__((ext_vector_type(4))) f4;
  const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
  // strided: lane l reads its own 64 contiguous bytes (4 x 16 B) -- every load instruction touches 64 cache lines;
  // coalesced: load i reads 16 B at (block, i, lane) -- 64 lanes x 16 B contiguous, 16 lines per instruction
  const f4* p = COALESCED ? (const f4*)(a.th + (size_t)blockIdx.x * 1024) + threadIdx.x : (const f4*)(a.th + lane * 16);
  f4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = p[COALESCED ? i * 64 : i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = wall_clock64();
  f4* q = COALESCED ? (f4*)(a.out + (size_t)blockIdx.x * 1024) + threadIdx.x : (f4*)(a.out + lane * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[COALESCED ? i * 64 : i] = v[i] + (float)k;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t3 = wall_clock64();
  if (threadIdx.x == 0 && a.stamps) {
    unsigned long long* s = a.stamps + (size_t)blockIdx.x * 4;
    s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3;
  }
}

template <bool CO> static void run_variant(const char* name);
static float *th, *out; static unsigned long long* st;
int main() {
  run_variant<false>("lane-strided (the GN kernel's pattern)");
  run_variant<true>("coalesced");
  return 0;
}
template <bool CO> static void run_variant(const char* name) {
  printf("== %s\n", name);
  const int blocks = 1024;
  hipMalloc(&th, (size_t)blocks * 64 * 16 * 4 * 10); hipMalloc(&out, (size_t)blocks * 64 * 16 * 4); hipMalloc(&st, blocks * 4 * 8);
  hipMemset(th, 0, (size_t)blocks * 64 * 16 * 4 * 10);
  Args a; memset(&a, 0, sizeof(a)); a.out = out; a.stamps = nullptr;
  for (int i = 0; i < 3000; ++i) { a.th = th + (size_t)(i % 10) * blocks * 64 * 16; hipLaunchKernelGGL(probe<CO>, dim3(blocks), dim3(64), 0, 0, a); }
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 1000; ++i) {
    a.th = th + (size_t)(i % 10) * blocks * 64 * 16;
    a.stamps = (i == 500) ? st : nullptr;
    hipLaunchKernelGGL(probe<CO>, dim3(blocks), dim3(64), 0, 0, a);
  }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("back-to-back: %.2f us/launch\n", ms);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), st, blocks * 4 * 8, hipMemcpyDeviceToHost);
  unsigned long long first = ~0ull, last_start = 0, last_end = 0;
  for (int b = 0; b < blocks; ++b) { first = std::min(first, h[b * 4]); last_start = std::max(last_start, h[b * 4]); last_end = std::max(last_end, h[b * 4 + 3]); }
  std::vector<double> d1, d2, d3;
  for (int b = 0; b < blocks; ++b) { d1.push_back((h[b*4+1]-h[b*4]) * 0.01); d2.push_back((h[b*4+2]-h[b*4+1]) * 0.01); d3.push_back((h[b*4+3]-h[b*4+2]) * 0.01); }
  auto stat = [](std::vector<double>& v, const char* n) { std::sort(v.begin(), v.end()); printf("%-22s min %.2f  median %.2f  max %.2f us\n", n, v.front(), v[v.size()/2], v.back()); };
  printf("first wave start -> last wave start (dispatch skew): %.2f us;  first start -> last end: %.2f us\n", (last_start - first) * 0.01, (last_end - first) * 0.01);
  stat(d1, "kernarg scalar loads"); stat(d2, "th loads (4 x 16 B)"); stat(d3, "stores acked");
}
