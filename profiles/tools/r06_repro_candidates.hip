// Round 6: five hand-written candidates for a SMALL reproducer of the exec-join miscompile (profiles/r06_compiler_fault.md).  None of them triggers it:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S profiles/tools/r06_repro_candidates.hip -o /tmp/c.s && python profiles/tools/exec_join_check.py /tmp/c.s   -> 0 findings
// c4 / c5 do produce join blocks with lane-mask merges in front of the exec restore (the precondition); the allocator splits its live ranges elsewhere.  The reproducer is the
// IR of a real kernel: profiles/tools/r06_llc_repro.sh.
#include <hip/hip_runtime.h>

namespace c1 {
// candidate minimal reproducer: many doubles live across a divergent `if`
template <int N>
__global__ void __launch_bounds__(64) k(const double* __restrict__ in, double* __restrict__ out, const int* __restrict__ flag) {
  const int t = threadIdx.x;
  double a[N];
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = in[i * 64 + t];
  double p = 1.0;
#pragma unroll
  for (int i = 0; i < N; ++i) p = fma(p, a[i], a[(i * 7 + 3) % N]);
  if (flag[t] > 0) {                      // divergent
    double e = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) e += a[i] * a[i + 8];
    out[128 + t] = e + p;
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) s = fma(a[i], a[(i * 5 + 1) % N], s) * p;
  out[t] = s;
}
template __global__ void k<100>(const double*, double*, const int*);
template __global__ void k<130>(const double*, double*, const int*);
template __global__ void k<160>(const double*, double*, const int*);
template __global__ void k<200>(const double*, double*, const int*);
}  // namespace c1

namespace c2 {
// candidate 2: register pressure rises right BEHIND the join of a divergent `if`
template <int N, int M>
__global__ void __launch_bounds__(64) k(const double* __restrict__ in, double* __restrict__ out, const int* __restrict__ flag) {
  const int t = threadIdx.x;
  double a[N];
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = in[i * 64 + t];
  double p = 1.0;
#pragma unroll
  for (int i = 0; i < N; ++i) p = fma(p, a[i], a[(i * 7 + 3) % N]);
  if (flag[t] > 0) out[128 + t] = p;      // divergent, tiny body
  double b[M];
#pragma unroll
  for (int i = 0; i < M; ++i) b[i] = in[(N + i) * 64 + t];
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i) s = fma(b[i], b[(i * 3 + 1) % M], s);
#pragma unroll
  for (int i = 0; i < N; ++i) s = fma(a[i], a[(i * 5 + 1) % N], s) * p + b[i % M];
  out[t] = s;
}
template __global__ void k<100, 40>(const double*, double*, const int*);
template __global__ void k<110, 60>(const double*, double*, const int*);
template __global__ void k<120, 80>(const double*, double*, const int*);
template __global__ void k<100, 100>(const double*, double*, const int*);
}  // namespace c2

namespace c3 {
// candidate 3: divergent `if` whose body holds uniform (null-pointer) branches; a 64-bit index is used inside the body and again far behind the join
template <int N, int M>
__global__ void __launch_bounds__(64) k(const double* __restrict__ in, double* __restrict__ out, float* e1, float* e2, const int* __restrict__ flag, long stride) {
  const int t = threadIdx.x;
  const long idx = (long)blockIdx.x * stride + flag[64 + t];
  double a[N];
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = in[i * 64 + t];
  double p = 1.0, q = 0.5;
#pragma unroll
  for (int i = 0; i < N; ++i) { p = fma(p, a[i], a[(i * 7 + 3) % N]); q = fma(q, a[(i * 3 + 1) % N], p); }
  if (flag[t] > 0) {                      // divergent
    if (e1) e1[idx] = (float)(p + q);     // uniform
    if (e2) e2[idx] = (float)(p - q);     // uniform
  }
  double b[M];
#pragma unroll
  for (int i = 0; i < M; ++i) b[i] = in[(N + i) * 64 + t];
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i) s = fma(b[i], b[(i * 3 + 1) % M], s);
#pragma unroll
  for (int i = 0; i < N; ++i) s = fma(a[i], a[(i * 5 + 1) % N], s) * p + b[i % M];
  out[idx] = s + q;
}
template __global__ void k<100, 40>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<110, 60>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<120, 80>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<100, 100>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<124, 30>(const double*, double*, float*, float*, const int*, long);
}  // namespace c3

namespace c4 {
// candidate 4: + a divergent boolean that is updated inside the divergent `if` and consumed behind the join (a lane-mask phi: SGPR copy at the top of the join block)
template <int N, int M>
__global__ void __launch_bounds__(64) k(const double* __restrict__ in, double* __restrict__ out, float* e1, float* e2, const int* __restrict__ flag, long stride) {
  const int t = threadIdx.x;
  const long idx = (long)blockIdx.x * stride + flag[64 + t];
  double a[N];
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = in[i * 64 + t];
  double p = 1.0, q = 0.5;
#pragma unroll
  for (int i = 0; i < N; ++i) { p = fma(p, a[i], a[(i * 7 + 3) % N]); q = fma(q, a[(i * 3 + 1) % N], p); }
  bool ok = flag[128 + t] != 0;
  if (flag[t] > 0) {                      // divergent
    if (e1) e1[idx] = (float)(p + q);     // uniform
    if (e2) { e2[idx] = (float)(p - q); ok = ok && (p > q); }
  }
  double b[M];
#pragma unroll
  for (int i = 0; i < M; ++i) b[i] = in[(N + i) * 64 + t];
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < M; ++i) s = fma(b[i], b[(i * 3 + 1) % M], s);
#pragma unroll
  for (int i = 0; i < N; ++i) s = fma(a[i], a[(i * 5 + 1) % N], s) * p + b[i % M];
  if (ok) out[idx] = s + q;
}
template __global__ void k<100, 40>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<110, 60>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<120, 80>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<124, 30>(const double*, double*, float*, float*, const int*, long);
}  // namespace c4

namespace c5 {
template <int N, int M, int R>
__global__ void __launch_bounds__(64) k(const double* __restrict__ in, double* __restrict__ out, float* e1, float* e2, const int* __restrict__ flag, long stride) {
  const int t = threadIdx.x;
  double a[N];
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = in[i * 64 + t];
  double p = 1.0, q = 0.5;
#pragma unroll
  for (int i = 0; i < N; ++i) { p = fma(p, a[i], a[(i * 7 + 3) % N]); q = fma(q, a[(i * 3 + 1) % N], p); }
  const long idx = (long)blockIdx.x * stride + flag[64 + t];       // defined right in front of the branch, used in its body and once at the very end
  const long idx2 = (long)blockIdx.y * stride + flag[192 + t];
  bool ok = flag[128 + t] != 0;
  if (flag[t] > 0) {                      // divergent
    if (e1) e1[idx] = (float)(p + q);     // uniform
    if (e2) { e2[idx2] = (float)(p - q); ok = ok && (p > q); }
  }
  double b[M];
#pragma unroll
  for (int i = 0; i < M; ++i) b[i] = in[(N + i) * 64 + t];
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int i = 0; i < M; ++i) s = fma(b[i], b[(i * 3 + 1 + r) % M], s);
#pragma unroll
    for (int i = 0; i < N; ++i) s = fma(a[i], a[(i * 5 + 1 + r) % N], s) * p + b[(i + r) % M];
  }
  if (ok) { out[idx] = s + q; out[idx2] = s - q; }
}
template __global__ void k<100, 40, 3>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<110, 30, 3>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<118, 20, 2>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<120, 12, 4>(const double*, double*, float*, float*, const int*, long);
template __global__ void k<122, 8, 4>(const double*, double*, float*, float*, const int*, long);
}  // namespace c5
