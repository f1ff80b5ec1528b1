// Calibration kernels for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters in the GN kernel's OWN access patterns
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").
//  calib_dword_rw : every lane reads 16 consecutive floats as four float4 loads (lane stride 64 B) and writes them back as four
//                   float4 stores, as gn_kernel<2,16,4,float> does for its `th` / `dtheta` rows.  Known: 4 n bytes each way.
//  calib_gather8  : every lane reads ONE 8-byte pair (the kernel's column-pair tap load) from its own 128-byte line of a large
//                   array -- `halves` = 1: offset 0 of the line only; 2: offsets 0 and 64 (both 64-byte halves of the line, two
//                   loads).  If FETCH_SIZE doubles from 1 to 2 the memory side fetches 64-byte sectors, otherwise whole lines.
#include <hip/hip_runtime.h>
extern "C" __global__ void calib_dword_rw(const float* __restrict__ in, float* __restrict__ out, long n16) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  typedef float __attribute__((ext_vector_type(4))) f4;
  const f4* p = (const f4*)(in + i * 16);
  f4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = p[k];
  f4* q = (f4*)(out + i * 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = v[k] + 1.0f;
}
extern "C" int calib_launch(const float* in, float* out, long n16, void* stream) {
  hipLaunchKernelGGL(calib_dword_rw, dim3((unsigned)((n16 + 63) / 64)), dim3(64), 0, (hipStream_t)stream, in, out, n16);
  return (int)hipGetLastError();
}

template <int HALVES>
__global__ void calib_gather8(const float* __restrict__ in, float* __restrict__ out, long nlines) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlines) return;
  // a multiplicative shuffle of the line index: neighbouring lanes read far-apart lines, like the taps of different trajectories
  const long line = (i * 40503L) % nlines;
  typedef float __attribute__((ext_vector_type(2))) f2;
  f2 a = *(const f2*)(in + line * 32);
  float s = a.x + a.y;
  if (HALVES == 2) { f2 b = *(const f2*)(in + line * 32 + 16); s += b.x + b.y; }
  if (s == 123456.789f) out[i] = s;      // (never true: keeps the loads alive without a write stream)
}
extern "C" int calib_gather_launch(const float* in, float* out, long nlines, int halves, void* stream) {
  const dim3 grid((unsigned)((nlines + 63) / 64)), block(64);
  if (halves == 2) hipLaunchKernelGGL(calib_gather8<2>, grid, block, 0, (hipStream_t)stream, in, out, nlines);
  else hipLaunchKernelGGL(calib_gather8<1>, grid, block, 0, (hipStream_t)stream, in, out, nlines);
  return (int)hipGetLastError();
}
