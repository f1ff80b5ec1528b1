// Calibration kernels for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters in the GN kernel's OWN access pattern
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern"): every lane reads 16 consecutive
// floats as four float4 loads (lane stride 64 B) and writes them back as four float4 stores, as gn_kernel<2,16,4,float>
// does for its four `th` / `dtheta` rows.  Known traffic: 4*n bytes read, 4*n bytes written.
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
