// atomic_probe.hip -- what bounds the shared-SDF gradient scatter of gn_backward_kernel?  1024 wavefronts x 16 atomic
// instructions (the backward kernel's count at B = 4096: 4 states x 4 taps per lane) onto a 256 x 256 fp32 grid, with the
// lane -> address pattern varied:
//   0 scattered : every lane its own pseudo-random pixel                          (64 distinct lines per instruction)
//   1 pairs     : lanes 2m, 2m+1 hit horizontally adjacent pixels                 (32 distinct lines per instruction)
//   2 quads     : lanes 4m..4m+3 hit 4 adjacent pixels of one row                 (16 distinct lines)
//   3 rows16    : lanes 16m..16m+15 hit 16 adjacent pixels                        (4 distinct lines)
// each with workgroup-scope atomics into the XCD's own partial grid (what the kernel does) or agent-scope atomics into one grid,
// at 100 % / 35 % active lanes.   Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 atomic_probe.hip -o /tmp/ap && /tmp/ap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int GROUPW, bool LOCAL>
__global__ void __launch_bounds__(64) probe(float* grid, int active_pct) {
  const int lane = threadIdx.x, wave = blockIdx.x;
  const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);
  float* g = grid + (LOCAL ? (size_t)xcc * 65536 : 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t h = hash32((uint32_t)(wave * 64 + lane / GROUPW) * 16u + i);
    // pixels concentrated on ~1/3 of the grid (the band around the obstacles), like the hinge-active taps
    const int row = 64 + (h % 96), col = ((h >> 8) % (256 / GROUPW)) * GROUPW + (lane % GROUPW);
    const bool act = (int)(hash32(h ^ 0x9e3779b9u) % 100u) < active_pct;
    if (act) {
      if (LOCAL) __hip_atomic_fetch_add(g + row * 256 + col, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(g + row * 256 + col, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int GROUPW, bool LOCAL>
float run(float* grid, int pct) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((probe<GROUPW, LOCAL>), dim3(1024), dim3(64), 0, 0, grid, pct);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 500;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<GROUPW, LOCAL>), dim3(1024), dim3(64), 0, 0, grid, pct);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}

int main() {
  float* grid; hipMalloc(&grid, 8 * 65536 * sizeof(float)); hipMemset(grid, 0, 8 * 65536 * sizeof(float));
  for (int pct : {100, 35}) {
    printf("active %3d %% | scattered  pairs  quads  rows16   (us per launch of 1024 waves x 16 atomic instructions)\n", pct);
    printf("  xcd-local    | %8.2f %6.2f %6.2f %7.2f\n", run<1, true>(grid, pct), run<2, true>(grid, pct), run<4, true>(grid, pct), run<16, true>(grid, pct));
    printf("  agent scope  | %8.2f %6.2f %6.2f %7.2f\n", run<1, false>(grid, pct), run<2, false>(grid, pct), run<4, false>(grid, pct), run<16, false>(grid, pct));
  }
  return 0;
}
