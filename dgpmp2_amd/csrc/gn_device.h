// gn_device.h -- device lane context and kernel entry points (HIP only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#ifndef DGP_HD
#define DGP_HD __host__ __device__ __forceinline__
#endif
#include "dgp_host.h"
#include "gn_long.h"

namespace dgp_dev {

// Device lane context: cross-lane fetches are ds_bpermute (any lane -> any lane inside the wavefront; the LDS
// crossbar is used, no LDS memory is touched).
struct DevCtx {
  char* lds_;      // the workgroup's (= wavefront's) LDS staging block, dgp::WaveStore<...>::kLdsBytes
  char* stash_;    // dgp::SinvStash block (d = 6 kernels), or null
  char* wb_;       // LDS copy of the Woodbury constant table (QK_WB kernels), dgp::kWbLdsBytes
  char* long_;     // gn_long.h: the wavefront's dynamic LDS block (per-row S_k^-1, z_k slots), or null
  char* chain_;    // the chain backward kernels: lane-private slots of the running cotangent and the accumulated start / goal gradients (dgp::ChainSlots<C, d>), or null
  __device__ __forceinline__ char* chain_lds() const { return chain_; }
  __device__ __forceinline__ char* long_lds() const { return long_; }
  // writes of this wavefront to global memory become visible to its own later loads (gn_long.h: MODE_SOLVE keeps the state in th_out; gn_backward.h: the
  // errors' prologue hands its trajectory gradient to the main program).  WORKGROUP scope (the fences inside __syncthreads; the workgroup is this one wavefront,
  // its CU's vector L1 is write-through and shared by the whole workgroup): an agent-scope fence (__threadfence) writes back / invalidates the XCD's L2 on gfx950 --
  // a thousand wavefronts doing that cost the single-launch training-iteration backward 20 us (measured: 57 instead of 36 us) for nothing.
  // What this relies on, spelled out: __syncthreads() IS fence(release, workgroup) + s_barrier + fence(acquire, workgroup) over all address spaces
  // (hip/amd_detail/amd_device_functions.h); LLVM's AMDGPU memory model makes workgroup-scope release / acquire on gfx90a+ sufficient for global memory when the
  // workgroup's waves share one CU's L1 -- i.e. NOT in threadgroup-split mode (the build never passes -mtgsplit; every kernel is __launch_bounds__(64): a launch with
  // more than one wavefront per workgroup fails) -- and the re-read must be a VECTOR load: the compiler only scalarises loads it can prove unclobbered, and these
  // addresses are stored to by the kernel itself (th_out / the gradient rows are plain non-const pointers).
  __device__ __forceinline__ void mem_sync() const { __syncthreads(); }
  __device__ __forceinline__ char* lds() const { return lds_; }
  __device__ __forceinline__ char* stash() const { return stash_; }
  __device__ __forceinline__ char* wb_lds() const { return wb_; }
  // the table inside the kernel-argument segment (GnParams is the first argument of every kernel)
  __device__ __forceinline__ const double* wb_source(const dgp::GnParams&) const {
    return (const double*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(dgp::GnParams, wb_tab));
  }
  __device__ __forceinline__ void lds_sync() const { __syncthreads(); }      // one wavefront per workgroup
  __device__ __forceinline__ int lane() const { return (int)(threadIdx.x & 63u); }
  // one wavefront per workgroup: every launch uses block(64).  (blockDim.x is a scalar load from the HIDDEN kernel arguments, behind the lines
  // warm_kernarg touches: one more exposed miss in front of the first row load -- 0.25 us of the d = 4 step.)
  __device__ __forceinline__ int wave() const { return (int)blockIdx.x; }
  __device__ __forceinline__ int fetch_i(int v, int src) const { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
  __device__ __forceinline__ double fetch(double v, int src) const {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
    hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
    return __hiloint2double(hi, lo);
  }
  // value of lane (l - S) / (l + S) inside the 16-lane DPP row; lanes without such a neighbour read 0 (bound_ctrl)
  template <int S>
  __device__ __forceinline__ double row_from_lower(double v) const {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + S, 0xf, 0xf, true);      // row_shr:S
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + S, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  }
  template <int S>
  __device__ __forceinline__ double row_from_upper(double v) const {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + S, 0xf, 0xf, true);      // row_shl:S
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + S, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  }
  // value of the lane N positions below (cyclically) inside the 16-lane DPP row (row_ror:N; N = 8 swaps the two halves)
  template <int N>
  __device__ __forceinline__ double row_rotate(double v) const {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  }
  __device__ __forceinline__ bool any(bool pred) const { return __any(pred ? 1 : 0) != 0; }
  __device__ __forceinline__ uint64_t ballot(bool pred) const { return __ballot(pred ? 1 : 0); }      // v_cmp into an SGPR pair
  // XCD (accelerator complex die) this wavefront runs on, 0..7 on MI355X: HW_REG_XCC_ID (id 20), bits [3:0]
  __device__ __forceinline__ int xcc_id() const { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf); }
  // xcd_local: the target is only touched by wavefronts of THIS XCD during the kernel (a per-XCD partial buffer), so the
  // read-modify-write may resolve in the XCD's own L2 (workgroup-scope atomic: no sc1, the line stays in L2) instead of at
  // the memory side, where device-scope atomics from eight non-coherent L2s serialise.
  template <typename T>
  __device__ __forceinline__ void atomic_add(T* p, T v, bool xcd_local = false) const {
    if (xcd_local) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

// Wavefronts per SIMD the register allocator must leave room for.  One wavefront per SIMD issues at most one VALU
// instruction every 4 cycles whatever its width; two co-resident ones fill each other's dependency stalls.  Forcing room
// for two costs ~40 spilled registers on the d = 4, C <= 2 kernels, which pays only where a launch has more wavefronts
// than the chip has SIMDs, i.e. for the shapes with 32 or 64 lanes per trajectory (measured at B = 4096, n = 64:
// (32,2) 33.6 -> 23.0 us, (64,1) 47.9 -> 44.3 us; but (16,2) at n = 32, one wavefront per SIMD, 10.9 -> 12.2 us).
#ifndef DGP_FORCE_WPS
#define DGP_FORCE_WPS 0       // tuning aid: -DDGP_FORCE_WPS=2 forces the two-wavefront register budget on every kernel
#endif
template <int DOF, int LPT, int C, int MODE>
struct WavesPerSimd { static constexpr int value = DGP_FORCE_WPS ? DGP_FORCE_WPS : (DOF == 2 && C <= 2 && LPT >= 32 && MODE != dgp::MODE_SOLVE) ? 2 : 1; };

// Touch every 64-byte line of the kernel-argument segment with a scalar load at kernel entry.  The argument block is
// ~1.3 KB and the compiler loads its fields where they are first used, one exposed scalar-cache miss (a fresh address every
// launch, so always a miss) per line, serially, in the middle of the program; issued together up front the misses overlap each
// other and the first global loads, and the later loads hit the scalar cache.  Results are discarded.
// (The loads target one fixed, clobbered SGPR and are waited for inside the same asm statement: scalar loads may return
// out of order, so a discarded load must never be outstanding while the compiler reuses its destination register.)
template <int LINES>
__device__ __forceinline__ void warm_kernarg_lines(const void* ka) {
  // ONE asm statement: the loads and their wait cannot be separated by anything the compiler schedules (a scalar load still
  // in flight would clobber whatever the compiler put into s90 in between)
  asm volatile(".set dgp_ka_off, 0\n\t.rept %1\n\ts_load_dword s90, %0, dgp_ka_off\n\t.set dgp_ka_off, dgp_ka_off + 64\n\t.endr\n\t"
               "s_waitcnt lgkmcnt(0)" :: "s"(ka), "n"(LINES) : "s90", "memory");
}
template <int BYTES>
__device__ __forceinline__ void warm_kernarg() {
#if defined(__HIP_DEVICE_COMPILE__)
  warm_kernarg_lines<(BYTES + 63) / 64>((const void*)__builtin_amdgcn_kernarg_segment_ptr());
#endif
}

// The same warm-up with the segment pointer coming back THROUGH the asm statement: every argument field read through the returned pointer is
// read behind the warm-up and -- the point here -- is re-loadable: the fused loop (MODE_SOLVE) otherwise keeps a hundred argument scalars live in
// SGPRs across its iterations and spills them into VGPR lanes (124 spilled SGPRs in <2,16,4,float,SOLVE,Woodbury>; 22 this way).
template <int LINES, int L1 = 0, int N1 = 0>      // lines [0, LINES) and [L1, L1 + N1) of the segment
__device__ __forceinline__ const char* warm_kernarg_laundered() {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const char* KP;
  KP ka = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile(".set dgp_ka_off, 0\n\t.rept %1\n\ts_load_dword s90, %0, dgp_ka_off\n\t.set dgp_ka_off, dgp_ka_off + 64\n\t.endr\n\t"
               ".set dgp_ka_off, %2\n\t.rept %3\n\ts_load_dword s90, %0, dgp_ka_off\n\t.set dgp_ka_off, dgp_ka_off + 64\n\t.endr\n\t"
               "s_waitcnt lgkmcnt(0)" : "+s"(ka) : "n"(LINES), "n"(L1 * 64), "n"(N1) : "s90", "memory");
  return (const char*)ka;       // (address-space cast: the loads stay scalar loads from the constant address space)
#else
  return nullptr;
#endif
}

#ifndef DGP_LAUNDER_KERNARG
// measured (profiles/r03_kernel_variants_late.txt, 10 fused iterations at B = 4096): d = 4 Woodbury 74.6 -> 70.8 us, d = 4 block elimination 85.1 -> 80.7 us,
// d = 6 Woodbury 219 -> 213 us; per-state Kronecker 98.4 -> 100.7 us (worse: left alone, like the general kernels and every STEP kernel -- 9.86 against 9.82 us)
#define DGP_LAUNDER_KERNARG(DOF, MODE, QK) ((MODE) == dgp::MODE_SOLVE && (dgp::is_wb(QK) || (QK) == dgp::QK_STATIC))
#endif
// QK: kernel variant by covariance representation, dgp::QK_* (static: the constant GP blocks are scalar operands; see gn_lane.h).
// TL: which twin of the source this translation unit is (DGP_TL: tiled grids; DGP_STEP_ERRS: step kernels with the errors epilogue; gn_lane.h) -- only there to give
// the twin units' kernels their own symbols
template <int DOF, int LPT, int C, typename IO, int MODE, int QK, int TL = DGP_TL + 2 * DGP_STEP_ERRS>
__global__ void __launch_bounds__(64, (WavesPerSimd<DOF, LPT, C, MODE>::value)) gn_kernel(const dgp::GnParams p_arg) {
  // (the Woodbury table behind the warmed lines is read with vector loads)
  constexpr bool kLaunder = DGP_LAUNDER_KERNARG(DOF, MODE, QK);
  const dgp::GnParams* pp = &p_arg;
  if constexpr (kLaunder) pp = (const dgp::GnParams*)warm_kernarg_laundered<((int)offsetof(dgp::GnParams, wb_tab) + 63) / 64>();
  else warm_kernarg<(int)offsetof(dgp::GnParams, wb_tab)>();
  const dgp::GnParams& p = *pp;
  // STEP: staging block of the full-line th / dtheta accesses; SOLVE: the fp64 trajectory rows parked between iterations
  // (+ an S_k^-1 stash where SinvStashBlocks asks for one -- currently only the backward kernel does: in STEP mode it would ALIAS
  // the staging block, th being loaded before the sweep and dtheta stored after the recovery; in SOLVE mode follow the trajectory)
  constexpr int kRows = (MODE == dgp::MODE_SOLVE) ? dgp::WaveStore<double, C, 2 * DOF>::kLdsBytes : dgp::WaveStore<IO, C, 2 * DOF>::kLdsBytes;
  constexpr int kStashS = dgp::is_wb(QK) ? 0 : dgp::SinvStash<2 * DOF, dgp::SinvStashBlocks<2 * DOF, C, MODE>::value>::kBytes;      // (the Woodbury kernels keep no S_k^-1)
  constexpr int kParkB = (dgp::is_wb(QK) && dgp::WbParks<DOF, MODE>::value != 0) ? dgp::LdsPark<dgp::WbParkCells<DOF, dgp::WbParks<DOF, MODE>::value>::value>::kBytes : 0;     // gn_woodbury.h, PARK
  constexpr int kStash = kStashS > kParkB ? kStashS : kParkB;
  constexpr int kLds = (MODE == dgp::MODE_SOLVE) ? kRows + kStash : (kRows > kStash ? kRows : kStash);
  constexpr int kWb = dgp::is_wb(QK) ? dgp::kWbLdsBytes : 0;
  __shared__ __attribute__((aligned(16))) char lds[kLds + kWb];
  DevCtx cx;
  cx.lds_ = lds;
  cx.stash_ = (MODE == dgp::MODE_SOLVE) ? lds + kRows : lds;
  cx.wb_ = lds + kLds;
  cx.chain_ = nullptr;
  dgp::gn_lane_program<DOF, LPT, C, IO, MODE, QK>(p, cx);
}

template <int DOF, int LPT, int C, typename IO, int QK, bool CHAIN = false, int TL = DGP_TL + 2 * DGP_STEP_ERRS>
__global__ void __launch_bounds__(64) gn_backward_kernel(const dgp::GnParams p_arg, const dgp::GnGradParams g_arg) {
  // d = 4 static-covariance kernels: both argument structs read through the laundered pointer, as the fused loop does (static backward 15.0 -> 14.6 us;
  // the per-state kernel gets slower that way, 19.4 -> 19.7 us, and d = 6 was not measured: both keep the by-value reads)
  constexpr bool kLaunder = (DOF == 2 || CHAIN) && (dgp::is_wb(QK) || QK == dgp::QK_STATIC);
  constexpr int kGOff = (int)sizeof(dgp::GnParams);                          // GnGradParams follows GnParams (both 8-byte aligned)
  static_assert(sizeof(dgp::GnParams) % 8 == 0 && alignof(dgp::GnGradParams) <= 8, "argument layout");
  const dgp::GnParams* pp = &p_arg;
  const dgp::GnGradParams* gg = &g_arg;
  if constexpr (kLaunder) {
    const char* ka = warm_kernarg_laundered<((int)offsetof(dgp::GnParams, wb_tab) + 63) / 64, kGOff / 64, (kGOff % 64 + (int)sizeof(dgp::GnGradParams) + 63) / 64>();
    pp = (const dgp::GnParams*)ka; gg = (const dgp::GnGradParams*)(ka + kGOff);
  } else {
    warm_kernarg<(int)offsetof(dgp::GnParams, wb_tab)>();
    warm_kernarg_lines<((int)sizeof(dgp::GnGradParams) + 63) / 64>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(dgp::GnParams));
  }
  const dgp::GnParams& p = *pp;
  const dgp::GnGradParams& g = *gg;
  constexpr int kPairBytes = 64 * 2 * (int)sizeof(dgp::TapEntry<IO>);        // sdf_scatter_pairs staging: two tap entries per lane
  constexpr int kRowBytes = dgp::WaveStore<IO, C, 2 * DOF>::kLdsBytes;
  // the adjoint solve's LDS: the S_k^-1 stash of the block elimination, or the parked recovery state of the Woodbury elimination (gn_woodbury.h, PARK)
  constexpr int kStash = dgp::is_wb(QK) ? (dgp::BwdParks<DOF, CHAIN>::value != 0
                                               ? dgp::LdsPark<dgp::WbParkCells<DOF, dgp::BwdParks<DOF, CHAIN>::value>::value>::kBytes : 0)
                                        : dgp::SinvStash<2 * DOF, dgp::SinvStashBlocks<2 * DOF, C, dgp::MODE_BACKWARD_SOLVE>::value>::kBytes;
  constexpr int kMax = kPairBytes > kRowBytes ? kPairBytes : kRowBytes;      // (the stash is dead by the time the pair staging is used: aliased)
  constexpr int kAll = kMax > kStash ? kMax : kStash;
  constexpr int kWb = dgp::is_wb(QK) ? dgp::kWbLdsBytes : 0;
  constexpr int kChain = CHAIN ? dgp::ChainSlots<C, 2 * DOF>::kBytes : (DOF == 2 ? dgp::FoldSlots<C, 2 * DOF>::kBytes : 0);      // (d = 4 single step: the prologue's hand-over slots)
  __shared__ __attribute__((aligned(16))) char lds[kAll + kWb + kChain];
  DevCtx cx;
  cx.lds_ = lds;
  cx.stash_ = lds;
  cx.wb_ = lds + kAll;
  cx.chain_ = lds + kAll + kWb;
  dgp::gn_backward_lane_program<DOF, LPT, C, IO, QK, CHAIN>(p, g, cx);
}

// Long trajectories (n > 256, gn_long.h): one trajectory per wavefront, rows per lane a runtime value, dynamic LDS.
template <int DOF, typename IO, int MODE>
__global__ void __launch_bounds__(64) gn_long_kernel(const dgp::GnParams p) {
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  DevCtx cx;
  cx.lds_ = nullptr; cx.stash_ = nullptr; cx.wb_ = nullptr; cx.chain_ = nullptr;
  cx.long_ = dyn_lds;
  dgp::gn_long_program<DOF, IO, MODE>(p, cx);
}
template <int DOF, typename IO>
__global__ void __launch_bounds__(64) gn_long_backward_kernel(const dgp::GnParams p, const dgp::GnGradParams g) {
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  DevCtx cx;
  cx.lds_ = nullptr; cx.stash_ = nullptr; cx.wb_ = nullptr; cx.chain_ = nullptr;
  cx.long_ = dyn_lds;
  dgp::gn_long_backward_program<DOF, IO>(p, g, cx);
}

// every (LPT, C) of dgp_host::shape_supported; the tiled units (DGP_TL == 1) hold the two shapes dgp_host::choose_shape picks for tiled grids
#if DGP_TL == 1 || DGP_STEP_ERRS == 1
#define DGP_FOR_EACH_SHAPE(X) X(16, 4) X(32, 4)
#else
#define DGP_FOR_EACH_SHAPE(X) X(16, 1) X(32, 1) X(64, 1) X(16, 2) X(32, 2) X(64, 2) X(16, 4) X(32, 4) X(64, 4)
#endif

// mode: dgp::MODE_* or MODE_BACKWARD
enum { MODE_BACKWARD = 3, MODE_CHAIN = 4 };      // MODE_CHAIN: dgp_gn_solve_backward (the chain kernels, static covariances)

// The kernels are spread over translation units (gn_inst.hip, one per (dof, io dtype, group)) so that they build in
// parallel.  Group of a launch: static-covariance STEP / SOLVE; the general-covariance STEP / SOLVE plus EVAL; the static and
// general backward kernels; everything for per-state Q_c^-1 tensors (QK_KRON: STEP, SOLVE, backward).
enum { GROUP_STATIC = 0, GROUP_GENERIC = 1, GROUP_BACKWARD = 2, GROUP_KRON = 3, GROUP_CHAIN = 4, NUM_GROUPS = 5 };
inline int launch_group(int mode, const dgp::GnParams& p) {
  if (mode == dgp::MODE_EVAL) return GROUP_GENERIC;
  if (mode == MODE_CHAIN) return GROUP_CHAIN;
  const int qk = dgp::kernel_variant(p);
  if (qk == dgp::QK_SCALED) return mode == MODE_BACKWARD ? GROUP_BACKWARD : GROUP_STATIC;      // (STEP and the single-step backward, host-checked)
  if (qk == dgp::QK_KRON) return GROUP_KRON;
  if (mode == MODE_BACKWARD) return GROUP_BACKWARD;
  return qk == dgp::QK_STATIC ? GROUP_STATIC : GROUP_GENERIC;
}

// (TL is part of the signature: the standard and the tiled unit of one (dof, io dtype, group) must not share ONE weak host instantiation of this template --
//  the linker would keep either, and both launchers would then start the same kernels)
template <int DOF, typename IO, int GROUP, int TL = DGP_TL + 2 * DGP_STEP_ERRS>
hipError_t launch_typed(DgpShape sh, int mode, const dgp::GnParams& p, const dgp::GnGradParams* g, hipStream_t s) {
  const int tpw = 64 / sh.lpt;
  const dim3 grid((unsigned)((p.B + tpw - 1) / tpw)), block(64);
  const bool qstat = dgp::use_static_kernels(p);
  if (launch_group(mode, p) != GROUP) return hipErrorInvalidValue;
  // dgp_time_next_launch(): this launch records its own begin / end on the caller's events (hipExtLaunchKernelGGL); one-shot
  dgp_host::LaunchEvents& le = dgp_host::launch_events();
  const hipEvent_t ev0 = (hipEvent_t)le.start, ev1 = (hipEvent_t)le.stop;
  const bool timed = ev0 && ev1;
  le.start = le.stop = nullptr;
#define DGP_LAUNCH(K)                                                               \
  do {                                                                              \
    if (timed) hipExtLaunchKernelGGL(K, grid, block, 0, s, ev0, ev1, 0, p);         \
    else hipLaunchKernelGGL(K, grid, block, 0, s, p);                               \
  } while (0)
#define DGP_LAUNCH_BWD(K)                                                           \
  do {                                                                              \
    if constexpr (DGP_STEP_ERRS == 0) {                                             \
      if (timed) hipExtLaunchKernelGGL(K, grid, block, 0, s, ev0, ev1, 0, p, *g);   \
      else hipLaunchKernelGGL(K, grid, block, 0, s, p, *g);                         \
    } else return hipErrorInvalidValue;                                             \
  } while (0)
  /* the step-errors twins hold MODE_STEP kernels only: every other launch is a discarded statement there (no instantiation) */
#define DGP_LAUNCH_NOSTEP(K)                                                        \
  do {                                                                              \
    if constexpr (DGP_STEP_ERRS == 0) DGP_LAUNCH(K); else return hipErrorInvalidValue; \
  } while (0)
#define DGP_CASE(L, CC)                                                                                                   \
  if (sh.lpt == L && sh.c == CC) {                                                                                         \
    if constexpr (GROUP == GROUP_STATIC) {                                                                                 \
      if (dgp::kernel_variant(p) == dgp::QK_SCALED) {                                                                      \
        if (mode != dgp::MODE_STEP) return hipErrorInvalidValue;                                                           \
        DGP_LAUNCH((gn_kernel<DOF, L, CC, IO, dgp::MODE_STEP, dgp::QK_SCALED>));                                           \
        return hipGetLastError();                                                                                          \
      }                                                                                                                    \
      if constexpr (CC == 4) {                                                                                             \
        if (dgp::wb_applies(p, L, CC)) {                                                                                   \
          if (p.n == L * CC) {                                                                                             \
            if (mode == dgp::MODE_STEP) DGP_LAUNCH((gn_kernel<DOF, L, CC, IO, dgp::MODE_STEP, dgp::QK_WB>));               \
            else DGP_LAUNCH_NOSTEP((gn_kernel<DOF, L, CC, IO, dgp::MODE_SOLVE, dgp::QK_WB>));                                     \
          } else {                                                                                                         \
            if (mode == dgp::MODE_STEP) DGP_LAUNCH((gn_kernel<DOF, L, CC, IO, dgp::MODE_STEP, dgp::QK_WBR>));              \
            else DGP_LAUNCH_NOSTEP((gn_kernel<DOF, L, CC, IO, dgp::MODE_SOLVE, dgp::QK_WBR>));                                    \
          }                                                                                                                \
          return hipGetLastError();                                                                                        \
        }                                                                                                                  \
      }                                                                                                                    \
      /* (round 6: the d = 6 block-elimination twin is back -- its wrong results were the exec-join miscompile the build now repairs, profiles/r06_compiler_fault.md; */ \
      /*  -DDGP_EXCLUDE_REPAIRED_TWINS=1 restores the round-5 exclusions of the two twins for the reproducer builds) */ \
      if constexpr (DGP_STEP_ERRS == 1 && DOF == 3 && DGP_EXCLUDE_REPAIRED_TWINS) return hipErrorInvalidValue; \
      else if (mode == dgp::MODE_STEP) DGP_LAUNCH((gn_kernel<DOF, L, CC, IO, dgp::MODE_STEP, dgp::QK_STATIC>));            \
      else DGP_LAUNCH_NOSTEP((gn_kernel<DOF, L, CC, IO, dgp::MODE_SOLVE, dgp::QK_STATIC>));                                       \
    } else if constexpr (GROUP == GROUP_GENERIC) {                                                                         \
      if constexpr (DGP_STEP_ERRS == 1 && L == 32 && sizeof(IO) == 4 && DGP_EXCLUDE_REPAIRED_TWINS) return hipErrorInvalidValue;      \
      else if (mode == dgp::MODE_STEP) DGP_LAUNCH((gn_kernel<DOF, L, CC, IO, dgp::MODE_STEP, dgp::QK_GENERAL>));           \
      else if (mode == dgp::MODE_SOLVE) DGP_LAUNCH_NOSTEP((gn_kernel<DOF, L, CC, IO, dgp::MODE_SOLVE, dgp::QK_GENERAL>));  \
      else DGP_LAUNCH_NOSTEP((gn_kernel<DOF, L, CC, IO, dgp::MODE_EVAL, dgp::QK_GENERAL>));                                \
    } else if constexpr (GROUP == GROUP_CHAIN) {                                                                           \
      if (p.qc_mode != dgp::QC_STATIC) return hipErrorInvalidValue;                                                        \
      if (!qstat) {      /* round 6: a NON-DIAGONAL static Q_c_inv -- the general-covariance chain kernels (no covariance gradient: static) */ \
        DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_GENERAL, true>));                                       \
        return hipGetLastError();                                                                                          \
      }                                                                                                                    \
      if constexpr (CC == 4) {                                                                                             \
        if (dgp::wb_applies(p, L, CC)) {                                                                                   \
          if (p.n == L * CC) DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_WB, true>));                       \
          else DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_WBR, true>));                                    \
          return hipGetLastError();                                                                                        \
        }                                                                                                                  \
      }                                                                                                                    \
      DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_STATIC, true>));                                          \
    } else if constexpr (GROUP == GROUP_KRON) {                                                                            \
      if (mode == dgp::MODE_STEP) DGP_LAUNCH((gn_kernel<DOF, L, CC, IO, dgp::MODE_STEP, dgp::QK_KRON>));                   \
      else if (mode == dgp::MODE_SOLVE) DGP_LAUNCH_NOSTEP((gn_kernel<DOF, L, CC, IO, dgp::MODE_SOLVE, dgp::QK_KRON>));     \
      else DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_KRON>));                                             \
    } else {                                                                                                               \
      if (dgp::kernel_variant(p) == dgp::QK_SCALED) {                                                                      \
        DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_SCALED>));                                              \
        return hipGetLastError();                                                                                          \
      }                                                                                                                    \
      if constexpr (CC == 4) {                                                                                             \
        if (qstat && dgp::wb_applies(p, L, CC)) {                                                                          \
          if (p.n == L * CC) DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_WB>));                             \
          else DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_WBR>));                                          \
          return hipGetLastError();                                                                                        \
        }                                                                                                                  \
      }                                                                                                                    \
      if (qstat) DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_STATIC>));                                     \
      else DGP_LAUNCH_BWD((gn_backward_kernel<DOF, L, CC, IO, dgp::QK_GENERAL>));                                          \
    }                                                                                                                      \
    return hipGetLastError();                                                                                              \
  }
  DGP_FOR_EACH_SHAPE(DGP_CASE)
#undef DGP_CASE
#undef DGP_LAUNCH
#undef DGP_LAUNCH_NOSTEP
#undef DGP_LAUNCH_BWD
  return hipErrorInvalidValue;
}

}  // namespace dgp_dev

// One translation unit per (dof, io dtype, group) -- see gn_inst.hip.
typedef hipError_t (*DgpLaunchFn)(DgpShape, int, const dgp::GnParams&, const dgp::GnGradParams*, hipStream_t);
#define DGP_DECL_INST(d, t) \
  hipError_t dgp_launch_##d##_##t##_g0(DgpShape, int, const dgp::GnParams&, const dgp::GnGradParams*, hipStream_t); \
  hipError_t dgp_launch_##d##_##t##_g1(DgpShape, int, const dgp::GnParams&, const dgp::GnGradParams*, hipStream_t); \
  hipError_t dgp_launch_##d##_##t##_g2(DgpShape, int, const dgp::GnParams&, const dgp::GnGradParams*, hipStream_t); \
  hipError_t dgp_launch_##d##_##t##_g3(DgpShape, int, const dgp::GnParams&, const dgp::GnGradParams*, hipStream_t); \
  hipError_t dgp_launch_##d##_##t##_g4(DgpShape, int, const dgp::GnParams&, const dgp::GnGradParams*, hipStream_t);
DGP_DECL_INST(2, f32) DGP_DECL_INST(2, f64) DGP_DECL_INST(3, f32) DGP_DECL_INST(3, f64)
// ... and their tiled twins (gn_inst.hip with -DDGP_TL=1): dgp_launch_<dof>t_<io>_g<group>
DGP_DECL_INST(2t, f32) DGP_DECL_INST(2t, f64) DGP_DECL_INST(3t, f32) DGP_DECL_INST(3t, f64)
// ... and the step-errors twins (gn_inst.hip with -DDGP_STEP_ERRS=1; groups 0 and 3 are built): dgp_launch_<dof>e_<io>_g<group>
DGP_DECL_INST(2e, f32) DGP_DECL_INST(2e, f64) DGP_DECL_INST(3e, f32) DGP_DECL_INST(3e, f64)
#undef DGP_DECL_INST
// gn_long_inst.hip: the long-trajectory kernels of every (dof, io dtype); mode: dgp::MODE_* or dgp_dev::MODE_BACKWARD
hipError_t dgp_launch_long(int dof, bool f64, int mode, const dgp::GnParams& p, const dgp::GnGradParams* g, hipStream_t s);
