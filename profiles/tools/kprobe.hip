// kprobe.hip -- ONE kernel instantiation of the product source, timed stand-alone (tuning aid: a d = 6 translation unit of the library
// takes minutes to build, one kernel half a minute).  Selected at compile time:
//   -DPROBE_DOF=2|3  -DPROBE_LPT=16|32|64  -DPROBE_C=1|2|4  -DPROBE_QK=0..4 (dgp::QK_*)  -DPROBE_MODE=0|1 (step | fused loop of 10 iterations)
// plus whatever experiment macro the kernel source understands (-DDGP_...).  Workload: B trajectories (argv[1], default 4096) of n = LPT * C
// states; dof 2: the benchmark's configs[1] constants and a 256x256 grid; dof 3: configs[3] (non-holonomic factor, reg 0, eps 0.2, 512x512).
// Prints the launch period over 500 back-to-back launches at steady clocks and a checksum of dtheta (sum |.|, max |.|) so that variants
// of the same kernel can be compared with each other (parity proper is the test-suite's job, on the integrated kernel).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Idgpmp2_amd/csrc -DPROBE_DOF=3 ... profiles/tools/kprobe.hip -o dgpmp2_amd/lib/kprobe_x
#ifndef PROBE_DOF
#define PROBE_DOF 3
#endif
#ifndef PROBE_LPT
#define PROBE_LPT 16
#endif
#ifndef PROBE_C
#define PROBE_C 4
#endif
#ifndef PROBE_QK
#define PROBE_QK 3
#endif
#ifndef PROBE_MODE
#define PROBE_MODE 0
#endif
#include "gn_device.h"
#include <vector>
#include <random>
#include <algorithm>

int main(int argc, char** argv) {
  constexpr int DOF = PROBE_DOF, D = 2 * DOF, n = PROBE_LPT * PROBE_C;
  const int B = argc > 1 ? atoi(argv[1]) : 4096;
  const int G = DOF == 3 ? 512 : 256;
  DgpConfig cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.struct_size = sizeof(cfg); cfg.dof = DOF; cfg.nlinks = 1; cfg.num_states = n; cfg.io_dtype = DGP_F32;
  cfg.total_time_sec = 10.0; cfg.x_lims[0] = -5; cfg.x_lims[1] = 5; cfg.y_lims[0] = -5; cfg.y_lims[1] = 5;
  cfg.K_s = 0.01; cfg.K_g = 0.01; cfg.cost_sigma = 0.01; cfg.sphere_radius = 0.4;
  cfg.epsilon_dist = DOF == 3 ? 0.2 : 0.4; cfg.reg = DOF == 3 ? 0.0 : 0.1;
  for (int i = 0; i < DOF; ++i) cfg.Q_c_inv[i * DOF + i] = 1.0;
  if (DOF == 3) { cfg.flags = DGP_FLAG_NONHOLONOMIC; cfg.K_d = 0.01; }
#if defined(PROBE_VEL)
  cfg.flags |= DGP_FLAG_VEL_LIMITS; cfg.K_v = 0.01; cfg.v_x = 1.0; cfg.v_y = 1.0;
#endif
  DgpHandle* h = nullptr;
  if (dgp_host::create(&cfg, &h) != DGP_OK) { printf("create failed: %s\n", dgp_host::err_buf()); return 1; }
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> th((size_t)B * n * D), st((size_t)B * D, 0.f), go((size_t)B * D, 0.f), sdf((size_t)G * G);
  for (int b = 0; b < B; ++b) {
    const float sx = 4 * U(rng), sy = 4 * U(rng), gx = 4 * U(rng), gy = 4 * U(rng);
    st[b * D] = sx; st[b * D + 1] = sy; go[b * D] = gx; go[b * D + 1] = gy;
    if (DOF == 3) go[b * D + 2] = 1.5707963f;
    for (int i = 0; i < n; ++i) {
      const float t = (float)i / (n - 1);
      float* x = &th[((size_t)b * n + i) * D];
      x[0] = sx + t * (gx - sx) + 0.05f * U(rng); x[1] = sy + t * (gy - sy) + 0.05f * U(rng);
      if (DOF == 3) { x[2] = t * 1.5707963f + 0.02f * U(rng); x[5] = 0.157f; }
      x[DOF] = (gx - sx) / 10.f; x[DOF + 1] = (gy - sy) / 10.f;
    }
  }
  for (int r = 0; r < G; ++r)
    for (int c = 0; c < G; ++c) {            // distance to three discs (the benchmark's circles)
      const float x = -5 + 10.f * c / (G - 1), y = 5 - 10.f * r / (G - 1);
      float d = 1e9f;
      const float cs[3][3] = {{-1.5f, 1.0f, 1.0f}, {1.5f, -1.0f, 1.2f}, {0.f, 2.5f, 0.7f}};
      for (auto& q : cs) d = fminf(d, sqrtf((x - q[0]) * (x - q[0]) + (y - q[1]) * (y - q[1])) - q[2]);
      sdf[(size_t)r * G + c] = d;
    }
  float *d_th, *d_st, *d_go, *d_sdf, *d_dth, *d_err, *d_eex;
  int32_t *d_info, *d_iters;
  const int NBUF = 4;
  hipMalloc(&d_th, th.size() * 4 * NBUF); hipMalloc(&d_st, st.size() * 4); hipMalloc(&d_go, go.size() * 4); hipMalloc(&d_sdf, sdf.size() * 4);
  hipMalloc(&d_dth, th.size() * 4); hipMalloc(&d_err, B * 4 * 16); hipMalloc(&d_eex, B * 4 * 16); hipMalloc(&d_info, B * 4); hipMalloc(&d_iters, B * 4);
  for (int c = 0; c < NBUF; ++c) hipMemcpy(d_th + (size_t)c * th.size(), th.data(), th.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_st, st.data(), st.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_go, go.data(), go.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_sdf, sdf.data(), sdf.size() * 4, hipMemcpyHostToDevice);
  DgpSdf sa; sa.data = d_sdf; sa.rows = G; sa.cols = G; sa.batch_stride = 0;
  // -DPROBE_QK=2 (per-state Q_c^-1, Kronecker kernels) / 0 with -DPROBE_QFULL (full Q^-1): covariance tensors as the learned modes pass them
  DgpCovs cv; cv.qc_mode = DGP_QC_STATIC; cv.qc_inv = nullptr; cv.obs_w = nullptr; cv.eps = nullptr;
  const DgpCovs* covs = nullptr;
#if PROBE_QK == 2 || defined(PROBE_QFULL)
  {
#if defined(PROBE_QFULL)
    const int q = D;
#else
    const int q = DOF;
#endif
    std::vector<float> qc((size_t)B * (n - 1) * q * q, 0.f), ow((size_t)B * n, 1e4f), ep((size_t)B * n, DOF == 3 ? 0.2f : 0.4f);
    const double dt = 10.0 / (n - 1), qa = 12.0 / (dt * dt * dt), qb = -6.0 / (dt * dt), qcc = 4.0 / dt;
    for (size_t f = 0; f < (size_t)B * (n - 1); ++f)
      for (int i = 0; i < q; ++i) {
#if defined(PROBE_QFULL)
        qc[f * q * q + i * q + i] = (float)(i < DOF ? qa : qcc);
        qc[f * q * q + i * q + (i + DOF) % q] = (float)qb;
#else
        qc[f * q * q + i * q + i] = 1.f;
#endif
      }
    float *d_qc, *d_ow, *d_ep;
    hipMalloc(&d_qc, qc.size() * 4); hipMalloc(&d_ow, ow.size() * 4); hipMalloc(&d_ep, ep.size() * 4);
    hipMemcpy(d_qc, qc.data(), qc.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_ow, ow.data(), ow.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_ep, ep.data(), ep.size() * 4, hipMemcpyHostToDevice);
#if defined(PROBE_QFULL)
    cv.qc_mode = DGP_QC_QFULL;
#else
    cv.qc_mode = DGP_QC_PERSTATE;
#endif
    cv.qc_inv = d_qc; cv.obs_w = d_ow; cv.eps = d_ep; covs = &cv;
  }
#endif
  dgp::GnParams p;
#if PROBE_MODE == 1
  if (dgp_host::fill_solve(h, B, d_th, d_st, d_go, &sa, covs, 10, 0.0, d_dth, d_iters, d_err, d_eex, nullptr, d_info, p) != DGP_OK) { printf("fill failed: %s\n", dgp_host::err_buf()); return 1; }
#else
  if (dgp_host::fill_step(h, B, d_th, d_st, d_go, &sa, covs, d_dth, d_err, d_eex, d_info, p) != DGP_OK) { printf("fill failed: %s\n", dgp_host::err_buf()); return 1; }
#endif
  if (dgp::is_wb(PROBE_QK) && !dgp::wb_applies(p, PROBE_LPT, PROBE_C)) { printf("Woodbury kernel does not apply to this configuration\n"); return 1; }
  const int tpw = 64 / PROBE_LPT;
  const int waves = (B + tpw - 1) / tpw;
  const dim3 grid(waves), block(64);
  int launch_no = 0;
  auto launch = [&]() {
    p.th = d_th + (size_t)(launch_no++ % NBUF) * th.size();
    hipLaunchKernelGGL((dgp_dev::gn_kernel<DOF, PROBE_LPT, PROBE_C, float, PROBE_MODE, PROBE_QK>), grid, block, 0, 0, p);
  };
  const int warm = PROBE_MODE == 1 ? 300 : 3000, reps = PROBE_MODE == 1 ? 100 : 500;
  for (int i = 0; i < warm; ++i) launch();        // clocks up
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  launch_no = 0; launch(); hipDeviceSynchronize();
  std::vector<float> out(th.size());
  std::vector<int32_t> info(B);
  hipMemcpy(out.data(), d_dth, out.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(info.data(), d_info, B * 4, hipMemcpyDeviceToHost);
  double s = 0, m = 0; int bad = 0, flagged = 0;
  for (float v : out) { if (!(v == v)) ++bad; else { s += fabs((double)v); m = std::max(m, fabs((double)v)); } }
  for (int v : info) flagged += v != 0;
  printf("kprobe dof=%d shape=(%d,%d) qk=%d mode=%d B=%d : %.2f us/launch%s   checksum sum|x|=%.9e max|x|=%.6e nan=%d info!=0: %d\n", DOF, PROBE_LPT, PROBE_C,
         PROBE_QK, PROBE_MODE, B, best * 1e3 / reps, PROBE_MODE == 1 ? " (10 GN iterations)" : "", s, m, bad, flagged);
  return 0;
}
