// Phase timing of gn_kernel<2,16,4,float,STEP,static> on the benchmark workload (B trajectories, n = 64, 256x256 SDF):
// the same kernel cut short after a phase (-DDGP_PHASE_STOP=k, see gn_lane.h), timed back to back with HIP events.
//   k = 1: loads of th / start / goal + cross-lane neighbours      k = 2: + SDF taps, obstacle factors, GP right-hand sides
//   k = 3: + local elimination and separator row                   k = 4: + PCR                 (undefined: whole kernel)
// usage (GPU box):  for k in 1 2 3 4 0; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DPROBE_STOP=$k \
//                       -Idgpmp2_amd/csrc profiles/tools/phase_probe.hip -o /tmp/pp$k && /tmp/pp$k; done
#if PROBE_STOP > 0
#define DGP_PHASE_STOP PROBE_STOP
#endif
#ifndef PROBE_QK
#define PROBE_QK 1      // dgp::QK_STATIC; -DPROBE_QK=3: the Woodbury kernel (dgp::QK_WB)
#endif
#include "gn_device.h"
#include <vector>
#include <random>
#include <algorithm>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, n = 64, G = 256;
  DgpConfig cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.struct_size = sizeof(cfg); cfg.dof = 2; cfg.nlinks = 1; cfg.num_states = n; cfg.io_dtype = DGP_F32;
  cfg.total_time_sec = 10.0; cfg.x_lims[0] = -5; cfg.x_lims[1] = 5; cfg.y_lims[0] = -5; cfg.y_lims[1] = 5;
  cfg.K_s = 1e-3; cfg.K_g = 1e-3; cfg.cost_sigma = 0.2; cfg.epsilon_dist = 0.6; cfg.sphere_radius = 0.3; cfg.reg = 1e-4;
  cfg.Q_c_inv[0] = 1.0; cfg.Q_c_inv[3] = 1.0;
  DgpHandle* h = nullptr;
  if (dgp_host::create(&cfg, &h) != DGP_OK) { printf("create failed: %s\n", dgp_host::err_buf()); return 1; }
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> th((size_t)B * n * 4), st((size_t)B * 4), go((size_t)B * 4), sdf((size_t)G * G);
  for (int b = 0; b < B; ++b) {
    float sx = 4 * U(rng), sy = 4 * U(rng), gx = 4 * U(rng), gy = 4 * U(rng);
    st[b * 4] = sx; st[b * 4 + 1] = sy; go[b * 4] = gx; go[b * 4 + 1] = gy;
    for (int i = 0; i < n; ++i) {
      float t = (float)i / (n - 1);
      float* x = &th[((size_t)b * n + i) * 4];
      x[0] = sx + t * (gx - sx) + 0.05f * U(rng); x[1] = sy + t * (gy - sy) + 0.05f * U(rng);
      x[2] = (gx - sx) / 10.f; x[3] = (gy - sy) / 10.f;
    }
  }
  for (int r = 0; r < G; ++r)
    for (int c = 0; c < G; ++c) {            // distance to a few discs
      float x = -5 + 10.f * (c + 0.5f) / G, y = 5 - 10.f * (r + 0.5f) / G, d = 1e9f;
      const float cs[4][3] = {{-2, 2, 1.0f}, {2, -1, 1.2f}, {0, 0, 0.8f}, {3, 3, 0.7f}};
      for (auto& q : cs) d = fminf(d, sqrtf((x - q[0]) * (x - q[0]) + (y - q[1]) * (y - q[1])) - q[2]);
      sdf[(size_t)r * G + c] = d;
    }
  float *d_th, *d_st, *d_go, *d_sdf, *d_dth, *d_err, *d_eex;
  hipMalloc(&d_th, th.size() * 4 * 10); hipMalloc(&d_st, st.size() * 4); hipMalloc(&d_go, go.size() * 4); hipMalloc(&d_sdf, sdf.size() * 4);
  hipMalloc(&d_dth, th.size() * 4); hipMalloc(&d_err, B * 4); hipMalloc(&d_eex, B * 4);
  for (int c = 0; c < 10; ++c) hipMemcpy(d_th + (size_t)c * th.size(), th.data(), th.size() * 4, hipMemcpyHostToDevice);   // ten input buffers, cycled like bench.py
  hipMemcpy(d_st, st.data(), st.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_go, go.data(), go.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_sdf, sdf.data(), sdf.size() * 4, hipMemcpyHostToDevice);
  DgpSdf sa; sa.data = d_sdf; sa.rows = G; sa.cols = G; sa.batch_stride = 0;
  dgp::GnParams p;
  if (dgp_host::fill_step(h, B, d_th, d_st, d_go, &sa, nullptr, d_dth, d_err, d_eex, nullptr, p) != DGP_OK) { printf("fill failed\n"); return 1; }
  const dim3 grid((B + 3) / 4), block(64);
  int launch_no = 0;
  auto launch = [&]() {
    p.th = d_th + (size_t)(launch_no++ % 10) * th.size();
    hipLaunchKernelGGL((dgp_dev::gn_kernel<2, 16, 4, float, dgp::MODE_STEP, PROBE_QK>), grid, block, 0, 0, p);
  };
  for (int i = 0; i < 2000; ++i) launch();        // clocks up
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 500;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("stream launches: %.2f us/launch\n", ms * 1e3 / reps);
  {  // the same launches replayed from a hipGraph (one graph = 50 dependent step kernels)
    hipStream_t cs; hipStreamCreate(&cs);
    hipGraph_t graph; hipGraphExec_t exec;
    hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((dgp_dev::gn_kernel<2, 16, 4, float, dgp::MODE_STEP, PROBE_QK>), grid, block, 0, cs, p);
    hipStreamEndCapture(cs, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(exec, cs);
    hipStreamSynchronize(cs);
    hipEventRecord(e0, cs);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(exec, cs);
    hipEventRecord(e1, cs); hipEventSynchronize(e1);
    float gms; hipEventElapsedTime(&gms, e0, e1);
    printf("hipGraph replay : %.2f us/launch\n", gms * 1e3 / 500);
  }
#if defined(DGP_PHASE_STAMPS)
  {  // in-kernel timeline: two consecutive launches in the middle of a back-to-back series write their stamps
    const int waves = (B + 3) / 4;
    unsigned long long *d_s0, *d_s1;
    hipMalloc(&d_s0, (size_t)waves * 128); hipMalloc(&d_s1, (size_t)waves * 128);
    dgp::GnParams p0 = p, p1 = p;
    p0.err_hist = d_s0; p1.err_hist = d_s1;
    for (int i = 0; i < 400; ++i) {
      const dgp::GnParams& q = (i == 200) ? p0 : ((i == 201) ? p1 : p);
      hipLaunchKernelGGL((dgp_dev::gn_kernel<2, 16, 4, float, dgp::MODE_STEP, PROBE_QK>), grid, block, 0, 0, q);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> s0((size_t)waves * 16), s1((size_t)waves * 16);
    hipMemcpy(s0.data(), d_s0, s0.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(s1.data(), d_s1, s1.size() * 8, hipMemcpyDeviceToHost);
    const char* names[11] = {"entry", "th/start/goal loaded", "taps + factors done", "local elimination done", "PCR + recovery done", "stores issued", "stores acked", "-", "tap loads issued", "GP right-hand sides done", "taps arrived"};
    unsigned long long first0 = ~0ull, last0 = 0, first1 = ~0ull;
    for (int w = 0; w < waves; ++w) { first0 = std::min(first0, s0[w * 16]); last0 = std::max(last0, s0[w * 16 + 6]); first1 = std::min(first1, s1[w * 16]); }
    printf("launch k: first entry -> last ack %.2f us;  gap to first entry of launch k+1: %.2f us\n", (last0 - first0) * 0.01, ((double)first1 - (double)last0) * 0.01);
    for (int sl : {0, 1, 8, 9, 10, 2, 3, 4, 5, 6}) {
      std::vector<double> v;
      for (int w = 0; w < waves; ++w) v.push_back((s0[w * 16 + sl] - first0) * 0.01);
      std::sort(v.begin(), v.end());
      printf("  %-24s min %5.2f  median %5.2f  max %5.2f us after the first wavefront's entry\n", names[sl], v.front(), v[v.size() / 2], v.back());
    }
    // per-wavefront phase DURATIONS, and their medians split by XCD (workgroup id mod 8)
    const int seq[10] = {0, 1, 8, 9, 10, 2, 3, 4, 5, 6};
    for (int q = 1; q < 10; ++q) {
      std::vector<double> v;
      std::vector<std::vector<double>> byx(8);
      for (int w = 0; w < waves; ++w) {
        const double dur = ((double)s0[w * 16 + seq[q]] - (double)s0[w * 16 + seq[q - 1]]) * 0.01;
        v.push_back(dur); byx[w % 8].push_back(dur);
      }
      std::sort(v.begin(), v.end());
      printf("  duration -> %-24s min %5.2f  p10 %5.2f  median %5.2f  p90 %5.2f  max %5.2f   per-XCD medians:", names[seq[q]], v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
      for (int x = 0; x < 8; ++x) { std::sort(byx[x].begin(), byx[x].end()); printf(" %.2f", byx[x][byx[x].size() / 2]); }
      printf("\n");
    }
  }
#endif
  std::vector<float> out(16);
  hipMemcpy(out.data(), d_dth, 64, hipMemcpyDeviceToHost);
  printf("stop_after=%d  B=%d  %.2f us/launch   (dtheta[0..3] = %g %g %g %g)\n", PROBE_STOP, B, ms * 1e3 / reps, out[0], out[1], out[2], out[3]);
  return 0;
}
