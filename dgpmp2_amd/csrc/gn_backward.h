// gn_backward.h -- backward (adjoint) lane program of the fused Gauss-Newton step.  See DESIGN.md.
#pragma once
#include "gn_lane.h"

namespace dgp {

static const bool kBackwardImplemented = false;

struct GnGradParams {
  const void *g_dtheta, *g_err_ext;
  void *g_th, *g_start, *g_goal, *g_sdf, *g_qc, *g_obs_w, *g_eps;
  int64_t g_sdf_bstride;
};

template <int DOF, int LPT, typename IO, typename Ctx>
DGP_HD void gn_backward_lane_program(const GnParams& p, const GnGradParams& g, Ctx& cx) {
  (void)p; (void)g; (void)cx;
}

}  // namespace dgp
