// gn_inst.hip -- kernel instantiations for ONE (dof, io dtype, kernel group), selected with
//   -DDGP_INST_DOF=2|3 -DDGP_INST_F64=0|1 -DDGP_INST_GROUP=0|1|2|3|4      (groups: dgp_dev::GROUP_*)
//   [-DDGP_TL=1: the same kernels compiled for 4 x 4-tiled grids (gn_lane.h: DGP_TL), the two launch shapes with four states per lane; launcher dgp_launch_<dof>t_...]
//   [-DDGP_STEP_ERRS=1 (groups 0 and 3): the STEP kernels of those two shapes with the errors-at-th+dtheta epilogue; launcher dgp_launch_<dof>e_...]
#include "gn_device.h"

#if DGP_INST_F64
typedef double inst_io_t;
#define DGP_INST_NAME3(d, g) dgp_launch_##d##_f64_g##g
#else
typedef float inst_io_t;
#define DGP_INST_NAME3(d, g) dgp_launch_##d##_f32_g##g
#endif
#if DGP_STEP_ERRS == 1
#define DGP_INST_NAME_T(d, g) DGP_INST_NAME3(d##e, g)
#define DGP_INST_NAME(d, g) DGP_INST_NAME_T(d, g)
#elif DGP_TL == 1
#define DGP_INST_NAME_T(d, g) DGP_INST_NAME3(d##t, g)
#define DGP_INST_NAME(d, g) DGP_INST_NAME_T(d, g)      // (one more level: the dof macro is expanded before the paste)
#else
#define DGP_INST_NAME(d, g) DGP_INST_NAME3(d, g)
#endif

hipError_t DGP_INST_NAME(DGP_INST_DOF, DGP_INST_GROUP)(DgpShape sh, int mode, const dgp::GnParams& p, const dgp::GnGradParams* g,
                                                       hipStream_t s) {
  return dgp_dev::launch_typed<DGP_INST_DOF, inst_io_t, DGP_INST_GROUP>(sh, mode, p, g, s);
}
