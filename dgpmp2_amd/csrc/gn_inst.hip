// gn_inst.hip -- kernel instantiations for ONE (dof, io dtype, kernel group), selected with
//   -DDGP_INST_DOF=2|3 -DDGP_INST_F64=0|1 -DDGP_INST_GROUP=0|1|2|3      (groups: dgp_dev::GROUP_*)
#include "gn_device.h"

#if DGP_INST_F64
typedef double inst_io_t;
#define DGP_INST_NAME3(d, g) dgp_launch_##d##_f64_g##g
#else
typedef float inst_io_t;
#define DGP_INST_NAME3(d, g) dgp_launch_##d##_f32_g##g
#endif
#define DGP_INST_NAME(d, g) DGP_INST_NAME3(d, g)

hipError_t DGP_INST_NAME(DGP_INST_DOF, DGP_INST_GROUP)(DgpShape sh, int mode, const dgp::GnParams& p, const dgp::GnGradParams* g,
                                                       hipStream_t s) {
  return dgp_dev::launch_typed<DGP_INST_DOF, inst_io_t, DGP_INST_GROUP>(sh, mode, p, g, s);
}
