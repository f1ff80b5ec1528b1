// gn_inst.hip -- kernel instantiations for ONE (dof, io dtype) pair, selected with -DDGP_INST_DOF=2|3 -DDGP_INST_F64=0|1.
#include "gn_device.h"

#if DGP_INST_F64
typedef double inst_io_t;
#define DGP_INST_NAME2(d) dgp_launch_##d##_f64
#else
typedef float inst_io_t;
#define DGP_INST_NAME2(d) dgp_launch_##d##_f32
#endif
#define DGP_INST_NAME1(d) DGP_INST_NAME2(d)

hipError_t DGP_INST_NAME1(DGP_INST_DOF)(DgpShape sh, int mode, const dgp::GnParams& p, const dgp::GnGradParams* g, hipStream_t s) {
  return dgp_dev::launch_typed<DGP_INST_DOF, inst_io_t>(sh, mode, p, g, s);
}
