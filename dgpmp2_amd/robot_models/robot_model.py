"""Sphere robot models.  In the reference (diff_gpmp2/robot_models/) these classes also carry batched masks and an
identity 'forward kinematics'; in this build the sphere centre (= state[0:2]) and its Jacobian (= I_d[0:2,:]) are fused
into the HIP kernel, so a robot model only has to describe itself."""


class RobotModel(object):
  def __init__(self, dofs, nlinks, wksp_dim, state_dim, sphere_radii=(), batch_size=1, num_traj_states=1, use_cuda=False):
    self.dofs, self.nlinks, self.wksp_dim, self.state_dim = dofs, nlinks, wksp_dim, state_dim
    self.sphere_radii = sphere_radii
    self.batch_size, self.num_traj_states, self.use_cuda = batch_size, num_traj_states, use_cuda

  def get_sphere_radii(self):
    return self.sphere_radii
