"""Multi-GPU execution of the planner: trajectories are independent (no coupling across the batch anywhere in
PlanLayer.forward, plan_layer.py:152-234), so the batch is cut into contiguous per-rank slices, every rank runs its
slice on its own GPU with ZERO communication during the Gauss-Newton iterations, and one all-gather (RCCL over xGMI
when the backend is 'nccl') collects the final trajectories.  One process per GPU (torch.distributed).

Nothing here touches the solver's arithmetic: `solve_fn` is whatever maps a local shard of inputs to a local shard of
outputs -- DiffGPMP2Planner.forward on a GPU rank, or any stand-in in the gloo/CPU tests of the sharding logic.
"""
import torch
import torch.distributed as dist


def shard_range(batch, rank, world_size):
  """Contiguous slice [lo, hi) of a batch of `batch` trajectories owned by `rank`: sizes differ by at most one, the
  first `batch % world_size` ranks get the extra trajectory."""
  base, extra = divmod(int(batch), int(world_size))
  lo = rank * base + min(rank, extra)
  return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank=None, world_size=None):
  """Slice every (B, ...) tensor of `tensors` to this rank's trajectories.  Tensors with a leading dimension of 1 or an
  expand()ed batch dimension (a shared SDF) are passed through untouched."""
  rank = dist.get_rank() if rank is None else rank
  world_size = dist.get_world_size() if world_size is None else world_size
  batch = max(t.shape[0] for t in tensors if t is not None)
  lo, hi = shard_range(batch, rank, world_size)
  out = []
  for t in tensors:
    if t is None or t.shape[0] == 1 or t.stride(0) == 0:
      out.append(t if (t is None or t.shape[0] == 1) else t[:hi - lo])
    else:
      out.append(t[lo:hi])
  return out


def gather_buffer(local, batch, group=None):
  """The (world * largest shard, ...) buffer all_gather_trajectories fills, for callers that gather repeatedly (a GN loop that
  collects trajectories every outer iteration, bench.py's timed regions) and pass it back as `out=`: no allocation per call."""
  world = dist.get_world_size(group)
  mx = max(shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world))
  return local.new_empty((world * mx,) + tuple(local.shape[1:]))


def all_gather_trajectories(local, batch, group=None, out=None):
  """All-gather per-rank result slices (B_r, ...) into the full (batch, ...) tensor on every rank.  Ragged shards
  (batch not divisible by the world size) are padded to the largest shard for the collective and trimmed afterwards.
  `out`: a buffer from gather_buffer() to gather into (the returned tensor is `out` itself when the shards are even)."""
  world = dist.get_world_size(group)
  sizes = [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]
  mx = max(sizes)
  pad = local
  if local.shape[0] < mx:
    pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
  pad = pad.contiguous()
  shape = (world * mx,) + tuple(local.shape[1:])
  if out is None:
    out = local.new_empty(shape)
  elif tuple(out.shape) != shape or out.dtype != local.dtype or out.device != local.device or not out.is_contiguous():
    raise ValueError('out must be a contiguous %s %s tensor on %s (gather_buffer() makes one)' % (shape, local.dtype, local.device))
  dist.all_gather_into_tensor(out, pad, group=group)
  if all(s == mx for s in sizes):
    return out
  return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], 0)


def plan_sharded(solve_fn, th_init, start, goal, sdf, group=None):
  """Run `solve_fn(th, start, goal, sdf) -> th_final` on this rank's slice of the GLOBAL inputs and return the gathered
  (B, n, d) final trajectories on every rank."""
  rank, world = dist.get_rank(group), dist.get_world_size(group)
  B = th_init.shape[0]
  th_l, st_l, go_l, sdf_l = shard_batch([th_init, start, goal, sdf], rank, world)
  th_final_l = solve_fn(th_l, st_l, go_l, sdf_l)
  return all_gather_trajectories(th_final_l, B, group)
