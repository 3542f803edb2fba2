"""Device-resident and multi-GPU forms of the vector env.

DeviceVecEnv   torch CUDA(=HIP) tensors in, tensors out; the step kernel is enqueued on
               torch's current stream, nothing touches the host.
ShardedVecEnv  one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).
               Global env g lives on rank g // envs_per_rank for its whole life; the
               physics never communicates.  Per env-step there is exactly one exchange each
               way with the trainer rank: actions scatter (N*A*4 B per rank) and one packed
               [obs | reward | done] gather ((O+2)*4 B per env).  A gather-to-root is
               inbound-link-parallel on the fully connected xGMI mesh, so it is not
               ring-bound.

Replaces the Pipe fan-out/fan-in of SubprocVecEnv (ppo/multiprocessing_env.py:119-128).
torch is plumbing here (device memory, streams, process groups); it computes nothing.
"""
import numpy as np

from . import _lib
from .snake_env import FrozenInfo, params_from_args


class DeviceVecEnv(object):
    def __init__(self, num_envs, device_index=0, args=None, n_modules=16, params=None, **over):
        import torch
        self.torch = torch
        self.params = params if params is not None else params_from_args(args, n_modules=n_modules, **over)
        torch.cuda.set_device(device_index)
        self.device = torch.device("cuda", device_index)
        self.stepper = _lib.Stepper(num_envs, device=device_index, params=self.params)
        self.num_envs = num_envs
        self.obs_dim = self.stepper.obs_dim
        self.act_dim = self.stepper.act_dim
        E = num_envs
        self.obs = torch.zeros((E, self.obs_dim), dtype=torch.float32, device=self.device)
        self.rew = torch.zeros((E,), dtype=torch.float32, device=self.device)
        self.done = torch.zeros((E,), dtype=torch.uint8, device=self.device)
        self.substeps = torch.zeros((E,), dtype=torch.int32, device=self.device)

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def reset(self):
        self.stepper.reset_device(0, self.obs.data_ptr(), self._stream())
        return self.obs

    def _sub_ptr(self, substeps):
        """where Snake.counter of this step goes: self.substeps, or a caller's int32 CUDA tensor [E] (a row of a per-step
        log, say: a rollout can then look at the counts after its last step instead of reducing them after every one)"""
        if substeps is None:
            return self.substeps.data_ptr()
        t = self.torch
        assert substeps.is_cuda and substeps.dtype == t.int32 and substeps.is_contiguous() and substeps.numel() == self.num_envs
        return substeps.data_ptr()

    def step(self, actions, vec_mode=True, substeps=None):
        """actions: float32 CUDA tensor [E, A], contiguous; clipped in place.  Asynchronous."""
        t = self.torch
        assert actions.is_cuda and actions.dtype == t.float32 and actions.is_contiguous()
        assert tuple(actions.shape) == (self.num_envs, self.act_dim)
        self.stepper.step_device(actions.data_ptr(), self.obs.data_ptr(), self.rew.data_ptr(),
                                 self.done.data_ptr(), self._sub_ptr(substeps), vec_mode, self._stream())
        return self.obs, self.rew, self.done

    def step_packed(self, actions, packed, vec_mode=True, substeps=None):
        """The same step, written by the kernel as rows [obs | reward | done] of `packed` ([E, >= O + 2] float32 on this
        device; the done cell holds the integer 0 / 1): the block a sharded env gathers (snk_step_packed).  self.obs /
        rew / done are NOT updated by this call.  Asynchronous."""
        t = self.torch
        assert actions.is_cuda and actions.dtype == t.float32 and actions.is_contiguous()
        assert tuple(actions.shape) == (self.num_envs, self.act_dim)
        assert packed.is_cuda and packed.dtype == t.float32 and packed.is_contiguous()
        assert packed.shape[0] == self.num_envs and packed.shape[1] >= self.obs_dim + 2
        self.stepper.step_packed_device(actions.data_ptr(), packed.data_ptr(), packed.shape[1], self._sub_ptr(substeps),
                                        vec_mode, self._stream())
        return packed

    def set_ground_friction(self, mu):
        self.stepper.set_ground_friction(np.asarray(mu, dtype=np.float32))

    def close(self):
        self.stepper.close()


class ShardedVecEnv(object):
    """Envs sharded over the ranks of a process group; SubprocVecEnv API on the root rank.

    local_env: object with num_envs, obs_dim, act_dim, reset() -> obs tensor [E,O], and either
               step_packed(actions tensor [E,A], packed tensor [E,O+2]) writing rows [obs | reward | done] into
               `packed` itself (DeviceVecEnv: the step kernel does; the done cell holds the INTEGER 0 / 1, i.e. its
               int32 bits inside the float32 block) -- or, without that method, the plain
               step(actions) -> (obs [E,O], rew [E], done [E]) on `device`, whose results are then copied into the block.
    Every rank must call reset()/step() collectively; non-root ranks pass actions=None and get
    (None, None, None, ()) back.
    fresh_infos: True = a fresh dict per env per step, as the reference's workers send (32 768 dicts cost the root of 8
    ranks about 1 ms of every step); False (default) = one read-only, picklable FrozenInfo repeated.
    """

    def __init__(self, local_env, root=0, group=None, device=None, fresh_infos=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.env = local_env
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.root = root
        self.device = device if device is not None else getattr(local_env, "device", torch.device("cpu"))
        self.E = local_env.num_envs
        self.O = local_env.obs_dim
        self.A = local_env.act_dim
        self.num_envs = self.E * self.world
        self.backend = dist.get_backend(group)
        # The collectives run where the backend can: on the device for "nccl" (RCCL over xGMI), through host
        # staging for "gloo" (CPU tests; rehearsing a multi-rank run on a box with one GPU).
        self._xdev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self._act = torch.zeros((self.E, self.A), dtype=torch.float32, device=self._xdev)
        self._pack = torch.zeros((self.E, self.O + 2), dtype=torch.float32, device=self.device)
        if self.rank == root:
            # one preallocated [world * E, O + 2] buffer; the gather writes each rank's block into its slice
            self._all = torch.zeros((self.num_envs, self.O + 2), dtype=torch.float32, device=self._xdev)
            self._gather_list = list(self._all.split(self.E, dim=0))
            self._all_dev = self._all if self._xdev == self.device else torch.zeros_like(self._all, device=self.device)
            self._act_all = torch.zeros((self.num_envs, self.A), dtype=torch.float32, device=self._xdev)
        else:
            self._gather_list = None
        self._infos = None if fresh_infos else (FrozenInfo(),) * self.num_envs
        self._packed_step = hasattr(local_env, "step_packed")

    def __len__(self):
        return self.num_envs

    def shard_slice(self, rank=None):
        r = self.rank if rank is None else rank
        return slice(r * self.E, (r + 1) * self.E)

    def _gather(self):
        """self._pack (this rank's [E, O + 2] block) -> the root's [world * E, O + 2] buffer: the ONE collective of the
        return path (ppo/multiprocessing_env.py:125-128 is a recv per worker)."""
        self.dist.gather(self._pack if self._xdev == self.device else self._pack.to(self._xdev), self._gather_list,
                         dst=self.root, group=self.group)
        if self.rank != self.root:
            return None
        if self._all_dev is not self._all:
            self._all_dev.copy_(self._all)
        return self._all_dev

    def reset(self):
        # (a reset is once per run, not per step: the observation is copied into the block's first O columns)
        self._pack[:, :self.O] = self.env.reset()
        self._pack[:, self.O:] = 0
        allp = self._gather()
        return None if allp is None else allp[:, :self.O]

    def step(self, actions=None):
        """The root's `actions` (tensor or ndarray [world * E, A] or [.., A, 1]) are NOT modified: SubprocVecEnv pickles
        them to its workers (ppo/multiprocessing_env.py:119-122), so checkBound (SnakeGymEnv.py:82-88) only ever clips
        the workers' copies -- here the scattered copies, which the step kernels clip.
        Returns VIEWS of the gathered block (obs [:, :O], reward [:, O], done from the integer cell [:, O + 1]); a
        caller that moves the results on (to the host, into a rollout buffer) should take step_block() and move the
        contiguous block in one copy."""
        t = self.torch
        allp = self.step_block(actions)
        if allp is None:
            return None, None, None, ()
        infos = self._infos if self._infos is not None else tuple({} for _ in range(self.num_envs))
        # the done cell holds the integer 0 / 1 (its bits travel in the float32 block)
        return allp[:, :self.O], allp[:, self.O], allp.view(t.int32)[:, self.O + 1] != 0, infos

    def step_block(self, actions=None, substeps=None):
        """One collective env-step; on the root returns the contiguous [world * E, O + 2] float32 block on `device`
        (rows [obs | reward | done as int32 bits], env g in row g), None elsewhere.  Valid until the next call."""
        t = self.torch
        if self.rank == self.root:
            a = t.as_tensor(actions, dtype=t.float32)       # shares memory with a float32 ndarray / tensor
            a2 = a[:, :, 0] if (a.dim() == 3 and a.shape[2] == 1) else a
            assert tuple(a2.shape) == (self.num_envs, self.A), a2.shape
            self._act_all.copy_(a2, non_blocking=True)      # (asynchronous only from pinned host memory / the device)
            chunks = list(self._act_all.split(self.E, dim=0))
        else:
            a2 = None
            chunks = None
        self.dist.scatter(self._act, chunks, src=self.root, group=self.group)
        # the local env writes its rows [obs | reward | done] into the block itself (DeviceVecEnv: the step kernel does,
        # snk_step_packed): nothing is copied between the physics and the gather
        a_loc = self._act if self._xdev == self.device else self._act.to(self.device)
        if self._packed_step:
            # (substeps: where the local env's substep counts of this step go -- DeviceVecEnv.step_packed's argument)
            if substeps is not None:
                self.env.step_packed(a_loc, self._pack, substeps=substeps)
            else:
                self.env.step_packed(a_loc, self._pack)
        else:
            # a local env with the plain contract: its results copied into the block (three small copies per step)
            obs, rew, done = self.env.step(a_loc)
            self._pack[:, :self.O] = obs
            self._pack[:, self.O] = rew
            self._pack.view(t.int32)[:, self.O + 1] = (t.as_tensor(done) != 0).to(t.int32)
        return self._gather()

    def close(self):
        if hasattr(self.env, "close"):
            self.env.close()
