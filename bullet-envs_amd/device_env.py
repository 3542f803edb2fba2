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
from .snake_env import params_from_args


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

    def step(self, actions, vec_mode=True):
        """actions: float32 CUDA tensor [E, A], contiguous; clipped in place.  Asynchronous."""
        t = self.torch
        assert actions.is_cuda and actions.dtype == t.float32 and actions.is_contiguous()
        assert tuple(actions.shape) == (self.num_envs, self.act_dim)
        self.stepper.step_device(actions.data_ptr(), self.obs.data_ptr(), self.rew.data_ptr(),
                                 self.done.data_ptr(), self.substeps.data_ptr(), vec_mode, self._stream())
        return self.obs, self.rew, self.done

    def set_ground_friction(self, mu):
        self.stepper.set_ground_friction(np.asarray(mu, dtype=np.float32))

    def close(self):
        self.stepper.close()


class ShardedVecEnv(object):
    """Envs sharded over the ranks of a process group; SubprocVecEnv API on the root rank.

    local_env: object with num_envs, obs_dim, act_dim, reset() -> obs tensor [E,O] and
               step(actions tensor [E,A]) -> (obs [E,O], rew [E], done [E]) on `device`.
    Every rank must call reset()/step() collectively; non-root ranks pass actions=None and get
    (None, None, None, ()) back.
    """

    def __init__(self, local_env, root=0, group=None, device=None):
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
            self._gather = list(self._all.split(self.E, dim=0))
            self._all_dev = self._all if self._xdev == self.device else torch.zeros_like(self._all, device=self.device)
            self._act_all = torch.zeros((self.num_envs, self.A), dtype=torch.float32, device=self._xdev)
        else:
            self._gather = None
        self._infos = None

    def __len__(self):
        return self.num_envs

    def shard_slice(self, rank=None):
        r = self.rank if rank is None else rank
        return slice(r * self.E, (r + 1) * self.E)

    def _gather_pack(self, obs, rew=None, done=None):
        t = self.torch
        self._pack[:, :self.O] = obs
        if rew is not None:
            self._pack[:, self.O] = rew
            self._pack[:, self.O + 1] = done.to(t.float32)
        else:
            self._pack[:, self.O:] = 0
        self.dist.gather(self._pack.to(self._xdev), self._gather, dst=self.root, group=self.group)
        if self.rank != self.root:
            return None
        if self._all_dev is not self._all:
            self._all_dev.copy_(self._all)
        return self._all_dev

    def reset(self):
        allp = self._gather_pack(self.env.reset())
        return None if allp is None else allp[:, :self.O]

    def step(self, actions=None):
        """The root's `actions` (tensor or ndarray [world * E, A] or [.., A, 1]) are NOT modified: SubprocVecEnv pickles
        them to its workers (ppo/multiprocessing_env.py:119-122), so checkBound (SnakeGymEnv.py:82-88) only ever clips
        the workers' copies -- here the scattered copies, which the step kernels clip."""
        t = self.torch
        if self.rank == self.root:
            a = t.as_tensor(actions, dtype=t.float32)       # shares memory with a float32 ndarray / tensor
            a2 = a[:, :, 0] if (a.dim() == 3 and a.shape[2] == 1) else a
            assert tuple(a2.shape) == (self.num_envs, self.A), a2.shape
            self._act_all.copy_(a2)
            chunks = list(self._act_all.split(self.E, dim=0))
        else:
            a2 = None
            chunks = None
        self.dist.scatter(self._act, chunks, src=self.root, group=self.group)
        obs, rew, done = self.env.step(self._act.to(self.device))
        allp = self._gather_pack(obs, rew, done)
        if allp is None:
            return None, None, None, ()
        if self._infos is None:       # train mode: empty dicts (SnakeGymEnv.py:46-47); made once -- 32 768 of them per step
            self._infos = tuple({} for _ in range(self.num_envs))      # at 8 ranks would cost the root 2 ms of every step
        return allp[:, :self.O], allp[:, self.O], allp[:, self.O + 1] > 0.5, self._infos

    def close(self):
        if hasattr(self.env, "close"):
            self.env.close()
