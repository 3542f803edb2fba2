"""World-size 2, 4 and 8 tests of the sharded vector env on CPU (gloo), trainer rank 0 and not 0.  The per-rank
stepper is a test double backed by the CPU oracle, so this exercises exactly the product's sharding / scatter / gather
plumbing (bullet-envs_amd/device_env.py:ShardedVecEnv: slice arithmetic, root != 0, one packed gather), which on the
GPU node runs over RCCL."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E, STEPS = 2, 3


class PlainOracleLocalEnv:
    """DeviceVecEnv look-alike on the CPU: E oracle envs, torch CPU tensors; the PLAIN local-env contract
    (reset / step only: ShardedVecEnv copies the results into its block itself)."""

    def __init__(self, ids):
        import torch
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        self.torch = torch
        self.envs = [orc.OracleEnv() for _ in ids]
        self.num_envs, self.obs_dim, self.act_dim = len(ids), 56, 8
        self.device = torch.device("cpu")

    def reset(self):
        return self.torch.tensor(np.stack([e.reset() for e in self.envs]), dtype=self.torch.float32)

    def step(self, actions):
        o, r, d = [], [], []
        for e, a in zip(self.envs, actions.numpy()):
            oo, rr, dd, _, _ = e.env_step(a.astype(np.float64), vec_mode=True)
            o.append(oo); r.append(rr); d.append(dd)
        t = self.torch
        return (t.tensor(np.stack(o), dtype=t.float32), t.tensor(r, dtype=t.float32),
                t.tensor(d, dtype=t.uint8))




class OracleLocalEnv(PlainOracleLocalEnv):
    def step_packed(self, actions, packed):
        """The form ShardedVecEnv calls: rows [obs | reward | done (integer bits)] written by the local env itself."""
        o, r, d = self.step(actions)
        O = self.obs_dim
        packed[:, :O] = o
        packed[:, O] = r
        packed.view(self.torch.int32)[:, O + 1] = d.to(self.torch.int32)
        return packed


def _actions(j, n):
    k = np.arange(8)
    a = (-np.sin((2 * k[None, :] + 1) * 4.0 + 0.2 * j + 0.7 * np.arange(n)[:, None])).astype(np.float32)
    return a * np.float32(1.5) if j == 1 else a       # step 1 leaves the [-1, 1] box: checkBound must clip it


def _worker(rank, world, port, out_path, root, plain=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("bullet-envs_amd")
    local = (PlainOracleLocalEnv if plain else OracleLocalEnv)(list(range(rank * E, (rank + 1) * E)))
    env = pkg.ShardedVecEnv(local, root=root, fresh_infos=plain)
    assert env.num_envs == world * E and len(env) == world * E
    assert env.shard_slice() == slice(rank * E, (rank + 1) * E)
    res = {}
    obs0 = env.reset()
    if rank == root:
        res["reset"] = obs0.numpy().copy()
    else:
        assert obs0 is None
    for j in range(STEPS):
        acts = _actions(j, world * E) if rank == root else None
        out = env.step(acts)
        if rank == root:
            # the caller's array is NOT touched: SubprocVecEnv pickles the actions to its workers
            # (ppo/multiprocessing_env.py:119-122), so checkBound (SnakeGymEnv.py:82-88) clips the workers' copies
            assert np.array_equal(acts, _actions(j, world * E))
            obs, rew, done, infos = out
            assert len(infos) == world * E and obs.shape == (world * E, 56)
            import pickle
            assert isinstance(infos[0], dict) and infos[0] == {} and pickle.loads(pickle.dumps(infos))[1] == {}
            if plain:        # fresh dicts, writable, as the reference's workers send them
                infos[0]["episode"] = j
                assert infos[1] == {}
            else:            # one shared dict: writing must fail loudly rather than leak into every env and step
                with pytest.raises(TypeError):
                    infos[0]["episode"] = j
            res["obs%d" % j] = obs.numpy().copy()
            res["rew%d" % j] = rew.numpy().copy()
            res["done%d" % j] = done.numpy().copy()
        else:
            assert out[0] is None
    if rank == root:
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,root,plain", [(2, 0, False), (4, 3, False), (8, 5, False), (2, 1, True)])
def test_sharded_env_gloo(tmp_path, oracle_mod, world, root, plain):
    """plain: a local env with reset() / step() only (no step_packed) and fresh_infos=True."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "sharded.npz")
    mp.spawn(_worker, args=(world, port, out_path, root, plain), nprocs=world, join=True)
    got = np.load(out_path)
    # the same world * E envs stepped directly, unsharded: global env g lives on rank g // E whoever the trainer rank is
    ref = [oracle_mod.OracleEnv() for _ in range(world * E)]
    assert np.allclose(got["reset"], np.stack([e.reset() for e in ref]))
    for j in range(STEPS):
        a = _actions(j, world * E)
        for i, e in enumerate(ref):
            o, r, d, _, _ = e.env_step(a[i].astype(np.float64), vec_mode=True)
            assert np.allclose(got["obs%d" % j][i], o.astype(np.float32))
            assert abs(got["rew%d" % j][i] - r) < 1e-6
            assert bool(got["done%d" % j][i]) == d
