"""On-device policy inference / rollout buffer (bullet-envs_amd/rollout.py).

CPU tests: the restated ActorCritic, compute_gae and ppo_update against vectors produced by
RUNNING the reference's ppo/model.py and ppo/agent.py (tests/golden/make_policy_vectors.py ->
tests/golden/policy_vectors.npz); rollout bookkeeping on a scripted env; gradient averaging
over 2 gloo ranks.  GPU test: a device-resident rollout replayed through the host API."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "policy_vectors.npz")


@pytest.fixture(scope="module")
def ro():
    return importlib.import_module("bullet-envs_amd").rollout


@pytest.fixture(scope="module")
def tm():
    """The trainer math (GAE, PPO update) lives outside the product package: tools/ppo_trainer_math.py."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    return importlib.import_module("ppo_trainer_math")


def _net_from(ro, g, prefix):
    net = ro.ActorCritic(56, 8, [16, 16])
    sd = {k[len(prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}
    assert set(sd) == set(net.state_dict())          # parameter names match the reference's
    net.load_state_dict(sd)
    return net


def test_actor_critic_matches_reference_vectors(ro):
    g = np.load(GOLD)
    net = _net_from(ro, g, "w0/")
    x = torch.tensor(g["fwd/x"])
    dist, value = net(x)
    assert np.allclose(dist.loc.detach().numpy(), g["fwd/mu"], atol=1e-6)
    assert np.allclose(dist.scale.detach().numpy(), g["fwd/sigma"], atol=1e-6)
    assert np.allclose(value.detach().numpy(), g["fwd/value"], atol=1e-6)
    a = torch.tensor(g["fwd/action"])
    assert np.allclose(dist.log_prob(a).detach().numpy(), g["fwd/log_prob"], atol=1e-5)
    assert np.allclose(dist.entropy().detach().numpy(), g["fwd/entropy"], atol=1e-6)
    mu, sigma, v = net.heads(x)
    assert torch.equal(mu, dist.loc) and torch.equal(sigma, dist.scale) and torch.equal(v, value)


def test_init_follows_reference_rule(ro):
    torch.manual_seed(0)
    net = ro.ActorCritic(56, 8, [256, 256])
    for name, p in net.named_parameters():
        if name.endswith("bias"):
            assert torch.all(p == 0.1)
        else:
            assert abs(float(p.detach().std()) - 0.1) < 0.02 and abs(float(p.detach().mean())) < 0.01
    assert sum(p.numel() for p in net.parameters()) == 2 * (56 * 256 + 256 + 256 * 256 + 256) + 257 + 2 * (256 * 8 + 8)


def test_compute_gae_matches_reference_vectors(tm):
    g = np.load(GOLD)
    t = lambda k: torch.tensor(g[k])
    ret = tm.compute_gae(t("gae/next_value"), t("gae/rewards"), t("gae/masks"), t("gae/values"))
    assert ret.shape == (6, 3, 1)
    assert np.allclose(ret.numpy(), g["gae/returns"], atol=1e-6)


def test_ppo_update_matches_reference_vectors(ro, tm):
    """Same minibatches (np.random.seed(5), randint with replacement), same Adam: the weights
    after 2 epochs x 4 minibatches and the logged means agree with the reference's."""
    g = np.load(GOLD)
    net = _net_from(ro, g, "w0/")
    opt = torch.optim.Adam(net.parameters(), lr=3e-4)
    t = lambda k: torch.tensor(g["ppo/" + k])
    np.random.seed(5)
    out = tm.ppo_update(net, opt, 2, 4, t("states"), t("actions"), t("log_probs"), t("returns"), t("advantages"))
    for k, v in net.state_dict().items():
        assert np.allclose(v.numpy(), g["w1/" + k], atol=2e-6), k
    ref = g["ppo/scalars"]
    assert np.allclose([out["loss"], out["critic_loss"], out["actor_loss"], out["entropy"]], ref, rtol=1e-4, atol=1e-6)


class ScriptedEnv(object):
    """DeviceVecEnv look-alike: clips the actions IN PLACE, reuses one obs buffer, done on a schedule."""

    def __init__(self, n):
        self.num_envs, self.obs_dim, self.act_dim = n, 56, 8
        self.obs = torch.zeros(n, 56)
        self.t = 0
        self.seen = []

    def step(self, actions):
        actions.clamp_(-1.0, 1.0)
        self.seen.append(actions.clone())
        self.t += 1
        self.obs[:] = self.t + actions.sum(dim=1, keepdim=True)      # buffer is overwritten every step
        rew = actions[:, 0] * 2.0
        done = ((torch.arange(self.num_envs) + self.t) % 3 == 0).to(torch.uint8)
        return self.obs, rew, done


def test_collect_bookkeeping(ro, tm):
    torch.manual_seed(3)
    n, T = 5, 4
    env = ScriptedEnv(n)
    net = ro.ActorCritic(56, 8, [16, 16])
    buf = ro.RolloutBuffer(T, n, 56, 8, torch.device("cpu"))
    s0 = torch.randn(n, 56)
    gen = torch.Generator().manual_seed(9)
    s_end = ro.collect(env, net, s0.clone(), buf, generator=gen)
    assert torch.equal(buf.states[0], s0)
    for i in range(T):
        # the stored action is the unclipped sample, the env saw its clipped copy
        assert torch.equal(env.seen[i], buf.actions[i].clamp(-1, 1))
        mu, sigma, v = net.heads(buf.states[i])
        assert torch.allclose(buf.values[i], v) and torch.allclose(
            buf.log_probs[i], torch.distributions.Normal(mu, sigma).log_prob(buf.actions[i]))
        assert torch.equal(buf.rewards[i][:, 0], env.seen[i][:, 0] * 2.0)
        assert torch.equal(buf.masks[i][:, 0], 1.0 - ((torch.arange(n) + i + 1) % 3 == 0).float())
        if i + 1 < T:      # next state is a COPY of the env's buffer at that time
            assert torch.equal(buf.states[i + 1], (i + 1) + env.seen[i].sum(dim=1, keepdim=True).expand(n, 56))
    assert torch.equal(s_end, T + env.seen[-1].sum(dim=1, keepdim=True).expand(n, 56))
    assert (buf.actions.abs() > 1).any()       # otherwise the clipping path was not exercised
    assert torch.allclose(buf.total_reward, buf.rewards.sum())
    st, ac, lp, ret, adv = buf.flat(tm.compute_gae(net(s_end)[1].detach(), buf.rewards, buf.masks, buf.values))
    assert st.shape == (T * n, 56) and ac.shape == (T * n, 8) and lp.shape == (T * n, 8) and adv.shape == (T * n, 1)
    assert torch.equal(st[n:2 * n], buf.states[1])           # step-major, like torch.cat over the lists


def _grad_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ro = importlib.import_module("bullet-envs_amd").rollout
    torch.manual_seed(1)
    net = ro.ActorCritic(56, 8, [16, 16])
    torch.manual_seed(100)
    x = torch.randn(2 * world, 56)[2 * rank:2 * rank + 2]       # this rank's shard of one global batch
    _, v = net(x)
    v.pow(2).mean().backward()
    ro.allreduce_gradients(net)
    if rank == 0:
        np.savez(out_path, **{k: p.grad.numpy() for k, p in net.named_parameters() if p.grad is not None})
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_average_world2_gloo(ro, tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "g.npz")
    mp.spawn(_grad_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    torch.manual_seed(1)
    net = ro.ActorCritic(56, 8, [16, 16])
    torch.manual_seed(100)
    x = torch.randn(4, 56)
    _, v = net(x)
    v.pow(2).mean().backward()        # mean over the global batch = average of the shard means
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert np.allclose(got[k], p.grad.numpy(), atol=1e-6), k


@pytest.mark.gpu
def test_device_rollout_matches_host_replay():
    """64 envs x 5 steps collected with obs/actions never leaving the GPU; the stored actions
    replayed through the host-buffer API give the same rewards / dones / next observations."""
    pkg = importlib.import_module("bullet-envs_amd")
    ro = pkg.rollout
    n, T = 64, 5
    env = pkg.DeviceVecEnv(n)
    dev = env.device
    torch.manual_seed(5)
    net = ro.ActorCritic(56, 8, [256, 256]).to(dev)
    buf = ro.RolloutBuffer(T, n, 56, 8, dev)
    gen = torch.Generator(device=dev).manual_seed(17)
    state = env.reset().clone()
    s_end = ro.collect(env, net, state, buf, generator=gen)
    torch.cuda.synchronize()
    host = pkg.SnakeVecEnv(n)
    obs = host.reset()
    assert np.array_equal(obs, buf.states[0].cpu().numpy())
    for i in range(T):
        a = buf.actions[i].cpu().numpy().copy()
        obs, rew, done, _ = host.step(a)
        assert np.array_equal(rew.astype(np.float32), buf.rewards[i, :, 0].cpu().numpy())
        assert np.array_equal(1.0 - done.astype(np.float32), buf.masks[i, :, 0].cpu().numpy())
        nxt = buf.states[i + 1].cpu().numpy() if i + 1 < T else s_end.cpu().numpy()
        assert np.array_equal(obs.astype(np.float32), nxt)
    host.close(); env.close()


@pytest.mark.gpu
def test_actor_critic_on_device_matches_reference_vectors(ro):
    """The golden vectors generated from the reference's ppo/model.py, with the module and its inputs on cuda:0
    (the GEMMs run through rocBLAS there: 1e-5 instead of the CPU's 1e-6)."""
    dev = torch.device("cuda", 0)
    g = np.load(GOLD)
    net = _net_from(ro, g, "w0/").to(dev)
    x = torch.tensor(g["fwd/x"], device=dev)
    dist, value = net(x)
    assert dist.loc.is_cuda and value.is_cuda
    assert np.allclose(dist.loc.detach().cpu().numpy(), g["fwd/mu"], atol=1e-5)
    assert np.allclose(dist.scale.detach().cpu().numpy(), g["fwd/sigma"], atol=1e-5)
    assert np.allclose(value.detach().cpu().numpy(), g["fwd/value"], atol=1e-5)
    a = torch.tensor(g["fwd/action"], device=dev)
    assert np.allclose(dist.log_prob(a).detach().cpu().numpy(), g["fwd/log_prob"], atol=1e-4)
    assert np.allclose(dist.entropy().detach().cpu().numpy(), g["fwd/entropy"], atol=1e-5)
