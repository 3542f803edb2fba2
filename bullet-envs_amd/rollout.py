"""On-device policy inference and rollout buffer.

With the physics at ~20 ms per 4096-env step, the trainer loop of the reference
(`ppo/train.py:110-140`) would spend its time moving observations and actions through host
memory: `torch.FloatTensor(state).to(device)` -> policy -> `action.cpu().numpy()` -> Pipes.
Here the same loop runs with every tensor resident on the GPU that owns the environments:

    net   = ActorCritic(56, 8, [256, 256]).to(env.device)      # ppo/model.py:17-46
    buf   = RolloutBuffer(num_steps, env.num_envs, 56, 8, env.device)
    state = env.reset().clone()
    state = collect(env, net, state, buf)                      # ppo/train.py:110-140
    ... the trainer's own GAE / PPO update on buf.flat(returns) (ppo/agent.py; host-side caller code, out of
    scope here -- tools/ppo_trainer_math.py keeps a restatement for the tests and the demo loop) ...

`env` is a DeviceVecEnv (one GPU) or, with one process per GPU, each rank's own DeviceVecEnv:
every rank runs the policy on its shard and only gradients cross GPUs
(`allreduce_gradients`, RCCL).  The networks are 2x256 MLPs on [N,56] inputs -- plain library
GEMMs through torch; nothing here is a hot kernel next to the 20 ms physics step.

Names, arguments, defaults and quirks follow the reference: the action handed to the env is a
copy (the env clips it in place, `SnakeGymEnv.py:82-88`), while the UNCLIPPED sample is what
gets stored and what log_prob is evaluated on (`ppo/train.py:118-131`); minibatches are drawn
WITH replacement by `np.random.randint` (`ppo/agent.py:27`), so seeding numpy reproduces the
reference's batches.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal


def _init_linear(m):
    """ppo/model.py:11-14: weights N(0, 0.1), biases 0.1."""
    if isinstance(m, nn.Linear):
        nn.init.normal_(m.weight, mean=0.0, std=0.1)
        nn.init.constant_(m.bias, 0.1)


def _mlp(sizes, last_activation):
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2 or last_activation:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    """Actor-critic of ppo/model.py:17-46; parameter names match, so reference checkpoints
    (`weights.pth['model']`, ppo/train.py:155-167) load with `load_state_dict`.

    critic: in -> h0 -> h1 -> 1 (ReLU between);  actor trunk: in -> h0 -> h1 (ReLU after both);
    mu = tanh(Linear(h1, out));  sigma = sigmoid(Linear(h1, out)) + 0.001.
    """

    def __init__(self, num_inputs, num_outputs, hidden_size):
        super().__init__()
        h0, h1 = hidden_size[0], hidden_size[1]
        self.critic = _mlp([num_inputs, h0, h1, 1], last_activation=False)
        self.actor = _mlp([num_inputs, h0, h1], last_activation=True)
        self.mu = nn.Linear(h1, num_outputs)
        self.sigma = nn.Sequential(nn.Linear(h1, num_outputs), nn.Sigmoid())
        self.apply(_init_linear)

    def heads(self, x):
        """(mu, sigma, value) as plain tensors."""
        value = self.critic(x)
        h = self.actor(x)
        return torch.tanh(self.mu(h)), self.sigma(h) + 0.001, value

    def forward(self, x):
        mu, sigma, value = self.heads(x)
        return Normal(mu, sigma), value


class RolloutBuffer(object):
    """The six Python lists of ppo/train.py:101-107 as preallocated device tensors [T, N, .]."""

    def __init__(self, num_steps, num_envs, obs_dim, act_dim, device):
        f = dict(dtype=torch.float32, device=device)
        self.num_steps, self.num_envs = num_steps, num_envs
        self.states = torch.zeros((num_steps, num_envs, obs_dim), **f)
        self.actions = torch.zeros((num_steps, num_envs, act_dim), **f)
        self.log_probs = torch.zeros((num_steps, num_envs, act_dim), **f)
        self.values = torch.zeros((num_steps, num_envs, 1), **f)
        self.rewards = torch.zeros((num_steps, num_envs, 1), **f)
        self.masks = torch.zeros((num_steps, num_envs, 1), **f)
        self.entropy = torch.zeros((), **f)
        self.total_reward = torch.zeros((), **f)

    def flat(self, returns):
        """(states, actions, log_probs, returns, advantages) concatenated over steps, detached
        where the reference detaches (ppo/train.py:176-181)."""
        T, N = self.num_steps, self.num_envs
        ret = returns.reshape(T * N, 1).detach()
        val = self.values.reshape(T * N, 1).detach()
        return (self.states.reshape(T * N, -1), self.actions.reshape(T * N, -1),
                self.log_probs.reshape(T * N, -1).detach(), ret, ret - val)


def collect(envs, net, state, buf, generator=None):
    """num_steps policy/env steps with everything on the device (ppo/train.py:110-140).

    envs.step(actions) -> (obs, reward, done) device tensors (DeviceVecEnv); `state` is the
    current observation [N, O] (a tensor the env does not overwrite).  Returns the next state.
    Values and log-probs are stored detached (the reference detaches them before the update).
    """
    buf.entropy.zero_()
    buf.total_reward.zero_()
    with torch.no_grad():
        for i in range(buf.num_steps):
            mu, sigma, value = net.heads(state)
            action = torch.normal(mu, sigma, generator=generator)          # dist.sample()
            obs, reward, done = envs.step(action.clone())[:3]              # the env clips its copy
            dist = Normal(mu, sigma)
            buf.log_probs[i] = dist.log_prob(action)
            buf.entropy += dist.entropy().mean()
            buf.values[i] = value
            buf.rewards[i] = reward.reshape(-1, 1)
            buf.masks[i] = 1.0 - done.reshape(-1, 1).to(torch.float32)
            buf.states[i] = state
            buf.actions[i] = action
            buf.total_reward += reward.sum()
            state = obs.clone()        # the env reuses its observation buffer
    return state


def allreduce_gradients(net, group=None):
    """Average the gradients over the ranks (one process per GPU, RCCL): the only exchange of the
    training loop when every rank steps its own shard with its own copy of the policy."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])      # one bucket: 150 k parameters
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= world
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g))
        o += n
