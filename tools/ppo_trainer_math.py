"""Trainer math of the reference (ppo/agent.py:14-57), restated for the tests and for tools/train_ppo_device.py.

NOT part of the product package: SURVEY.md section 2 marks the PPO/ARS trainers out of scope (they stay host-side
callers of the env); SURVEY 8(f)-1 asks for on-device policy inference, the rollout buffer and the gradient
all-reduce only (bullet-envs_amd/rollout.py).  Kept here so that tests/golden/policy_vectors.npz (generated from the
reference's own agent.py) still pins the whole loop around the hot path.
"""
import numpy as np
import torch


def compute_gae(next_value, rewards, masks, values, gamma=0.99, tau=0.95):
    """Generalised advantage estimation (ppo/agent.py:14-22) on stacked [T, N, 1] tensors;
    returns `returns` [T, N, 1] (= advantage + value)."""
    T = rewards.shape[0]
    returns = torch.empty_like(rewards)
    gae = torch.zeros_like(next_value)
    nxt = next_value
    for step in reversed(range(T)):
        delta = rewards[step] + gamma * nxt * masks[step] - values[step]
        gae = delta + gamma * tau * masks[step] * gae
        returns[step] = gae + values[step]
        nxt = values[step]
    return returns


def ppo_update(net, optimizer, ppo_epochs, mini_batch_size, states, actions, log_probs, returns, advantages,
               clip_param=0.2, grad_sync=None):
    """Clipped-surrogate PPO update (ppo/agent.py:24-57).  Minibatch indices come from
    np.random.randint(0, batch, mini) -- with replacement, like the reference -- and
    batch // mini minibatches are drawn per epoch.  grad_sync: callable(net) run between
    backward() and step() (e.g. allreduce_gradients).  Returns the mean losses the reference logs."""
    batch_size = states.size(0)
    n_mb = batch_size // mini_batch_size
    tot = dict(loss=0.0, actor_loss=0.0, critic_loss=0.0, entropy=0.0)
    for _ in range(ppo_epochs):
        for _ in range(n_mb):
            ids = torch.as_tensor(np.random.randint(0, batch_size, mini_batch_size), device=states.device)
            state, action = states[ids, :], actions[ids, :]
            old_log_probs, return_, advantage = log_probs[ids, :], returns[ids, :], advantages[ids, :]
            dist, value = net(state)
            entropy = dist.entropy().mean()
            ratio = (dist.log_prob(action) - old_log_probs).exp()
            surr1 = ratio * advantage
            surr2 = torch.clamp(ratio, 1.0 - clip_param, 1.0 + clip_param) * advantage
            actor_loss = -torch.min(surr1, surr2).mean()
            critic_loss = (return_ - value).pow(2).mean()
            loss = 0.5 * critic_loss + actor_loss - 0.001 * entropy
            optimizer.zero_grad()
            loss.backward()
            if grad_sync is not None:
                grad_sync(net)
            optimizer.step()
            tot["loss"] += loss.item(); tot["actor_loss"] += actor_loss.item()
            tot["critic_loss"] += critic_loss.item(); tot["entropy"] += entropy.item()
    denom = ppo_epochs * (batch_size / float(mini_batch_size))
    return {k: v / denom for k, v in tot.items()}
