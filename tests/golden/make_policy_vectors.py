"""Generates tests/golden/policy_vectors.npz by RUNNING the reference's own modules
(/root/reference/ppo/model.py, /root/reference/ppo/agent.py -- they import only torch/numpy):

  * ActorCritic(56, 8, [16, 16]): weights, 5 inputs, and the reference's mu / sigma / value
  * compute_gae on random rewards / masks / values (T=6, N=3)
  * ppo_update: 2 epochs, minibatch 4 on a 18-sample batch with np.random.seed(5): the
    weights after the update and the four logged scalars

Run here (the reference does not exist on the GPU box):  python tests/golden/make_policy_vectors.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/ppo")
import agent   # noqa: E402
import model   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "policy_vectors.npz")
torch.manual_seed(11)
np.random.seed(11)
d = {}

net = model.ActorCritic(56, 8, [16, 16]).to("cpu")
for k, v in net.state_dict().items():
    d["w0/" + k] = v.numpy().copy()
x = torch.randn(5, 56)
dist, value = net(x)
d["fwd/x"] = x.numpy()
d["fwd/mu"] = dist.loc.detach().numpy()
d["fwd/sigma"] = dist.scale.detach().numpy()
d["fwd/value"] = value.detach().numpy()
a = torch.randn(5, 8)
d["fwd/action"] = a.numpy()
d["fwd/log_prob"] = dist.log_prob(a).detach().numpy()
d["fwd/entropy"] = dist.entropy().detach().numpy()

T, N = 6, 3
rewards = [torch.randn(N, 1) for _ in range(T)]
masks = [(torch.rand(N, 1) > 0.3).float() for _ in range(T)]
values = [torch.randn(N, 1) for _ in range(T)]
next_value = torch.randn(N, 1)
returns = agent.compute_gae(next_value, rewards, masks, values)
d["gae/rewards"] = torch.stack(rewards).numpy()
d["gae/masks"] = torch.stack(masks).numpy()
d["gae/values"] = torch.stack(values).numpy()
d["gae/next_value"] = next_value.numpy()
d["gae/returns"] = torch.stack(returns).numpy()


class Writer(object):
    def __init__(self):
        self.s = {}

    def add_scalar(self, name, v, step):
        self.s[name] = float(v)


B = T * N
states = torch.randn(B, 56)
actions = torch.randn(B, 8)
with torch.no_grad():
    dist, vals = net(states)
    old_lp = dist.log_prob(actions)
ret = vals + torch.randn(B, 1) * 0.5
adv = ret - vals
opt = torch.optim.Adam(net.parameters(), lr=3e-4)
w = Writer()
np.random.seed(5)
agent.ppo_update(net, opt, 2, 4, states, actions, old_lp, ret, adv, w, 0)
for k, v in (("states", states), ("actions", actions), ("log_probs", old_lp), ("returns", ret), ("advantages", adv)):
    d["ppo/" + k] = v.numpy()
for k, v in net.state_dict().items():
    d["w1/" + k] = v.numpy().copy()
d["ppo/scalars"] = np.array([w.s["loss/epoch"], w.s["critic_loss/epoch"], w.s["actor_loss/epoch"], w.s["entropy/epoch"]])
np.savez_compressed(OUT, **d)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
