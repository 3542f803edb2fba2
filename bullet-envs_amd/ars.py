"""Device-side pieces of the ARS caller: the running observation
normaliser and the linear policy of `ars/train.py`, batched over the environments.

Reference (`ars/train.py:85-99,152-173`): every env-step, for each direction d in order,
    normalizer.observe(state_d); state_d = normalizer.normalize(state_d); action_d = W_d @ state_d
i.e. ONE Welford accumulator updated sequentially, so direction d is normalised with statistics
that already contain directions 0..d of this step.  `observe_normalize` reproduces exactly that
order dependence for a whole [N, O] batch with cumulative sums (no Python loop over N), on
whatever device the tensors live on.  `store`/`restore` use the reference's mean.txt / var.txt.
"""
import os

import numpy as np
import torch


class Normalizer(object):
    def __init__(self, nb_inputs, device="cpu", dtype=torch.float64):
        """nb_inputs: the feature count, or the reference's [1, features] (ars/train.py:196)."""
        feat = int(np.prod(nb_inputs))
        z = lambda: torch.zeros(feat, dtype=dtype, device=device)
        self.n, self.mean, self.mean_diff, self.var = z(), z(), z(), z()

    def observe(self, x):
        """One observation (ars/train.py:159-164)."""
        x = torch.as_tensor(x, dtype=self.mean.dtype, device=self.mean.device).reshape(-1)
        self.n += 1.0
        last_mean = self.mean.clone()
        self.mean += (x - self.mean) / self.n
        self.mean_diff += (x - last_mean) * (x - self.mean)
        self.var = (self.mean_diff / self.n).clamp(min=1e-2)

    def normalize(self, inputs):
        return (torch.as_tensor(inputs, dtype=self.mean.dtype, device=self.mean.device) - self.mean) / torch.sqrt(self.var)

    def observe_normalize(self, X, observe=True):
        """X [N, O]: row d observed then normalised, rows in order -- the reference's inner loop
        (ars/train.py:88-94) without the loop.  Returns the normalised rows [N, O]."""
        X = torch.as_tensor(X, dtype=self.mean.dtype, device=self.mean.device)
        if not observe:
            return (X - self.mean) / torch.sqrt(self.var)
        N = X.shape[0]
        k = torch.arange(1, N + 1, dtype=X.dtype, device=X.device).unsqueeze(1)
        n_k = self.n.unsqueeze(0) + k                                         # count after row k
        csum = torch.cumsum(X, dim=0)
        mean_k = (self.mean * self.n).unsqueeze(0) / n_k + csum / n_k          # running mean after row k
        mean_prev = torch.cat([self.mean.unsqueeze(0), mean_k[:-1]], dim=0)    # ... before row k
        m2_k = self.mean_diff.unsqueeze(0) + torch.cumsum((X - mean_prev) * (X - mean_k), dim=0)
        var_k = (m2_k / n_k).clamp(min=1e-2)
        out = (X - mean_k) / torch.sqrt(var_k)
        self.n, self.mean, self.mean_diff, self.var = n_k[-1].clone(), mean_k[-1].clone(), m2_k[-1].clone(), var_k[-1].clone()
        return out

    def store(self, path):
        np.savetxt(os.path.join(path, "mean.txt"), self.mean.cpu().numpy().reshape(1, -1))
        np.savetxt(os.path.join(path, "var.txt"), self.var.cpu().numpy().reshape(1, -1))

    def restore(self, path):
        """ars/test.py:107-109."""
        dev, dt = self.mean.device, self.mean.dtype
        self.mean = torch.as_tensor(np.loadtxt(os.path.join(path, "mean.txt")).reshape(-1), dtype=dt, device=dev)
        self.var = torch.as_tensor(np.loadtxt(os.path.join(path, "var.txt")).reshape(-1), dtype=dt, device=dev)


def policy(states, weights):
    """Linear policies of all directions at once: weights [N, A, O], states [N, O] ->
    actions [N, A, 1], the stack of `np.matmul(weights_d, state_d.reshape(-1, 1))`
    (ars/train.py:38-39,95-98) that the vector env accepts as (N, A, 1)."""
    return torch.bmm(weights, states.unsqueeze(2))
