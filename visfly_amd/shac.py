"""EXPERIMENTAL (parity unpinned, network shapes NOT the reference's -- see "Parity status" below).

SHAC as the reference runs it (utils/algorithms/shac.py:215-278): a short-horizon first-order actor update through
the differentiable simulator plus twin Q critics regressed onto TD-lambda returns.

What the reference's ``learn`` loop does per iteration, and where each piece runs here:

* H env steps with the stochastic actor, ``actor_loss = -sum_t reward_t * disc_t`` minus the bootstrap
  ``next_value * disc * gamma`` where a horizon or an episode is cut (:251-257).  ``next_value`` comes from the TARGET
  critics on detached observations and detached next actions (:247), so it is a constant of the actor objective: the
  actor gradient is exactly the BPTT gradient -> ``BPTT._grad_reverse_sweep`` (adjoint env kernel + one-launch network
  backward); the bootstrap only enters the logged loss value.
* ``compute_td_returns`` (utils/algorithms/common.py:893-923) -> ``vf_td_returns`` (bit-identical to the reference on
  the golden vectors).  The reference's buffer calls it with the literal gamma 0.99 (common.py:1232-1239), mirrored.
* ``gradient_steps`` critic updates on the whole horizon buffer: ``mse(returns, min(Q1, Q2))`` (:267-270), joint
  grad-norm clip 0.5 over both Q networks, Adam, Polyak update of the target critics (:274).

Parity status: UNPINNED -- the reference's SHAC needs stable-baselines3 (SACPolicy / ContinuousCritic), which is not
importable in the build container, so no golden vectors exist for the loop itself.  Network shapes differ from SB3's
where the MFMA kernels' layer model requires it: the actor keeps a state-independent log_std (SB3's Actor has a
log_std head), and each Q network is an ``MlpPolicy`` value trunk over the concatenated (observation, action) row
instead of SB3's features extractor -> concat(features, action) -> Q MLP (utils/policies/td_policies.py:146-252; it would
need an identity extractor branch for the action columns, which the fused MLP layer tables do not have).  What IS pinned:
the actor gradient (= the BPTT reverse sweep, gradient fixtures from the reference's autograd) and ``vf_td_returns``
(bit-identical to the reference's compute_td_returns).  Treat learning curves from this class as indicative only.
"""
import ctypes as C
from typing import Optional

import torch as th

from . import _lib, parallel
from .bptt import BPTT
from .ppo import MlpPolicy, _ptr

EP_EPISODE_DONE = 8      # VF_EP_EPISODE_DONE


class _Critic:
    """one Q network + its Adam state"""

    def __init__(self, in_dim, arch, device, seed):
        self.net = MlpPolicy({"sa": in_dim}, {"sa": list(arch[:1])}, [1], list(arch[1:]) or [arch[-1]], device, seed=seed)
        self.net.lazy_pack = True
        n = self.net.n_params
        self.m, self.v = th.zeros(n, device=device), th.zeros(n, device=device)
        self.sumsq = th.zeros(1, device=device)

    def q(self, sa, save=False, slot=0):
        _, value = self.net.forward({"sa": sa}, save_activations=save, slot=slot)
        return value.view(-1)


class SHAC(BPTT):
    def __init__(self, env, horizon: int = 32, tau: float = 0.005, gamma: float = 0.99, gradient_steps: int = 5,
                 learning_rate: float = 1e-3, critic_arch=(128, 64, 64), lamda: float = 0.95, seed: int = 42, **kw):
        super().__init__(env, horizon=horizon, gamma=gamma, learning_rate=learning_rate, seed=seed, **kw)
        self.tau, self.gradient_steps, self.lamda = tau, gradient_steps, lamda
        dev = self.device
        in_dim = sum(self.policy.obs_dims[k] for k in self.obs_keys) + 4
        self.critics = [_Critic(in_dim, critic_arch, dev, seed + 101 + i) for i in range(2)]
        self.targets = [_Critic(in_dim, critic_arch, dev, seed + 101 + i) for i in range(2)]
        for c, t in zip(self.critics, self.targets):
            t.net.flat.copy_(c.net.flat)
            t.net.mark_updated()
        self._critic_step = 0
        self._buf = None

    # ---- rollout bookkeeping (called from BPTT._grad_reverse_sweep) ------------------------------------------------
    def _sa(self, obs, action):
        return th.cat([obs[k].detach() for k in self.obs_keys] + [action.detach()], dim=1).contiguous()

    def _on_step(self, t, pre_obs, action, obs, reward, done, disc):
        env, N, dev = self.env, self.env.num_envs, self.device
        if t == 0:
            H = self.H
            self._buf = dict(sa=[], reward=th.empty((H, N), device=dev), done=th.empty((H, N), dtype=th.uint8, device=dev),
                             ep_done=th.empty((H, N), dtype=th.uint8, device=dev), next_value=th.empty((H, N), device=dev),
                             boot=th.zeros(N, device=dev))
        b = self._buf
        # next action of the (stochastic) actor on the new observation, Q-target on (obs', a') -- all detached (:242-247)
        mean, _ = self.policy.forward({k: obs[k].detach().contiguous() for k in self.obs_keys}, save_activations=False,
                                      slot=self.H, need_value=False)      # a slot of its own: the horizon's activations stay intact
        eps = th.randn((N, 4), device=dev, generator=self._gen)
        nxt = th.empty((N, 4), device=dev)
        _lib.check(_lib.lib().vf_reparam_fwd(_ptr(mean), _ptr(self.policy.log_std), _ptr(eps), _ptr(nxt), N,
                                             _lib.current_stream(dev)))
        sa_next = self._sa(obs, nxt)
        nv = th.minimum(self.targets[0].q(sa_next).clone(), self.targets[1].q(sa_next))
        ep_done = done & ((env._ep_flags & EP_EPISODE_DONE) != 0)
        b["sa"].append(self._sa(pre_obs, action))
        b["reward"][t].copy_(reward)
        b["done"][t].copy_(done)
        b["ep_done"][t].copy_(ep_done)
        b["next_value"][t].copy_(nv)
        cut = (done | (t == self.H - 1)) & ~ep_done                                   # :254
        b["boot"] += nv * disc * self.gamma * cut                                      # :256 (logged loss only)

    # ---- one iteration ----------------------------------------------------------------------------------------------
    def _update(self):
        loss = self._grad_reverse_sweep()
        b, N, H = self._buf, self.env.num_envs, self.H
        actor_loss = loss - b["boot"].mean() / self.world
        out = self._apply(actor_loss)
        L, st = _lib.lib(), _lib.current_stream(self.device)
        returns = th.empty((H, N), device=self.device)
        # SimpleRolloutBuffer.compute_returns passes the literal 0.99 (common.py:1232-1239)
        _lib.check(L.vf_td_returns(_ptr(b["reward"]), b["done"].data_ptr(), b["ep_done"].data_ptr(), _ptr(b["next_value"]),
                                   _ptr(returns), H, N, 0.99, float(self.lamda), st))
        sa, target = th.cat(b["sa"], dim=0), returns.view(-1)
        self.logs["train/critic_loss"] = float(self._train_critics(sa, target))
        self.logs["train/actor_loss_with_bootstrap"] = float(out)
        return out

    def _train_critics(self, sa, target):
        L, st, M = _lib.lib(), _lib.current_stream(self.device), sa.shape[0]
        gM = M * self.world
        loss = None
        for _ in range(self.gradient_steps):
            q = [c.q(sa, save=True) for c in self.critics]
            values = th.minimum(q[0], q[1])
            diff = values - target
            loss = (diff * diff).mean()                                               # mse_loss(target, values) :269
            coef = diff * (2.0 / gM)
            first = q[0] <= q[1]                                                      # th.min routes the gradient to one net
            total = th.zeros(1, device=self.device)
            for i, c in enumerate(self.critics):
                dv = th.where(first if i == 0 else ~first, coef, th.zeros_like(coef)).contiguous()
                c.net.backward(None, dv, None)
                c.net.grad[c.net.log_std_off:] = 0.0
                parallel.allreduce_sum_(c.net.grad)
                _lib.check(L.vf_sumsq(_ptr(c.net.grad), c.net.n_params, _ptr(c.sumsq), _ptr(self._scratch), st))
                total += c.sumsq
            self._critic_step += 1
            for c, t in zip(self.critics, self.targets):                              # joint clip over both nets (:272)
                pmap, packed = c.net.pack_map()
                cfg = _lib.AdamCfg(self.lr, self.betas[0], self.betas[1], self.adam_eps, 0.0, 0.5, self._critic_step, 0,
                                   _ptr(pmap), _ptr(packed))
                _lib.check(L.vf_adam_step(_ptr(c.net.flat), _ptr(c.net.grad), _ptr(c.m), _ptr(c.v), c.net.n_params,
                                          _ptr(total), C.byref(cfg), st))
                c.net.mark_updated(packed_current=pmap is not None)
                t.net.flat.lerp_(c.net.flat, self.tau)                                # polyak_update :274
                t.net.mark_updated()
        return loss
