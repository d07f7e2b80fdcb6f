"""First-order (differentiable-simulation) policy optimisation: BPTT over a horizon of env steps.

Restates ``BPTT.learn`` of the reference (utils/algorithms/BPTT.py:77-180): roll the policy through
H env steps with ``requires_grad=True``, loss = mean_i sum_t -reward_t * discount_t with
``discount <- discount * gamma * ~done + done`` (:107-124), back-propagate through the simulator,
clip the gradient norm to 0.5, Adam step, ``env.detach()`` (:127-134).

Where the reference relies on a torch autograd tape of ~3.8k nodes per control step, here the
simulator's reverse pass is the hand-written adjoint kernel ``vf_env_step_bwd`` (one launch per
step) and the policy's is the MFMA linear-layer kernels; ``torch.autograd`` is only the scheduler
that orders those launches (two custom Functions), so any torch policy can be dropped in as well.
"""
import ctypes as C
import os
import time
from typing import Dict, Optional

import torch as th

from . import _lib, checkpoint, parallel
from .ppo import MlpPolicy, _ptr


class EnvStepFunction(th.autograd.Function):
    """env.step with the adjoint kernel as backward.  `token` threads the hidden simulator state through
    the graph so that autograd runs step t's backward after step t+1's (the adjoint slab is in/out)."""

    @staticmethod
    def forward(ctx, action, token, env, is_test):
        obs, reward, done, info = env._step_no_grad(action.detach(), is_test, record=True)
        ctx.env, ctx.t = env, env._tape_t - 1
        env._last_step_aux = (obs, done, info)
        return obs["state"], reward, token + 1

    @staticmethod
    def backward(ctx, d_obs, d_reward, d_token):
        d_action = ctx.env.backward_step(ctx.t, d_obs, d_reward)
        return d_action, d_token, None, None


class PolicyFunction(th.autograd.Function):
    """MlpPolicy mean head: forward / backward on the MFMA linear kernels; parameter gradients are
    accumulated into ``policy.grad`` (flat), the observation gradient is returned to autograd."""

    @staticmethod
    def forward(ctx, policy, keys, anchor, *obs):
        # `anchor` is a dummy leaf that requires grad: the parameters live outside autograd (flat buffer), so without
        # it the node of the FIRST step -- whose observation carries no graph -- would never run its backward and
        # that step's parameter gradient would be lost
        ctx.policy, ctx.keys = policy, keys
        ctx.save_for_backward(*obs)
        mean, _ = policy.forward({k: o.detach().contiguous() for k, o in zip(keys, obs)})
        return mean.clone()

    @staticmethod
    def backward(ctx, d_mean):
        pol = ctx.policy
        obs = ctx.saved_tensors
        pol.forward({k: o.detach().contiguous() for k, o in zip(ctx.keys, obs)})   # activations of THIS step
        d_in = pol.backward(d_mean.contiguous(), None, None, accumulate=True, need_input_grad=True)
        return (None, None, None) + tuple(d_in.get(k) for k in ctx.keys)


# the reference constructs ``BPTT(env, policy, policy_kwargs, ...)`` with one of utils/policies/td_policies.py's policies (BPTT.py:13,
# 29-49); its loop then runs ``self.policy.actor.action_log_prob(obs)`` (:113), i.e. the SAC-style Actor with the state-dependent
# log_std head (td_policies.py:146-252).  These names select that actor here; None keeps the MlpPolicy with a state-independent log_std
REFERENCE_ACTOR_POLICIES = ("MultiInputPolicy", "MTDPolicy", "CnnPolicy", "MlpPolicy")
LOG_STD_MAX, LOG_STD_MIN = 2.0, -10.0        # td_policies.py:31-32


class BPTT:
    def __init__(self, env, horizon: int = 32, gamma: float = 0.99, learning_rate: float = 1e-3,
                 max_grad_norm: float = 0.5, weight_decay: float = 0.0, policy_kwargs: Optional[dict] = None,
                 seed: int = 0, betas=(0.9, 0.999), adam_eps: float = 1e-8, policy=None):
        if policy is not None and not isinstance(policy, str):
            policy = getattr(policy, "__name__", str(policy))
        if policy is not None and policy not in REFERENCE_ACTOR_POLICIES:
            raise NotImplementedError(f"policy {policy}: one of {REFERENCE_ACTOR_POLICIES} (vector observations) or None")
        # True: the reference's own actor (two trunks, mu / clamped state-dependent log_std heads); SHAC always uses it
        self.reference_actor = policy is not None or getattr(self, "reference_actor", False)
        self._eps_override = None            # tests: the exploration noise of the next horizon(s), fed instead of drawn
        self.env, self.H, self.gamma = env, horizon, gamma
        self.lr, self.max_grad_norm, self.weight_decay, self.betas, self.adam_eps = learning_rate, max_grad_norm, weight_decay, betas, adam_eps
        self.device = env.device
        env.set_requires_grad(True, horizon=horizon)        # shac.py:124 sets env.requires_grad = True
        env.tensor_output = True
        if not env._is_initial:
            env.reset()
        obs = env.get_observation()
        self.obs_keys = [k for k in obs.keys() if k in ("state", "target")]
        if hasattr(env, "obs_gate_exact"):       # RacingEnv: the policy does not read "gate"; skip its per-step bookkeeping launches
            env.obs_gate_exact = False
        self.policy = self._make_policy(obs, policy_kwargs, seed)
        self.policy.lazy_pack = True        # this trainer calls mark_updated() after every optimiser step
        self.world, self.rank = parallel.world_size(), parallel.rank()
        # same initial parameters on every rank (only gradients are exchanged afterwards), different exploration noise
        parallel.broadcast_(self.policy.flat)
        self.policy.mark_updated()
        self.use_autograd = False           # True: torch.autograd schedules the same kernels (cross-check path)
        self.fused_rollout = True           # forward half of a horizon as one persistent launch where the library has the kernel
        self.fused_reverse = True           # ... and the reverse half (only after a fused forward: same tape / slot buffers)
        self._defer_wgrad = None            # decided at the first update (MlpPolicy.backward_data_supported)
        n = self.policy.n_params
        self.exp_avg, self.exp_avg_sq = th.zeros(n, device=self.device), th.zeros(n, device=self.device)
        self._sumsq, self._scratch = th.zeros(1, device=self.device), th.zeros(4096, device=self.device)
        self.seed = seed
        self._gen = th.Generator(device=self.device).manual_seed(int(seed) + 1000003 * self.rank)
        self._opt_step = 0
        self.num_timesteps = 0
        self.logs: Dict[str, float] = {}

    def _make_reference_actor(self, obs, policy_kwargs, seed):
        """td_policies.Actor (:146-252): extractor -> (latent_pi -> mu | log_latent_pi -> log_std); nn.Linear default initialisation
        (SB3's SAC policies do not use orthogonal init), log_latent_pi = deepcopy(latent_pi) (:205).  One MlpPolicy layer table with
        two 4-wide heads; sets self._extractor / _ext_keys / _critic_arch (SHAC builds its critics from them)."""
        pk = dict(policy_kwargs or {})
        na = pk.get("net_arch")
        self._critic_arch = getattr(self, "_critic_arch", None)
        if isinstance(na, dict) and "qf" in na:                       # SB3 get_actor_critic_arch: dict(pi=..., qf=...)
            self._critic_arch = list(na["qf"])
            pk["net_arch"] = dict(pi=list(na["pi"]), vf=list(na["pi"]))
        # share_features_extractor=True (td_policies.py:127, SACPolicy._build): the critic runs the ACTOR's extractor under no_grad.  BPTT has
        # no critic (nothing changes); SHAC reads the flag (shac.py: _share_extractor)
        self._share_extractor = bool(pk.pop("share_features_extractor", False))
        if any(k in pk for k in ("features_extractor_class", "net_arch", "features_extractor_kwargs", "activation_fn")):
            pk.setdefault("activation_fn", "relu")                    # MTDPolicy's default activation IS ReLU (td_policies.py:297)
        pk.setdefault("activation_fn", "relu")
        pk = checkpoint.policy_kwargs_from_reference(pk, self.obs_keys)
        self._relu_only(pk)
        self._extractor = pk.get("extractor", {k: [128, 64] for k in self.obs_keys})
        self._ext_keys = list(self._extractor.keys())
        arch = list(pk.get("pi", [64, 64]))
        if self._critic_arch is None:
            self._critic_arch = list(arch)
        pol = MlpPolicy({k: obs[k].shape[1] for k in self.obs_keys}, self._extractor, arch, arch, self.device, seed=seed,
                        ortho_init=False, head_dims=(4, 4), log_std_param=False)
        hidden = lambda trunk: [ly for ly in pol.layers if ly.dst.startswith(trunk + ":")]
        for a, b in zip(hidden("pi"), hidden("vf")):                  # log_latent_pi starts as a copy of latent_pi
            pol.weight(b).copy_(pol.weight(a))
            pol.bias(b).copy_(pol.bias(a))
        return pol

    @staticmethod
    def _relu_only(pk):
        """BPTT / SHAC run their horizons on the persistent chain launches, which are ReLU networks (td_policies' own default,
        td_policies.py:297); Tanh / ELU / LeakyReLU policies: PPO"""
        from .ppo import activation_kind
        if (activation_kind(pk.get("activation", 1)), activation_kind(pk.get("extractor_activation", 1))) != (1, 1):
            raise NotImplementedError("BPTT / SHAC: activation_fn other than relu is not implemented (the horizon kernels are ReLU networks)")

    def _head_fwd(self, mu, log_std, eps, action):
        """action = tanh(mu + eps exp(clamp(log_std))) -- Actor.action_log_prob's sample (SB3 squashed Gaussian)"""
        _lib.check(_lib.lib().vf_shac_head_fwd(_ptr(mu), _ptr(log_std), _ptr(eps), _ptr(action), mu.shape[0], LOG_STD_MIN,
                                               LOG_STD_MAX, _lib.current_stream(self.device)))

    def _make_policy(self, obs, policy_kwargs, seed):
        """the actor network: the reference's Actor when a td_policies policy was named (or by SHAC), else the MlpPolicy with a
        state-independent log_std"""
        if self.reference_actor:
            return self._make_reference_actor(obs, policy_kwargs, seed)
        pk = dict(policy_kwargs or {})
        pk.setdefault("activation_fn", "relu")             # (this actor is not a reference class: its default stays what it was)
        pk = checkpoint.policy_kwargs_from_reference(pk, self.obs_keys)
        self._relu_only(pk)
        self.weight_decay = pk.get("weight_decay", self.weight_decay)
        return MlpPolicy({k: obs[k].shape[1] for k in self.obs_keys},
                         pk.get("extractor", {k: [128, 64] for k in self.obs_keys}), pk.get("pi", [64, 64]),
                         pk.get("vf", [64, 64]), self.device, log_std_init=pk.get("log_std_init", -1.0), seed=seed,
                         ortho_init=pk.get("ortho_init", True))

    def _update(self):
        """one horizon: roll out, back-propagate through simulator and policy, clip + Adam (BPTT.py:100-134)"""
        loss = self._grad_autograd() if self.use_autograd else self._grad_reverse_sweep()
        return self._apply(loss)

    def _grad_reverse_sweep_reference_actor(self):
        """BPTT.py:107-134 with the reference's actor: per step actor.action_log_prob(obs) (both trunks, clamped log_std, squashed
        reparameterised sample; the log-prob it also returns is discarded by the loop) -> env.step -> loss / discount recurrence
        (:123-124); reverse, t = H-1 .. 0: adjoint env step -> action head reverse (d mu, d log_std) -> both trunks + extractor
        -> gradient w.r.t. the observation of step t, which step t-1 returned.  Like the MlpPolicy sweep: the activations of every
        step stay in their slot, the weight gradient is reduced once over the rows of all H steps, and where the library has the
        kernels each half is ONE persistent launch (vf_bptt_rollout / vf_bptt_reverse, actor class (b)) that leaves what the
        launch-by-launch loop below leaves, bit for bit."""
        env, pol, N, H = self.env, self.policy, self.env.num_envs, self.H
        L, st, dev = _lib.lib(), _lib.current_stream(self.device), self.device
        keys = self.obs_keys
        pol.grad.zero_()
        pol.reserve_slots(N, H)
        if self._defer_wgrad is None:
            self._defer_wgrad = pol.backward_data_supported(N, both_heads=True)
        defer = self._defer_wgrad
        f = dict(dtype=th.float32, device=dev)
        disc, loss_vec = th.ones(N, **f), th.zeros(N, **f)
        eps = self._eps_override if self._eps_override is not None else th.randn((H, N, 4), device=dev, generator=self._gen)
        assert eps.shape == (H, N, 4)
        acts, drews = th.empty((H, N, 4), **f), th.empty((H, N), **f)
        ls_rows = pol._slot_blocks[N][1]["value"]                                      # (slots, N, 4): slot t's log_std head
        t0 = env._tape_t
        fused = False
        if defer and self.fused_rollout:
            fused = env.rollout_policy(pol, keys, eps, acts, drews, loss_vec, disc, float(self.gamma), 1.0 / (N * self.world))
        if fused:
            self._last_rollout = dict(action=acts, done=env._tape_done[t0:t0 + H])
        else:
            self._last_rollout = dict(action=acts, reward=th.empty((H, N), **f), done=th.empty((H, N), dtype=th.bool, device=dev))
            obs = env.get_observation()
        for t in range(0 if not fused else H, H):
            o = {k: obs[k].detach().contiguous() for k in keys}
            mu, ls = pol.forward(o, slot=t)                                            # actor.action_log_prob(obs) :113
            self._head_fwd(mu, ls, eps[t], acts[t])                                    # tanh output: the clip of :114-116 is the identity
            obs, reward, done, _ = env._step_no_grad(acts[t], False, record=True, borrow=True)      # :119
            self._last_rollout["reward"][t].copy_(reward)
            self._last_rollout["done"][t].copy_(done)
            _lib.check(L.vf_bptt_accumulate(_ptr(reward), done.data_ptr(), _ptr(disc), _ptr(loss_vec), _ptr(drews[t]),
                                            float(self.gamma), 1.0 / (N * self.world), N, st))     # :123-124
        g_obs = None
        d_mus, d_lss = th.empty((H, N, 4), **f), th.empty((H, N, 4), **f)
        rev = False
        if fused and self.fused_reverse:
            rev = env.reverse_policy(pol, H, eps, acts, drews, d_mus, None, d_log_stds=d_lss)
        for t in reversed(range(0 if not rev else H, H)):
            d_action = env.backward_step(t0 + t, g_obs, drews[t])
            _lib.check(L.vf_shac_head_bwd(_ptr(d_action), _ptr(acts[t]), _ptr(ls_rows[t]), _ptr(eps[t]), _ptr(d_mus[t]), _ptr(d_lss[t]), N,
                                          LOG_STD_MIN, LOG_STD_MAX, st))
            if defer:
                d_in = pol.backward_data(d_mus[t], t, d_value=d_lss[t])
            else:
                d_in = pol.backward(d_mus[t], d_lss[t], None, accumulate=True, need_input_grad=t > 0, slot=t)
            g_obs = d_in.get("state") if t > 0 else None
        if defer:
            pol.weight_grad_slots(N, H, d_mus, accumulate=True, d_value_all=d_lss)
        return loss_vec.mean() / self.world

    def _grad_reverse_sweep(self):
        """explicit reverse sweep: no autograd tape.  Forward: policy (activations of every step stay resident in their
        own slot) -> reparameterised action -> fused env step (state checkpoint on the tape) -> loss / discount
        bookkeeping, one launch each.  Reverse, t = H-1 .. 0: adjoint env step -> action head -> whole-network backward
        (parameter gradients accumulate in MFMA partials + fold) -> gradient w.r.t. the observation of step t, which
        is what step t-1 returned.  dLoss/d reward_t = -disc_t / N is known in the forward pass (:123)."""
        if self.reference_actor:
            return self._grad_reverse_sweep_reference_actor()
        env, pol, N, H = self.env, self.policy, self.env.num_envs, self.H
        L, st, dev = _lib.lib(), _lib.current_stream(self.device), self.device
        pol.grad.zero_()
        # reference-default policy shapes: the weight gradient is reduced ONCE per horizon over the rows of all H
        # steps (their activations / masked gradients are stored back to back), the per-step reverse pass is the
        # register-chained data-gradient kernel only
        pol.reserve_slots(N, H)          # no-op while the slots exist (they are dropped if many other batch sizes pass through)
        if self._defer_wgrad is None:
            self._defer_wgrad = pol.backward_data_supported(N)
        defer = self._defer_wgrad
        disc, loss_vec = th.ones(N, device=dev), th.zeros(N, device=dev)
        g_ls = th.zeros((N, 4), device=dev)
        log_std = pol.log_std
        t0 = env._tape_t
        obs = env.get_observation()
        # per-horizon buffers in one allocation each; the exploration noise of the whole horizon is one draw
        acts, drews = th.empty((H, N, 4), device=dev), th.empty((H, N), device=dev)
        epss = th.randn((H, N, 4), device=dev, generator=self._gen)
        ckpt_done = False
        fused = False
        if defer and self.fused_rollout and pol._act_fused is not False:
            # the whole forward half as one persistent launch (vf_bptt_rollout): same kernels' arithmetic, same buffers
            fused = env.rollout_policy(pol, self.obs_keys, epss, acts, drews, loss_vec, disc, float(self.gamma), 1.0 / (N * self.world))
        for t in range(0 if not fused else H, H):
            action = acts[t]
            o = {k: obs[k].detach().contiguous() for k in self.obs_keys}
            if not (defer and pol.forward_act(o, epss[t], action, slot=t)):      # action head fused into the forward launch
                mean, _ = pol.forward(o, slot=t, need_value=False)
                _lib.check(L.vf_reparam_fwd(_ptr(mean), _ptr(log_std), _ptr(epss[t]), _ptr(action), N, st))
            obs, reward, done, _ = env._step_no_grad(action, False, record=True, borrow=True,   # acts[t] outlives the reverse sweep
                                                     prefilled=ckpt_done)
            # loss / discount bookkeeping, fused with the state checkpoint of step t + 1 (one launch instead of two)
            ckpt_done = t + 1 < H and env._tape_t < env._tape.shape[0]
            if ckpt_done:
                _lib.check(L.vf_bptt_accumulate_checkpoint(_ptr(reward), done.data_ptr(), _ptr(disc), _ptr(loss_vec), _ptr(drews[t]),
                                                           float(self.gamma), 1.0 / (N * self.world), N, _ptr(env._slab),
                                                           _ptr(env._tape[env._tape_t]), env._slab.numel(), st))
            else:
                _lib.check(L.vf_bptt_accumulate(_ptr(reward), done.data_ptr(), _ptr(disc), _ptr(loss_vec), _ptr(drews[t]),
                                                float(self.gamma), 1.0 / (N * self.world), N, st))
        g_obs = None
        d_means = th.empty((H, N, 4), device=dev)
        g_ls_rows = th.zeros((H, N, 4), device=dev) if defer else None     # log_std gradient terms, one row per (step, agent)
        rev = False
        if fused and self.fused_reverse:
            # the whole reverse half as one persistent launch (vf_bptt_reverse)
            rev = env.reverse_policy(pol, H, epss, acts, drews, d_means, g_ls_rows)
        for t in reversed(range(0 if not rev else H, H)):
            d_action = env.backward_step(t0 + t, g_obs, drews[t])
            d_mean = d_means[t]
            if defer:       # action head's reverse + reverse chain in one launch (step 0's observation gradient is unused)
                d_in = pol.backward_data_act(d_action, acts[t], epss[t], g_ls_rows[t], d_mean, slot=t)
            else:
                _lib.check(L.vf_reparam_bwd(_ptr(d_action), _ptr(acts[t]), _ptr(log_std), _ptr(epss[t]), _ptr(d_mean), _ptr(g_ls), N, st))
                d_in = pol.backward(d_mean, None, None, accumulate=True, need_input_grad=t > 0, slot=t)
            g_obs = d_in.get("state") if t > 0 else None
        if defer:
            pol.weight_grad_slots(N, H, d_means, accumulate=True)
        pol.grad[pol.log_std_off:] = g_ls_rows.sum(dim=(0, 1)) if defer else g_ls.sum(dim=0)
        return loss_vec.mean() / self.world

    def _grad_autograd(self):
        """the same gradient with torch.autograd as the scheduler (two custom Functions wrap the kernels); kept as the
        cross-check of the reverse sweep and as the template for dropping in an arbitrary torch policy"""
        env, pol, N = self.env, self.policy, self.env.num_envs
        if self.reference_actor:
            raise NotImplementedError("use_autograd=True drives the one-head MlpPolicy actor (log_std parameter); the reference's "
                                      "two-head actor runs on the reverse sweep only")
        pol.grad.zero_()
        log_std = pol.log_std.detach().clone().requires_grad_(True)
        anchor = th.zeros(1, device=self.device, requires_grad=True)
        disc = th.ones(N, device=self.device)
        loss_vec = th.zeros(N, device=self.device)
        obs = env.get_observation()
        epss = th.randn((self.H, N, 4), device=self.device, generator=self._gen)     # same draw as the reverse sweep
        for t in range(self.H):
            mean = PolicyFunction.apply(pol, self.obs_keys, anchor, *[obs[k] for k in self.obs_keys])
            action = th.tanh(mean + log_std.exp() * epss[t])     # reparameterised squashed Gaussian (td_policies Actor)
            obs, reward, done, _ = env.step(action)
            loss_vec = loss_vec + -1 * reward * disc          # :123
            disc = disc * self.gamma * ~done + done           # :124
        loss = loss_vec.mean() / self.world
        loss.backward()
        pol.grad[pol.log_std_off:] = log_std.grad
        return loss.detach()

    def _apply(self, loss):
        env, pol, N = self.env, self.policy, self.env.num_envs
        parallel.allreduce_sum_(pol.grad)
        L, st = _lib.lib(), _lib.current_stream(self.device)
        self._opt_step += 1
        _lib.check(L.vf_sumsq(_ptr(pol.grad), pol.n_params, _ptr(self._sumsq), _ptr(self._scratch), st))
        pmap, packed = pol.pack_map()
        cfg = _lib.AdamCfg(self.lr, self.betas[0], self.betas[1], self.adam_eps, self.weight_decay, self.max_grad_norm,
                           self._opt_step, 0, _ptr(pmap), _ptr(packed))
        _lib.check(L.vf_adam_step(_ptr(pol.flat), _ptr(pol.grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), pol.n_params,
                                  _ptr(self._sumsq), C.byref(cfg), st))
        pol.mark_updated(packed_current=pmap is not None)
        env.detach()                                          # :134
        self.num_timesteps += self.H * N * self.world
        return loss.detach() * self.world

    def predict(self, obs, state=None, episode_start=None, deterministic: bool = False):
        """SB3-style predict (utils/evaluate.py:94): -> (action, None); deterministic: a = tanh(mean)"""
        N = obs[self.obs_keys[0]].shape[0]
        if self.reference_actor:         # shac.py:334-343 / MTDPolicy.predict
            mu, ls = self.policy.forward({k: obs[k].detach().contiguous() for k in self.obs_keys}, save_activations=False, slot=self.H + 1)
            if deterministic:
                return th.tanh(mu), None
            eps = th.randn((N, 4), device=self.device, generator=self._gen)
            action = th.empty((N, 4), device=self.device)
            self._head_fwd(mu, ls.contiguous(), eps, action)
            return action, None
        mean, _ = self.policy.forward({k: obs[k].detach().contiguous() for k in self.obs_keys}, save_activations=False,
                                      slot=self.H + 1, need_value=False)
        if deterministic:
            return th.tanh(mean), None
        eps = th.randn((N, 4), device=self.device, generator=self._gen)
        return th.tanh(mean + self.policy.log_std.exp() * eps), None

    # ---- checkpoints ----
    # MlpPolicy actor: the SB3-layout zip of checkpoint.py.  Reference actor (and SHAC): the reference pickles the whole SB3 policy
    # object (shac.py:328-332), which cannot exist here; a plain torch archive of the flat parameter buffers, the Adam moments and
    # step counters, the generator state and the spec the networks are rebuilt from.
    def _state(self):
        return {"actor": self.policy.flat.cpu(), "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(),
                "opt_step": int(self._opt_step), "num_timesteps": int(self.num_timesteps), "rng": self._gen.get_state().cpu(),
                "spec": dict(extractor=self._extractor, pi=self.policy.spec["pi"], qf=self._critic_arch, horizon=self.H,
                             gamma=self.gamma, learning_rate=self.lr, algo=type(self).__name__,
                             share_features_extractor=bool(getattr(self, "_share_extractor", False)))}

    def _load_state(self, d, load_optimizer=True):
        assert d["actor"].numel() == self.policy.flat.numel(), "actor: the archive holds a different network (pass the same policy_kwargs, or use load())"
        self.policy.flat.copy_(d["actor"])
        self.policy.mark_updated()
        if load_optimizer and "exp_avg" in d:
            self.exp_avg.copy_(d["exp_avg"])
            self.exp_avg_sq.copy_(d["exp_avg_sq"])
            self._opt_step, self.num_timesteps = int(d["opt_step"]), int(d.get("num_timesteps", 0))
            self._gen.set_state(d["rng"])

    def save(self, path: str):
        if not self.reference_actor:
            return checkpoint.save(self, path)
        th.save(self._state(), path if path.endswith(".pth") else path + ".pth")

    def set_parameters(self, path: str, load_optimizer: bool = True):
        if not self.reference_actor:
            return checkpoint.load_into(self, path, load_optimizer)
        self._load_state(th.load(path if path.endswith(".pth") else path + ".pth", map_location="cpu"), load_optimizer)
        return self

    @staticmethod
    def _policy_kwargs_from_spec(spec):
        """the stored network shapes as SB3-style policy_kwargs (what _make_reference_actor parses)"""
        return dict(features_extractor_class="StateTargetExtractor" if len(spec["extractor"]) > 1 else "StateExtractor",
                    features_extractor_kwargs={"net_arch": {k: {"layer": list(v)} for k, v in spec["extractor"].items()}},
                    net_arch=dict(pi=list(spec["pi"]), qf=list(spec["qf"])), activation_fn="relu",
                    share_features_extractor=bool(spec.get("share_features_extractor", False)))

    @classmethod
    def load(cls, path: str, env, **kwargs):
        if path.endswith(".pth") or (not path.endswith(".zip") and os.path.exists(path + ".pth")):
            d = th.load(path if path.endswith(".pth") else path + ".pth", map_location="cpu")
            spec, load_opt = d["spec"], kwargs.pop("load_optimizer", True)
            kwargs.setdefault("policy_kwargs", cls._policy_kwargs_from_spec(spec))
            kwargs.setdefault("horizon", spec["horizon"])
            # the hyper-parameters the archive was trained with (a resumed run silently trained with the constructor's
            # defaults next to the restored Adam moments); an explicit keyword argument still wins
            import inspect
            accepted = set(inspect.signature(cls.__init__).parameters)     # (BPTT.load on an archive SHAC wrote: tau / gradient_steps / lamda are not BPTT's)
            for k in ("gamma", "learning_rate", "tau", "gradient_steps", "lamda"):
                if k in spec and k in accepted:
                    kwargs.setdefault(k, spec[k])
            if cls is BPTT:
                kwargs.setdefault("policy", "MultiInputPolicy")
            algo = cls(env, **kwargs)
            algo._load_state(d, load_opt)
            return algo
        return checkpoint.load_into(cls(env, **checkpoint.ctor_kwargs_from_archive(path, kwargs)), path)

    def learn(self, total_timesteps: int, log_interval: Optional[int] = None):
        t0, start, it = time.time(), self.num_timesteps, 0
        while self.num_timesteps - start < total_timesteps:
            loss = self._update()
            it += 1
            if log_interval and it % log_interval == 0:
                self.logs["train/actor_loss"] = float(loss)
                print(f"[bptt] it {it} steps {self.num_timesteps} actor_loss {self.logs['train/actor_loss']:.4f}")
        th.cuda.synchronize(self.device)
        self.logs["train/actor_loss"] = float(loss)
        self.logs["time/fps"] = (self.num_timesteps - start) / max(time.time() - t0, 1e-9)
        return self.policy
