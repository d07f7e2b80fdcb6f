#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden-vector generator for the PPO update loop.  Runs ONLY in the build container.

Executes the reference's OWN ``PPO.train()`` (utils/algorithms/PPO.py:177-337), unmodified, on the reference's OWN
``CustomMultiInputActorCriticPolicy`` (utils/policies/policies.py:55-345: ``_build_mlp_extractor`` -> ``MlpExtractor2`` -> ``create_mlp``,
``evaluate_actions``, ``_get_action_dist_from_latent``) over its ``StateTargetExtractor`` (utils/policies/extractors.py:662-678) and its
``DictRolloutBuffer`` (utils/algorithms/common.py:218-352: ``compute_returns_and_advantage``, ``get``, ``_get_samples``) for
n_epochs x (full minibatches + the trailing partial one), and records what the HIP trainer (visfly_amd.ppo.PPO.train) must reproduce:
per optimiser step the loss, the value loss, the flat gradient ``clip_grad_norm_`` sees, the parameters after ``optimizer.step()``; per
run the epoch at which ``target_kl`` stops the loop, the number of optimiser steps, and everything ``train()`` hands to its logger.
Two runs: plain (the YAMLs' hyper-parameters) and one with ``clip_range_vf``, an entropy coefficient and a ``target_kl`` that trips in
the second epoch.

stable-baselines3 is not installable here.  What the reference's classes SUBCLASS or call of it is restated below, each piece marked
[SB3 2.2.1] with the file it restates -- constructor bookkeeping and the distribution's closed forms; what the loop computes is the
reference's code:
  ``ActorCriticPolicy`` / ``MultiInputActorCriticPolicy`` (common/policies.py: __init__, _build, extract_features, init_weights),
  ``MlpExtractor`` (common/torch_layers.py), ``make_proba_distribution`` / ``DiagGaussianDistribution.proba_distribution_net`` /
  ``SquashedDiagGaussianDistribution.log_prob`` / ``.entropy`` (common/distributions.py), ``BaseBuffer.swap_and_flatten`` / ``to_torch``
  / ``get_obs_shape`` (common/buffers.py, common/preprocessing.py), ``BaseAlgorithm._update_learning_rate`` (common/base_class.py).
``PPO.__init__`` (the SB3 ``OnPolicyAlgorithm`` set-up: env wrapping, logger, save paths) is NOT run: the instance is allocated with
``object.__new__`` and given exactly the attributes ``train()`` reads; ``train`` itself is the reference's function object.

The minibatch permutation: the reference's ``DictRolloutBuffer.get`` draws ``th.random.permutation`` -- an attribute torch does not have
(the class is a tensor port of SB3's numpy buffer, which draws ``np.random.permutation``; SB3's own buffer is what the reference's
``PPO`` instantiates through ``ori_PPO._setup_model``).  The generator provides the attribute as a feed of recorded permutations, so
that the fixture carries them; rows are SB3's ``swap_and_flatten`` order (env-major: row = env * n_steps + step).

Usage:  python oracle/gen_ppo_loop.py
"""
import os
import sys
import types
from functools import partial

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402
import gen_shac as GS  # noqa: E402

import torch as th  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch._dynamo  # noqa: E402,F401

OUT = G.OUT


def _install_ppo_sb3():
    sp = GS._install_sb3()
    m = sys.modules
    d = m["stable_baselines3.common.distributions"]
    get_action_dim = m["stable_baselines3.common.preprocessing"].get_action_dim
    DiagGaussianDistribution, Squashed, TanhBijector = d.DiagGaussianDistribution, d.SquashedDiagGaussianDistribution, d.TanhBijector

    # ---- [SB3 2.2.1] common/distributions.py ------------------------------------------------------------------------------------
    def proba_distribution_net(self, latent_dim, log_std_init=0.0):          # DiagGaussianDistribution.proba_distribution_net
        mean_actions = nn.Linear(latent_dim, self.action_dim)
        log_std = nn.Parameter(th.ones(self.action_dim) * log_std_init, requires_grad=True)
        return mean_actions, log_std
    DiagGaussianDistribution.proba_distribution_net = proba_distribution_net
    DiagGaussianDistribution.entropy = lambda self: self.distribution.entropy().sum(dim=1)

    def squashed_log_prob(self, actions, gaussian_actions=None):             # SquashedDiagGaussianDistribution.log_prob
        if gaussian_actions is None:
            gaussian_actions = TanhBijector.inverse(actions)
        log_prob = DiagGaussianDistribution.log_prob(self, gaussian_actions)
        log_prob -= th.sum(th.log(1 - actions ** 2 + self.epsilon), dim=1)
        return log_prob
    Squashed.log_prob = squashed_log_prob
    Squashed.entropy = lambda self: None                                     # "No analytical form"
    Squashed.proba_distribution = lambda self, mean_actions, log_std: (DiagGaussianDistribution.proba_distribution(self, mean_actions, log_std), self)[1]

    def make_proba_distribution(action_space, use_sde=False, dist_kwargs=None):
        assert isinstance(action_space, sp.Box) and not use_sde
        return DiagGaussianDistribution(get_action_dim(action_space), **(dist_kwargs or {}))
    d.make_proba_distribution, d.get_action_dim = make_proba_distribution, get_action_dim
    for n in ("CategoricalDistribution", "MultiCategoricalDistribution", "BernoulliDistribution"):
        setattr(d, n, type(n, (), {}))
    d.Distribution = type("Distribution", (), {})
    d.__all__ = ["DiagGaussianDistribution", "SquashedDiagGaussianDistribution", "CategoricalDistribution", "MultiCategoricalDistribution",
                 "BernoulliDistribution", "StateDependentNoiseDistribution", "make_proba_distribution", "get_action_dim", "Distribution",
                 "TanhBijector"]

    # ---- [SB3 2.2.1] common/torch_layers.py: MlpExtractor ------------------------------------------------------------------------
    class MlpExtractor(nn.Module):
        def __init__(self, feature_dim, net_arch, activation_fn, device="auto"):
            super().__init__()
            policy_net, value_net = [], []
            last_pi = last_vf = feature_dim
            pi_dims, vf_dims = (net_arch.get("pi", []), net_arch.get("vf", [])) if isinstance(net_arch, dict) else (net_arch, net_arch)
            for dim in pi_dims:
                policy_net += [nn.Linear(last_pi, dim), activation_fn()]
                last_pi = dim
            for dim in vf_dims:
                value_net += [nn.Linear(last_vf, dim), activation_fn()]
                last_vf = dim
            self.latent_dim_pi, self.latent_dim_vf = last_pi, last_vf
            self.policy_net, self.value_net = nn.Sequential(*policy_net), nn.Sequential(*value_net)

        def forward(self, features):
            return self.forward_actor(features), self.forward_critic(features)

        def forward_actor(self, features):
            return self.policy_net(features)

        def forward_critic(self, features):
            return self.value_net(features)

    # ---- [SB3 2.2.1] common/policies.py: ActorCriticPolicy ----------------------------------------------------------------------
    pol = m["stable_baselines3.common.policies"]
    BasePolicy = pol.BasePolicy
    tl = m["stable_baselines3.common.torch_layers"]

    class ActorCriticPolicy(BasePolicy):
        def __init__(self, observation_space, action_space, lr_schedule, net_arch=None, activation_fn=nn.Tanh, ortho_init=True,
                     use_sde=False, log_std_init=0.0, full_std=True, use_expln=False, squash_output=False,
                     features_extractor_class=None, features_extractor_kwargs=None, share_features_extractor=True,
                     normalize_images=True, optimizer_class=th.optim.Adam, optimizer_kwargs=None):
            if optimizer_kwargs is None:
                optimizer_kwargs = {}
                if optimizer_class == th.optim.Adam:
                    optimizer_kwargs["eps"] = 1e-5
            super().__init__(observation_space, action_space, features_extractor_class, features_extractor_kwargs,
                             optimizer_class=optimizer_class, optimizer_kwargs=optimizer_kwargs, squash_output=squash_output,
                             normalize_images=normalize_images)
            if net_arch is None:
                net_arch = dict(pi=[64, 64], vf=[64, 64])
            self.net_arch, self.activation_fn, self.ortho_init = net_arch, activation_fn, ortho_init
            self.share_features_extractor = share_features_extractor
            self.features_extractor = self.make_features_extractor()
            self.features_dim = self.features_extractor.features_dim
            if self.share_features_extractor:
                self.pi_features_extractor = self.vf_features_extractor = self.features_extractor
            else:
                self.pi_features_extractor = self.features_extractor
                self.vf_features_extractor = self.make_features_extractor()
            self.log_std_init = log_std_init
            assert not (squash_output and not use_sde), "squash_output=True is only available when using gSDE (use_sde=True)"
            self.use_sde, self.dist_kwargs = use_sde, None
            self.action_dist = make_proba_distribution(action_space, use_sde=use_sde, dist_kwargs=None)
            self._build(lr_schedule)

        def _build_mlp_extractor(self):
            self.mlp_extractor = MlpExtractor(self.features_dim, net_arch=self.net_arch, activation_fn=self.activation_fn, device=self.device)

        @staticmethod
        def init_weights(module, gain=1):                                    # BasePolicy.init_weights
            if isinstance(module, (nn.Linear, nn.Conv2d)):
                nn.init.orthogonal_(module.weight, gain=gain)
                if module.bias is not None:
                    module.bias.data.fill_(0.0)

        def _build(self, lr_schedule):
            self._build_mlp_extractor()
            latent_dim_pi = self.mlp_extractor.latent_dim_pi
            assert isinstance(self.action_dist, DiagGaussianDistribution)
            self.action_net, self.log_std = self.action_dist.proba_distribution_net(latent_dim=latent_dim_pi, log_std_init=self.log_std_init)
            self.value_net = nn.Linear(self.mlp_extractor.latent_dim_vf, 1)
            if self.ortho_init:
                module_gains = {self.features_extractor: np.sqrt(2), self.mlp_extractor: np.sqrt(2), self.action_net: 0.01, self.value_net: 1}
                if not self.share_features_extractor:
                    del module_gains[self.features_extractor]
                    module_gains[self.pi_features_extractor] = np.sqrt(2)
                    module_gains[self.vf_features_extractor] = np.sqrt(2)
                for module, gain in module_gains.items():
                    module.apply(partial(self.init_weights, gain=gain))
            self.optimizer = self.optimizer_class(self.parameters(), lr=lr_schedule(1), **self.optimizer_kwargs)

        def extract_features(self, obs, features_extractor=None):
            if self.share_features_extractor:
                return BasePolicy.extract_features(self, obs, self.features_extractor if features_extractor is None else features_extractor)
            return (BasePolicy.extract_features(self, obs, self.pi_features_extractor),
                    BasePolicy.extract_features(self, obs, self.vf_features_extractor))

    class MultiInputActorCriticPolicy(ActorCriticPolicy):                   # (only its default features extractor differs)
        pass

    pol.ActorCriticPolicy, pol.MultiInputActorCriticPolicy, pol.MlpExtractor = ActorCriticPolicy, MultiInputActorCriticPolicy, MlpExtractor
    pol.ActorCriticCnnPolicy = type("ActorCriticCnnPolicy", (ActorCriticPolicy,), {})
    tl.MlpExtractor = MlpExtractor

    # ---- [SB3 2.2.1] common/buffers.py: BaseBuffer pieces the reference's DictRolloutBuffer inherits -----------------------------
    BaseBuffer = m["stable_baselines3.common.buffers"].BaseBuffer

    def bb_init(self, buffer_size, observation_space, action_space, device="auto", n_envs=1):
        self.buffer_size, self.observation_space, self.action_space = buffer_size, observation_space, action_space
        if isinstance(observation_space, sp.Dict):                           # preprocessing.get_obs_shape
            self.obs_shape = {k: tuple(s.shape) for k, s in observation_space.spaces.items()}
        else:
            self.obs_shape = tuple(observation_space.shape)
        self.action_dim = int(np.prod(action_space.shape))
        self.pos, self.full, self.device, self.n_envs = 0, False, th.device("cpu"), n_envs

    def swap_and_flatten(arr):                                               # [n_steps, n_envs, ...] -> [n_envs * n_steps, ...], env-major
        shape = arr.shape
        if len(shape) < 3:
            shape = (*shape, 1)
        return arr.swapaxes(0, 1).reshape(shape[0] * shape[1], *shape[2:])
    BaseBuffer.__init__ = bb_init
    BaseBuffer.swap_and_flatten = staticmethod(swap_and_flatten)
    BaseBuffer.to_torch = lambda self, array, copy=True: th.as_tensor(array, device=self.device)

    # ---- [SB3 2.2.1] the algorithm base: only what train() calls on `self` ---------------------------------------------------------
    update_learning_rate = m["stable_baselines3.common.utils"].update_learning_rate

    class OriPPO:
        def _update_learning_rate(self, optimizers):                         # common/base_class.py: BaseAlgorithm._update_learning_rate
            self.logger.record("train/learning_rate", self.lr_schedule(self._current_progress_remaining))
            if not isinstance(optimizers, list):
                optimizers = [optimizers]
            for optimizer in optimizers:
                update_learning_rate(optimizer, self.lr_schedule(self._current_progress_remaining))

    for n in ("stable_baselines3.ppo", "stable_baselines3.ppo.ppo", "stable_baselines3.common.save_util", "stable_baselines3.common.env_util",
              "stable_baselines3.common.monitor", "stable_baselines3.common.vec_env.patch_gym"):
        if n not in m:
            G._auto(n)
    m["stable_baselines3.ppo.ppo"].PPO = OriPPO
    return sp


def _policy_linears(policy):
    """the policy's Linear layers in visfly_amd.MlpPolicy's schedule order: state extractor, target extractor, pi trunk, action_net,
    vf trunk, value_net"""
    fe = policy.features_extractor
    mods = list(GS._linears(fe.state_extractor, fe.target_extractor, policy.mlp_extractor.policy_net)) + [policy.action_net]
    mods += list(GS._linears(policy.mlp_extractor.value_net)) + [policy.value_net]
    return mods


def _flat(policy, grad=False):
    parts = []
    for l in _policy_linears(policy):
        for q in (l.weight, l.bias):
            parts.append((q.grad if grad else q.detach()).reshape(-1).numpy().astype(np.float32))
    q = policy.log_std
    parts.append((q.grad if grad else q.detach()).reshape(-1).numpy().astype(np.float32))
    return np.concatenate(parts)


# the schedules of the `ppo_loop_nav_sched` case, as functions of SB3's progress_remaining (tests/test_ppo_loop_gpu.py passes the same
# callables to visfly_amd.ppo.PPO): linear learning rate, clip ranges shrinking to half
def sched_lr(lr):
    return lambda p: lr * p


def sched_clip(c):
    return lambda p: c * (0.5 + 0.5 * p)


def gen_ppo_loop(name, T=8, N=64, batch_size=160, n_epochs=2, seed=21, lr=1e-3, ent_coef=0.0, clip_range_vf=None, target_kl=None,
                 adv_scale=1.0, progress=1.0, sched=False):
    sp = _install_ppo_sb3()
    import VisFly.utils.algorithms.PPO as RP
    import VisFly.utils.algorithms.common as RC
    import VisFly.utils.policies.policies as RPOL
    import VisFly.utils.policies.extractors as E

    # [SB3 2.2.1] common/type_aliases.py: the sample record of DictRolloutBuffer._get_samples (a NamedTuple; the stub module's attribute
    # is a mock)
    from typing import NamedTuple

    class DictRolloutBufferSamples(NamedTuple):
        observations: dict
        actions: th.Tensor
        old_values: th.Tensor
        old_log_prob: th.Tensor
        advantages: th.Tensor
        returns: th.Tensor
    RC.DictRolloutBufferSamples = DictRolloutBufferSamples

    rng = np.random.default_rng(seed)
    th.manual_seed(seed)
    gamma, lam, clip_range, vf_coef, max_grad_norm, weight_decay = 0.99, 0.95, 0.2, 0.5, 0.5, 1e-5
    obs_space = sp.Dict({"state": sp.Box(-1, 1, (13,)), "target": sp.Box(-1, 1, (3,))})
    act_space = sp.Box(-1, 1, (4,))
    # the YAMLs' policy_kwargs (exps/examples/alg_cfgs/*/PPO.yaml) with the state-vector extractor
    policy = RPOL.CustomMultiInputActorCriticPolicy(
        obs_space, act_space, lr_schedule=(sched_lr(lr) if sched else (lambda _: lr)), net_arch=dict(pi=[64, 64], vf=[64, 64]), activation_fn=nn.ReLU, ortho_init=False,
        log_std_init=-0.5, features_extractor_class=E.StateTargetExtractor,
        features_extractor_kwargs={"net_arch": {"state": {"layer": [128, 64]}, "target": {"layer": [128, 64]}}, "activation_fn": nn.ReLU},
        optimizer_kwargs={"weight_decay": weight_decay})
    assert type(policy.action_dist).__name__ == "SquashedDiagGaussianDistribution" and policy.share_features_extractor
    assert type(policy.mlp_extractor).__name__ == "MlpExtractor2"
    lin = _policy_linears(policy)
    assert len(list(policy.parameters())) == 2 * len(lin) + 1, "every parameter of the policy is in the flat layout"
    with th.no_grad():
        policy.action_net.weight.mul_(0.3)
    params0 = _flat(policy)

    # ---- a fixed rollout in the reference's own buffer: actions / values / log-probs from the policy's own forward() ----
    f = lambda a: th.from_numpy(np.ascontiguousarray(a, np.float32))
    obs_state = rng.normal(size=(T, N, 13)).astype(np.float32)
    obs_target = rng.normal(size=(T, N, 3)).astype(np.float32)
    rewards = (rng.normal(scale=0.3, size=(T, N)) * adv_scale).astype(np.float32)
    episode_starts = (rng.uniform(size=(T, N)) < 0.12).astype(np.float32)
    dones = (rng.uniform(size=N) < 0.25).astype(np.float32)
    eps_all = rng.standard_normal((T, N, 4)).astype(np.float32)
    import torch.distributions.normal as TDN
    feed = {"i": 0}

    def fed_standard_normal(shape, dtype, device):
        e = th.from_numpy(eps_all[feed["i"]].copy())
        feed["i"] += 1
        return e
    real_sn = TDN._standard_normal
    TDN._standard_normal = fed_standard_normal
    buf = RC.DictRolloutBuffer(T, obs_space, act_space, gae_lambda=lam, gamma=gamma, n_envs=N)
    with th.no_grad():
        for t in range(T):
            o = {"state": f(obs_state[t]), "target": f(obs_target[t])}
            actions, values, log_probs = policy(o)                            # policies.py:195-233: the rollout's forward
            # (the tensor port's add() cannot run -- common.py:307 assigns a numpy array into a torch tensor; it is not on the reference's
            # PPO path, SB3's own buffer is -- so the rows are written where add() would have put them)
            for k in o:
                buf.observations[k][t] = o[k]
            buf.actions[t], buf.rewards[t], buf.episode_starts[t] = actions.reshape(N, 4), f(rewards[t]), f(episode_starts[t])
            buf.values[t], buf.log_probs[t] = values.flatten(), log_probs
        buf.pos, buf.full = T, True
        last_values = f(rng.normal(size=N))
    TDN._standard_normal = real_sn
    buf.compute_returns_and_advantage(last_values, dones)
    assert buf.full
    rollout = dict(obs_state=obs_state, obs_target=obs_target, actions=G.f32(buf.actions), values=G.f32(buf.values),
                   log_probs=G.f32(buf.log_probs), advantages=G.f32(buf.advantages), returns=G.f32(buf.returns), rewards=rewards,
                   episode_starts=episode_starts, last_values=G.f32(last_values), dones=dones)

    # ---- the reference's PPO instance: allocated, not constructed; exactly the attributes train() reads ----
    algo = object.__new__(RP.PPO)
    logs = {}
    algo.logger = types.SimpleNamespace(record=lambda k, v, **kw: logs.__setitem__(k, v))
    # train() reads three things of the buffer: get() for the minibatches, and .values / .returns as NUMPY arrays for its
    # explained-variance log line (np.var) -- what SB3's own buffer holds; the tensor port's are tensors
    algo.policy = policy
    algo.rollout_buffer = types.SimpleNamespace(get=buf.get, values=G.f32(buf.values), returns=G.f32(buf.returns))
    # SB3's get_schedule_fn (common/utils.py [SB3 2.2.1]) turns a float into a constant callable and passes a callable through
    algo.lr_schedule = sched_lr(lr) if sched else (lambda _: lr)
    algo.clip_range = sched_clip(clip_range) if sched else (lambda _: clip_range)
    algo.clip_range_vf = None if clip_range_vf is None else (sched_clip(clip_range_vf) if sched else (lambda _: clip_range_vf))
    algo._current_progress_remaining = progress       # what learn() sets before train() (PPO.py:150-152)
    algo.n_epochs, algo.batch_size, algo.action_space, algo.use_sde = n_epochs, batch_size, act_space, False
    algo.normalize_advantage, algo.ent_coef, algo.vf_coef, algo.max_grad_norm = True, ent_coef, vf_coef, max_grad_norm
    algo.target_kl, algo.verbose, algo._n_updates = target_kl, 0, 0

    perms = []

    def fed_permutation(n):
        p = rng.permutation(n).astype(np.int64)
        perms.append(p)
        return th.from_numpy(p)
    th.random.permutation = fed_permutation                                   # (see the module docstring)

    rec = {"loss": [], "value_loss": [], "grad": [], "norm": [], "params": [], "mb_rows": []}
    real_clip, real_mse, real_backward = th.nn.utils.clip_grad_norm_, RP.F.mse_loss, th.Tensor.backward
    opt = policy.optimizer
    real_step = opt.step

    def clip_rec(params, max_norm, *a, **k):
        rec["grad"].append(_flat(policy, grad=True))
        r = real_clip(list(params), max_norm, *a, **k)
        rec["norm"].append(float(r))
        return r

    def mse_rec(a, b, *aa, **kk):
        r = real_mse(a, b, *aa, **kk)
        rec["value_loss"].append(float(r.detach()))
        rec["mb_rows"].append(int(a.numel()))
        return r

    def backward_rec(self, *a, **k):
        rec["loss"].append(float(self.detach()))
        return real_backward(self, *a, **k)

    def step_rec(*a, **k):
        r = real_step(*a, **k)
        rec["params"].append(_flat(policy))
        return r
    th.nn.utils.clip_grad_norm_, RP.F.mse_loss, th.Tensor.backward, opt.step = clip_rec, mse_rec, backward_rec, step_rec
    try:
        RP.PPO.train(algo)                                                    # <- the reference's own train(), unmodified
    finally:
        th.nn.utils.clip_grad_norm_, RP.F.mse_loss, th.Tensor.backward = real_clip, real_mse, real_backward
        del th.random.permutation
    n_steps_opt = len(rec["params"])
    assert len(rec["grad"]) == n_steps_opt == len(rec["loss"])
    stopped = len(rec["value_loss"]) > n_steps_opt            # the minibatch whose approx_kl tripped target_kl computed its losses, took no step
    epochs_started = len(perms)
    print(f"{name}: T={T} N={N} batch={batch_size} epochs started {epochs_started}/{n_epochs}, optimiser steps {n_steps_opt}, minibatch rows "
          f"{rec['mb_rows']}, early stop {stopped}, loss {rec['loss'][0]:.5f} -> {rec['loss'][-1]:.5f}, |g| {rec['norm'][0]:.3f} -> {rec['norm'][-1]:.3f}, "
          f"approx_kl (last epoch) {float(logs['train/approx_kl']):.5f}, clip_fraction {float(logs['train/clip_fraction']):.4f}")
    if target_kl is not None:
        assert stopped and epochs_started == 2, "the target_kl case is meant to stop inside the second epoch"
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), params0=params0, perms=np.stack(perms), eps=eps_all,
        loss=np.asarray(rec["loss"], np.float64), value_loss=np.asarray(rec["value_loss"], np.float64), grad=np.stack(rec["grad"]),
        grad_norm=np.asarray(rec["norm"], np.float64), params=np.stack(rec["params"]), mb_rows=np.asarray(rec["mb_rows"], np.int64),
        early_stop=np.bool_(stopped), epochs_started=np.int32(epochs_started), n_updates=np.int32(algo._n_updates),
        log_entropy_loss=np.float64(logs["train/entropy_loss"]), log_policy_gradient_loss=np.float64(logs["train/policy_gradient_loss"]),
        log_value_loss=np.float64(logs["train/value_loss"]), log_approx_kl=np.float64(logs["train/approx_kl"]),
        log_clip_fraction=np.float64(logs["train/clip_fraction"]), log_loss=np.float64(logs["train/loss"]),
        log_std_mean=np.float64(logs["train/std"]), log_explained_variance=np.float64(logs["train/explained_variance"]),
        log_learning_rate=np.float64(logs["train/learning_rate"]), log_clip_range=np.float64(logs["train/clip_range"]),
        log_clip_range_vf=np.float64(logs.get("train/clip_range_vf", -1.0)), log_n_updates=np.int32(logs["train/n_updates"]),
        progress_remaining=np.float64(progress), sched=np.bool_(sched),
        T=np.int32(T), N=np.int32(N), batch_size=np.int32(batch_size), n_epochs=np.int32(n_epochs), lr=np.float64(lr),
        gamma=np.float64(gamma), gae_lambda=np.float64(lam), clip_range=np.float64(clip_range), ent_coef=np.float64(ent_coef),
        vf_coef=np.float64(vf_coef), max_grad_norm=np.float64(max_grad_norm), weight_decay=np.float64(weight_decay),
        clip_range_vf=np.float64(-1.0 if clip_range_vf is None else clip_range_vf), target_kl=np.float64(-1.0 if target_kl is None else target_kl),
        adam_eps=np.float64(policy.optimizer.defaults["eps"]), log_std_init=np.float64(-0.5),
        label=np.asarray("the reference's own PPO.train (PPO.py:177-337) on its CustomMultiInputActorCriticPolicy / StateTargetExtractor / "
                         "create_mlp / DictRolloutBuffer; SB3 base classes restated (oracle/gen_ppo_loop.py)"),
        **rollout)


CASES = {
    "ppo_loop_nav": dict(),
    "ppo_loop_nav_kl": dict(seed=22, ent_coef=0.01, clip_range_vf=0.3, target_kl=None, lr=3e-3),      # target_kl filled in below
    # learning_rate / clip_range / clip_range_vf as callables of progress_remaining, train() entered at 40 % of the run remaining
    "ppo_loop_nav_sched": dict(seed=23, ent_coef=0.005, clip_range_vf=0.4, lr=2e-3, progress=0.4, sched=True),
}

if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, choices=[None] + list(CASES))
    ap.add_argument("--target-kl", type=float, default=None)
    a = ap.parse_args()
    if a.only is None:         # separate interpreters: the run patches module-level state of torch and of the imported reference
        import subprocess
        for n in CASES:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--only", n])
    else:
        kw = dict(CASES[a.only])
        if a.only == "ppo_loop_nav_kl":
            kw["target_kl"] = 0.004 if a.target_kl is None else a.target_kl
        gen_ppo_loop(a.only, **kw)
