#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden-vector generator for SHAC.  Runs ONLY in the build container.

Executes the reference's OWN SHAC loop -- ``TemporalDifferBase.learn`` (utils/algorithms/shac.py:185-326) with the reference's
own ``MTDPolicy`` / ``Actor`` / ``ContinuousCritic`` (utils/policies/td_policies.py:82-252,270-400), ``StateExtractor`` /
``create_mlp`` (utils/policies/extractors.py), ``SimpleRolloutBuffer`` / ``compute_td_returns`` (utils/algorithms/common.py:
893-923,1198-1250) and the differentiable ``HoverEnv`` (requires_grad=True, CR-sqrt oracle) -- for ONE iteration
(H control steps, one actor update, ``gradient_steps`` critic updates) and records every intermediate the HIP path must
reproduce: the horizon buffer, next values, actor loss INCLUDING the bootstrap term, its flat gradient, the clipped Adam step,
TD-lambda returns, and per critic step the twin-Q loss, flat gradient, parameters after Adam and the Polyak-updated targets.

stable-baselines3 is not installable here.  Its classes that the reference's modules SUBCLASS are restated below, each marked
[SB3 2.2.1] with the file it restates (constructor bookkeeping only -- what the loop computes is the reference's code):
``BasePolicy`` / ``BaseModel`` (common/policies.py), ``ContinuousCritic`` (common/policies.py), ``sac.policies.Actor`` /
``SACPolicy`` (sac/policies.py), ``SquashedDiagGaussianDistribution`` / ``TanhBijector`` (common/distributions.py),
``get_actor_critic_arch`` (common/torch_layers.py), ``polyak_update`` / ``get_parameters_by_name`` / ``get_schedule_fn`` /
``update_learning_rate`` (common/utils.py).

Reference defect C-10 (repaired, label `repaired-oracle`): ``SimpleRolloutBuffer.flatten`` stacks the observations to
(H, N, 13) (``TensorDict.stack``, utils/type.py:165-175 -- the flattening reshape is commented out) but the actions to (H*N, 4)
(``th.vstack``), so ``ContinuousCritic.forward``'s ``th.cat([features, actions], dim=-1)`` raises for every H > 1.  Minimal
repair: the commented-out reshape of ``TensorDict.stack`` is applied (rows t*N + i, the order of the flattened actions / returns).

The exploration noise: ``Normal.rsample`` draws from torch's global generator, which the env's auto-reset shares.  The generator
replaces ``torch.distributions.normal._standard_normal`` by a feed of recorded, platform-stable draws (numpy), so that (a) the
fixture carries the noise and (b) the global stream is consumed by the env alone, in the order `spawn="replay"` replays.

Usage:  python oracle/gen_shac.py
"""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402  (stubs, fake SceneManager, CR-sqrt patch, constants extraction)

import torch as th  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch._dynamo  # noqa: E402,F401  (torch.optim imports it lazily; it must not meet the stub modules below)

OUT = G.OUT


# ----------------------------------------------------------------------------------------------------------------------
# [SB3 2.2.1] restatements: only what the reference's td_policies.py / shac.py subclass or call
# ----------------------------------------------------------------------------------------------------------------------
def _install_sb3():
    G.import_envs()
    G._sb3_stubs()
    sp = sys.modules["gymnasium.spaces"]

    def get_action_dim(space):                                   # common/preprocessing.py
        return int(np.prod(space.shape))

    def get_actor_critic_arch(net_arch):                         # common/torch_layers.py
        if isinstance(net_arch, list):
            return net_arch, net_arch
        assert isinstance(net_arch, dict) and "pi" in net_arch and "qf" in net_arch
        return net_arch["pi"], net_arch["qf"]

    class BaseModel(nn.Module):                                  # common/policies.py: BaseModel
        def __init__(self, observation_space, action_space, features_extractor_class=None, features_extractor_kwargs=None,
                     features_extractor=None, normalize_images=True, optimizer_class=th.optim.Adam, optimizer_kwargs=None):
            super().__init__()
            self.observation_space, self.action_space = observation_space, action_space
            self.features_extractor = features_extractor
            self.normalize_images = normalize_images
            self.optimizer_class = optimizer_class
            self.optimizer_kwargs = {} if optimizer_kwargs is None else optimizer_kwargs
            self.features_extractor_class = features_extractor_class
            self.features_extractor_kwargs = {} if features_extractor_kwargs is None else features_extractor_kwargs

        def _update_features_extractor(self, net_kwargs, features_extractor=None):
            net_kwargs = net_kwargs.copy()
            if features_extractor is None:
                features_extractor = self.make_features_extractor()
            net_kwargs.update(dict(features_extractor=features_extractor, features_dim=features_extractor.features_dim))
            return net_kwargs

        def make_features_extractor(self):
            return self.features_extractor_class(self.observation_space, **self.features_extractor_kwargs)

        def extract_features(self, obs, features_extractor):
            return features_extractor(obs)                        # (preprocess_obs is the identity for Box observations)

        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return th.device("cpu")

        def set_training_mode(self, mode):
            self.train(mode)

    class BasePolicy(BaseModel):                                 # common/policies.py: BasePolicy
        def __init__(self, *args, squash_output=False, **kwargs):
            super().__init__(*args, **kwargs)
            self._squash_output = squash_output

        @property
        def squash_output(self):
            return self._squash_output

    class ContinuousCritic(BaseModel):                           # common/policies.py: ContinuousCritic.__init__
        def __init__(self, observation_space, action_space, net_arch, features_extractor, features_dim, activation_fn=nn.ReLU,
                     normalize_images=True, n_critics=2, share_features_extractor=True):
            super().__init__(observation_space, action_space, features_extractor=features_extractor, normalize_images=normalize_images)
            self.share_features_extractor, self.n_critics = share_features_extractor, n_critics
            self.q_networks = []                                  # the reference's subclass rebuilds them with ITS create_mlp

    class TanhBijector:                                          # common/distributions.py
        @staticmethod
        def atanh(x):
            return 0.5 * (x.log1p() - (-x).log1p())

        @staticmethod
        def inverse(y):
            eps = th.finfo(y.dtype).eps
            return TanhBijector.atanh(y.clamp(min=-1.0 + eps, max=1.0 - eps))

    class DiagGaussianDistribution:                              # common/distributions.py
        def __init__(self, action_dim):
            self.action_dim, self.mean_actions, self.log_std, self.distribution = action_dim, None, None, None

        def proba_distribution(self, mean_actions, log_std):
            from torch.distributions import Normal
            self.distribution = Normal(mean_actions, th.ones_like(mean_actions) * log_std.exp())
            return self

        def log_prob(self, actions):
            return self.distribution.log_prob(actions).sum(dim=1)

        def sample(self):
            return self.distribution.rsample()

        def mode(self):
            return self.distribution.mean

        def get_actions(self, deterministic=False):
            return self.mode() if deterministic else self.sample()

        def actions_from_params(self, mean_actions, log_std, deterministic=False):
            self.proba_distribution(mean_actions, log_std)
            return self.get_actions(deterministic=deterministic)

    class SquashedDiagGaussianDistribution(DiagGaussianDistribution):   # common/distributions.py
        def __init__(self, action_dim, epsilon=1e-6):
            super().__init__(action_dim)
            self.epsilon, self.gaussian_actions = epsilon, None

        def sample(self):
            self.gaussian_actions = super().sample()
            return th.tanh(self.gaussian_actions)

        def mode(self):
            self.gaussian_actions = super().mode()
            return th.tanh(self.gaussian_actions)

        def log_prob_from_params(self, mean_actions, log_std):
            action = self.actions_from_params(mean_actions, log_std)
            return action, self.log_prob(action, self.gaussian_actions)

    class SACActor(BasePolicy):                                  # sac/policies.py: Actor.__init__ (no-SDE branch)
        def __init__(self, observation_space, action_space, net_arch, features_extractor, features_dim, activation_fn=nn.ReLU,
                     use_sde=False, log_std_init=-3, full_std=True, use_expln=False, clip_mean=2.0, normalize_images=True):
            super().__init__(observation_space, action_space, features_extractor=features_extractor,
                             normalize_images=normalize_images, squash_output=True)
            assert not use_sde
            self.use_sde, self.net_arch, self.features_dim, self.activation_fn = use_sde, net_arch, features_dim, activation_fn
            action_dim = get_action_dim(action_space)
            last_layer_dim = net_arch[-1] if len(net_arch) > 0 else features_dim
            self.action_dist = SquashedDiagGaussianDistribution(action_dim)
            self.mu = nn.Linear(last_layer_dim, action_dim)
            self.log_std = nn.Linear(last_layer_dim, action_dim)

    class SACPolicy(BasePolicy):                                 # sac/policies.py: SACPolicy.__init__ / _build
        def __init__(self, observation_space, action_space, lr_schedule, net_arch=None, activation_fn=nn.ReLU, use_sde=False,
                     log_std_init=-3, use_expln=False, clip_mean=2.0, features_extractor_class=None,
                     features_extractor_kwargs=None, normalize_images=True, optimizer_class=th.optim.Adam, optimizer_kwargs=None,
                     n_critics=2, share_features_extractor=False):
            super().__init__(observation_space, action_space, features_extractor_class, features_extractor_kwargs,
                             optimizer_class=optimizer_class, optimizer_kwargs=optimizer_kwargs, squash_output=True,
                             normalize_images=normalize_images)
            if net_arch is None:
                net_arch = [256, 256]
            actor_arch, critic_arch = get_actor_critic_arch(net_arch)
            self.net_arch, self.activation_fn = net_arch, activation_fn
            self.net_args = {"observation_space": observation_space, "action_space": action_space, "net_arch": actor_arch,
                             "activation_fn": activation_fn, "normalize_images": normalize_images}
            self.actor_kwargs = self.net_args.copy()
            self.actor_kwargs.update({"use_sde": use_sde, "log_std_init": log_std_init, "use_expln": use_expln, "clip_mean": clip_mean})
            self.critic_kwargs = self.net_args.copy()
            self.critic_kwargs.update({"n_critics": n_critics, "net_arch": critic_arch, "share_features_extractor": share_features_extractor})
            self.share_features_extractor = share_features_extractor
            self._build(lr_schedule)

        def _build(self, lr_schedule):
            self.actor = self.make_actor()
            self.actor.optimizer = self.optimizer_class(self.actor.parameters(), lr=lr_schedule(1), **self.optimizer_kwargs)
            if self.share_features_extractor:
                self.critic = self.make_critic(features_extractor=self.actor.features_extractor)
                critic_parameters = [p for n, p in self.critic.named_parameters() if "features_extractor" not in n]
            else:
                self.critic = self.make_critic(features_extractor=None)
                critic_parameters = list(self.critic.parameters())
            self.critic_target = self.make_critic(features_extractor=None)
            self.critic_target.load_state_dict(self.critic.state_dict())
            self.critic.optimizer = self.optimizer_class(critic_parameters, lr=lr_schedule(1), **self.optimizer_kwargs)
            self.critic_target.set_training_mode(False)

    def polyak_update(params, target_params, tau):               # common/utils.py
        with th.no_grad():
            for param, target_param in zip(params, target_params):
                target_param.data.mul_(1 - tau)
                th.add(target_param.data, param.data, alpha=tau, out=target_param.data)

    def get_parameters_by_name(model, included_names):           # common/utils.py
        return [p for n, p in model.state_dict().items() if any(k in n for k in included_names)]

    def get_schedule_fn(value_schedule):                         # common/utils.py
        if isinstance(value_schedule, (float, int)):
            v = float(value_schedule)
            return lambda _: v
        assert callable(value_schedule)
        return value_schedule

    def update_learning_rate(optimizer, learning_rate):          # common/utils.py
        for g in optimizer.param_groups:
            g["lr"] = learning_rate

    def safe_mean(arr):
        return np.nan if len(arr) == 0 else float(np.mean(arr))

    m = sys.modules
    pol = m["stable_baselines3.common.policies"]
    pol.BasePolicy, pol.ContinuousCritic, pol.BaseModel = BasePolicy, ContinuousCritic, BaseModel
    m["stable_baselines3.common.preprocessing"].get_action_dim = get_action_dim
    tl = m["stable_baselines3.common.torch_layers"]
    tl.get_actor_critic_arch = get_actor_critic_arch
    for n in ("CombinedExtractor", "FlattenExtractor", "NatureCNN"):
        setattr(tl, n, type(n, (nn.Module,), {}))
    d = m["stable_baselines3.common.distributions"]
    d.SquashedDiagGaussianDistribution, d.TanhBijector, d.DiagGaussianDistribution = SquashedDiagGaussianDistribution, TanhBijector, DiagGaussianDistribution
    d.StateDependentNoiseDistribution = type("StateDependentNoiseDistribution", (), {})
    d.SelfSquashedDiagGaussianDistribution = SquashedDiagGaussianDistribution
    for n in ("stable_baselines3.sac", "stable_baselines3.sac.policies", "stable_baselines3.common.logger"):
        if n not in m:
            G._auto(n)
    m["stable_baselines3.sac.policies"].Actor, m["stable_baselines3.sac.policies"].SACPolicy = SACActor, SACPolicy
    u = m["stable_baselines3.common.utils"]
    u.polyak_update, u.get_parameters_by_name, u.get_schedule_fn = polyak_update, get_parameters_by_name, get_schedule_fn
    u.update_learning_rate, u.safe_mean = update_learning_rate, safe_mean
    return sp


def _linears(*mods):
    return [l for mod in mods for l in mod.modules() if isinstance(l, nn.Linear)]


def _flat(mods, grad=False):
    out = []
    for l in mods:
        for q in (l.weight, l.bias):
            out.append(G.f32(q.grad if grad else q).reshape(-1))
    return np.concatenate(out)


def gen_shac(name="shac_hover", N=64, H=8, seed=42, gradient_steps=3, lr=1e-3, tau=0.005, share=False):
    """share: MTDPolicy(share_features_extractor=True) -- the critic runs the ACTOR's features extractor under no_grad and its optimiser
    leaves it alone (SACPolicy._build; td_policies.py:127), the target keeps its own, Polyak-averaged copy (fixture shac_hover_shared)"""
    sp = _install_sb3()
    HoverEnvShim, _, _ = G.import_envs()
    G.use_cr_sqrt(True)
    import VisFly.utils.type as T

    def stack_flat(x_list):                                      # defect C-10: TensorDict.stack with its commented-out reshape
        r = T.TensorDict({})
        for key in x_list[0].keys():
            v = th.stack([x[key] for x in x_list])
            r[key] = th.reshape(v, (-1, *v.shape[2:]))
        return r
    T.TensorDict.stack = staticmethod(stack_flat)
    import VisFly.utils.algorithms.shac as S
    from VisFly.utils.policies.td_policies import MTDPolicy
    import VisFly.utils.policies.extractors as E

    # ---- exploration noise feed (see module docstring) ----
    rng = np.random.default_rng(seed + 9)
    eps_all = rng.standard_normal((2 * H, N, 4)).astype(np.float32)
    feed = {"i": 0}
    import torch.distributions.normal as TDN

    def fed_standard_normal(shape, dtype, device):
        assert tuple(shape) == (N, 4), shape
        e = th.from_numpy(eps_all[feed["i"]].copy())
        feed["i"] += 1
        return e
    TDN._standard_normal = fed_standard_normal

    # ---- env: spawn box reaching down to the floor (a few collisions = episode_done) + 5-step episodes (truncations) ----
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 0.65], "half": [1.0, 1.0, 0.6]}}]}}
    max_steps = 5
    env = HoverEnvShim(num_agent_per_scene=N, num_scene=1, seed=seed, visual=False, dynamics_kwargs=dict(G.ENV_DYN), device="cpu",
                       requires_grad=True, tensor_output=True, max_episode_steps=max_steps, random_kwargs=spawn)
    consts = G.extract_consts(env.envs.dynamics)
    policy_kwargs = dict(features_extractor_class=E.StateExtractor,
                         features_extractor_kwargs={"net_arch": {"state": {"layer": [128, 64]}}},
                         net_arch=dict(pi=[64, 64], qf=[64, 64]), activation_fn=nn.ReLU, share_features_extractor=bool(share))
    th.manual_seed(seed + 1)                                     # network initialisation (nn.Linear defaults)
    import copy as _copy
    real_deepcopy = _copy.deepcopy
    S.deepcopy = lambda e: e if isinstance(e, HoverEnvShim) else real_deepcopy(e)     # eval env: never stepped within one iteration
    algo = S.TemporalDifferBase(env, MTDPolicy, policy_kwargs=policy_kwargs, learning_rate=lr, horizon=H, tau=tau, gamma=0.99,
                                gradient_steps=gradient_steps, device="cpu", seed=seed, save_path="/tmp/vf_shac_gen", dump_step=1e12)
    algo._create_logger = lambda **kw: types.SimpleNamespace(record=lambda *a, **k: None, dump=lambda *a, **k: None)
    actor, critic, target = algo.policy.actor, algo.policy.critic, algo.policy.critic_target
    assert type(actor).__module__.endswith("td_policies") and type(critic).__module__.endswith("td_policies")
    assert (critic.features_extractor is actor.features_extractor) == bool(share)
    # de-correlate the two actor trunks (log_latent_pi is a deepcopy of latent_pi) and the heads, so that a wiring mix-up shows
    with th.no_grad():
        g = th.Generator().manual_seed(seed + 2)
        for l in _linears(actor.log_latent_pi, actor.log_std):
            l.weight.add_(th.randn(l.weight.shape, generator=g) * 0.05)
        actor.log_std.bias.fill_(-1.0)
        actor.mu.weight.mul_(0.5)
    target.load_state_dict(critic.state_dict())
    # layer order = visfly_amd's schedule: extractor, trunk 0 (+ head), trunk 1 (+ head)
    a_lin = _linears(actor.features_extractor.state_extractor, actor.latent_pi) + [actor.mu] + _linears(actor.log_latent_pi) + [actor.log_std]
    c_lin = _linears(critic.features_extractor.state_extractor, critic.qf0, critic.qf1)
    t_lin = _linears(target.features_extractor.state_extractor, target.qf0, target.qf1)
    assert len(list(actor.parameters())) == 2 * len(a_lin) and len(list(critic.parameters())) == 2 * len(c_lin)
    save = {"actor_params0": _flat(a_lin), "critic_params0": _flat(c_lin)}

    # ---- the env must be re-seeded AFTER network construction: Dynamics.__init__ seeded the global stream, __init__ of the
    # algorithm reset the env (shac.py:121-123) and set_seed re-seeds.  Re-seed + reset here so that the run below starts from
    # a documented point of the stream: manual_seed(seed) -> env.reset() -> H steps
    th.manual_seed(seed)
    env.reset()
    fs_init = G.f32(env.envs.dynamics.full_state)

    rec = {"clip": []}
    real_clip = th.nn.utils.clip_grad_norm_

    def clip_rec(params, max_norm, *a, **k):
        params = list(params)
        q0 = next(critic.qf0.parameters())             # (with a shared extractor the critic's first parameter IS the actor's)
        which = "critic" if any(p is q0 for p in params) else "actor"
        lin = a_lin if which == "actor" else c_lin
        rec["clip"].append((which, _flat(lin, grad=True), float(max_norm)))
        return real_clip(params, max_norm, *a, **k)
    th.nn.utils.clip_grad_norm_ = clip_rec

    steps = {"actor": [], "critic": [], "target": []}
    for which, mod, lin in (("actor", actor, a_lin), ("critic", critic, c_lin)):
        opt = mod.optimizer
        real_step = opt.step

        def step_rec(*a, _real=real_step, _w=which, _lin=lin, **k):
            r = _real(*a, **k)
            steps[_w].append(_flat(_lin))
            return r
        opt.step = step_rec
    real_polyak = S.polyak_update
    pol_calls = {"n": 0}

    def polyak_rec(params, target_params, tau):
        params, target_params = list(params), list(target_params)
        real_polyak(params, target_params, tau)
        if len(params) and pol_calls["n"] % 2 == 0:               # the second call per step is the (empty) batch-norm statistics
            steps["target"].append(_flat(t_lin))
        pol_calls["n"] += 1
    S.polyak_update = polyak_rec

    # buffer + losses
    losses = {"actor": [], "critic": []}
    real_mse = th.nn.functional.mse_loss

    def mse_rec(a, b, *aa, **kk):
        v = real_mse(a, b, *aa, **kk)
        losses["critic"].append(float(v))
        return v
    th.nn.functional.mse_loss = mse_rec
    buf = algo.rollout_buffer
    real_compute = buf.compute_returns
    snap = {}

    def compute_rec():
        snap["reward"] = np.stack([G.f32(x) for x in buf.reward])
        snap["done"] = np.stack([x.numpy().astype(np.uint8) for x in buf.done])
        snap["episode_done"] = np.stack([x.numpy().astype(np.uint8) for x in buf.episode_done])
        snap["next_value"] = np.stack([G.f32(x) for x in buf.value])
        snap["obs"] = np.stack([G.f32(x["state"]) for x in buf.obs])
        snap["next_obs"] = np.stack([G.f32(x["state"]) for x in buf.next_obs])
        snap["action"] = np.stack([G.f32(x) for x in buf.action])
        real_compute()
        snap["returns"] = G.f32(buf.returns).reshape(H, N)
    buf.compute_returns = compute_rec
    real_backward = th.Tensor.backward

    def backward_rec(self, *a, **k):
        if not losses["actor"]:
            losses["actor"].append(float(self.detach()))
        return real_backward(self, *a, **k)
    th.Tensor.backward = backward_rec
    try:
        algo.learn(total_timesteps=H * N)
    finally:
        th.Tensor.backward = real_backward
        th.nn.functional.mse_loss = real_mse
        th.nn.utils.clip_grad_norm_ = real_clip
    assert feed["i"] == 2 * H, feed
    assert len(steps["actor"]) == 1 and len(steps["critic"]) == gradient_steps == len(steps["target"]), {k: len(v) for k, v in steps.items()}
    assert [w for w, _, _ in rec["clip"]] == ["actor"] + ["critic"] * gradient_steps
    done, epd = snap["done"].astype(bool), snap["episode_done"].astype(bool)
    print(f"{name}: N={N} H={H} dones={int(done.sum())} episode_done={int(epd.sum())} truncated={int((done & ~epd).sum())} "
          f"actor_loss={losses['actor'][0]:.6f} |g_actor|={np.linalg.norm(rec['clip'][0][1]):.4f} "
          f"critic_loss={[round(x, 6) for x in losses['critic']]} |g_critic|={[round(float(np.linalg.norm(c[1])), 4) for c in rec['clip'][1:]]}")
    assert epd.sum() > 0 and (done & ~epd).sum() > 0
    save.update(
        eps=eps_all, fs_init=fs_init, seed=np.int32(seed), max_episode_steps=np.int32(max_steps), spawn=np.asarray(repr(spawn)),
        dyn_kw=np.asarray(repr(dict(G.ENV_DYN))), H=np.int32(H), gamma=np.float64(0.99), lamda=np.float64(0.95), tau=np.float64(tau),
        lr=np.float64(lr), gradient_steps=np.int32(gradient_steps), max_grad_norm=np.float64(0.5),
        log_std_min=np.float32(-10), log_std_max=np.float32(2),
        actor_loss=np.float64(losses["actor"][0]), actor_grad=rec["clip"][0][1], actor_params1=steps["actor"][0],
        critic_loss=np.asarray(losses["critic"], np.float64), critic_grad=np.stack([c[1] for c in rec["clip"][1:]]),
        critic_params=np.stack(steps["critic"]), target_params=np.stack(steps["target"]),
        label=np.asarray("repaired-oracle (C-10: observations flattened like the actions); reference learn() loop, Actor, "
                         "ContinuousCritic, StateExtractor, create_mlp, SimpleRolloutBuffer, compute_td_returns; SB3 base classes restated"),
        **{"buf_" + k: v for k, v in snap.items()}, **{"c_" + k: v for k, v in consts.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


def gen_bptt_loop(name="bptt_loop_hover", N=64, H=8, seed=42, lr=1e-3):
    """ONE iteration of the reference's own ``BPTT.learn`` (utils/algorithms/BPTT.py:77-134) with its own ``MTDPolicy.actor`` -- the
    SAC-style ``Actor`` with the state-dependent ``log_std`` head (utils/policies/td_policies.py:146-252), which is what
    ``self.policy.actor.action_log_prob(obs)`` (:113) runs -- over the differentiable HoverEnv (CR oracle), exploration noise fed
    from the fixture.  Recorded: the (clipped) actions, rewards and done flags of the H steps, ``actor_loss`` (:127), its flat
    gradient as ``clip_grad_norm_(.., 0.5)`` sees it (:130), the actor parameters after ``optimizer.step()`` (:133).  5-step
    episodes and a spawn box that reaches the floor, so that the discount recurrence (:124) sees truncations and true episode ends
    inside the horizon."""
    _install_sb3()
    HoverEnvShim, _, _ = G.import_envs()
    G.use_cr_sqrt(True)
    import VisFly.utils.algorithms.shac as S
    import VisFly.utils.algorithms.BPTT as B
    from VisFly.utils.policies.td_policies import MTDPolicy
    import VisFly.utils.policies.extractors as E

    rng = np.random.default_rng(seed + 19)
    eps_all = rng.standard_normal((H, N, 4)).astype(np.float32)
    feed = {"i": 0}
    import torch.distributions.normal as TDN

    def fed_standard_normal(shape, dtype, device):
        assert tuple(shape) == (N, 4), shape
        e = th.from_numpy(eps_all[feed["i"]].copy())
        feed["i"] += 1
        return e
    TDN._standard_normal = fed_standard_normal

    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 0.65], "half": [1.0, 1.0, 0.6]}}]}}
    max_steps = 5
    env = HoverEnvShim(num_agent_per_scene=N, num_scene=1, seed=seed, visual=False, dynamics_kwargs=dict(G.ENV_DYN), device="cpu",
                       requires_grad=True, tensor_output=True, max_episode_steps=max_steps, random_kwargs=spawn)
    consts = G.extract_consts(env.envs.dynamics)
    policy_kwargs = dict(features_extractor_class=E.StateExtractor,
                         features_extractor_kwargs={"net_arch": {"state": {"layer": [128, 64]}}},
                         net_arch=dict(pi=[64, 64], qf=[64, 64]), activation_fn=nn.ReLU, share_features_extractor=False)
    th.manual_seed(seed + 1)
    import copy as _copy
    real_deepcopy = _copy.deepcopy
    S.deepcopy = lambda e: e if isinstance(e, HoverEnvShim) else real_deepcopy(e)
    B.deepcopy = S.deepcopy
    # BPTT.learn wraps its loop in tqdm; a transparent stand-in keeps the generator's output clean (pbar.n / pbar.update only)

    class _Bar:
        def __init__(self, total=None):
            self.n = 0

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def update(self, k):
            self.n += k
    B.tqdm = _Bar
    algo = B.BPTT(env, MTDPolicy, policy_kwargs=policy_kwargs, learning_rate=lr, horizon=H, gamma=0.99, device="cpu", seed=seed,
                  save_path="/tmp/vf_bptt_gen", dump_step=1e12)
    algo._create_logger = lambda **kw: types.SimpleNamespace(record=lambda *a, **k: None, dump=lambda *a, **k: None)
    actor = algo.policy.actor
    assert type(actor).__module__.endswith("td_policies") and type(actor).__name__ == "Actor"
    with th.no_grad():                                            # de-correlate the two trunks and the heads (as gen_shac)
        g = th.Generator().manual_seed(seed + 2)
        for l in _linears(actor.log_latent_pi, actor.log_std):
            l.weight.add_(th.randn(l.weight.shape, generator=g) * 0.05)
        actor.log_std.bias.fill_(-1.0)
        actor.mu.weight.mul_(0.5)
    a_lin = _linears(actor.features_extractor.state_extractor, actor.latent_pi) + [actor.mu] + _linears(actor.log_latent_pi) + [actor.log_std]
    assert len(list(actor.parameters())) == 2 * len(a_lin)
    save = {"actor_params0": _flat(a_lin)}
    th.manual_seed(seed)
    env.reset()
    fs_init = G.f32(env.envs.dynamics.full_state)

    rec = {"action": [], "reward": [], "done": [], "clip": [], "loss": [], "params": []}
    real_step_env = env.step

    def step_rec(a, *aa, **kk):
        rec["action"].append(G.f32(a))
        o, r, d, info = real_step_env(a, *aa, **kk)
        rec["reward"].append(G.f32(r))
        rec["done"].append(d.numpy().astype(np.uint8))
        return o, r, d, info
    env.step = step_rec
    real_clip = th.nn.utils.clip_grad_norm_

    def clip_rec(params, max_norm, *a, **k):
        params = list(params)
        rec["clip"].append((_flat(a_lin, grad=True), float(max_norm)))
        return real_clip(params, max_norm, *a, **k)
    th.nn.utils.clip_grad_norm_ = clip_rec
    opt = actor.optimizer
    real_opt_step = opt.step

    def opt_step_rec(*a, **k):
        r = real_opt_step(*a, **k)
        rec["params"].append(_flat(a_lin))
        return r
    opt.step = opt_step_rec
    real_backward = th.Tensor.backward

    def backward_rec(self, *a, **k):
        if not rec["loss"]:
            rec["loss"].append(float(self.detach()))
        return real_backward(self, *a, **k)
    th.Tensor.backward = backward_rec
    try:
        algo.learn(total_timesteps=H * N)
    finally:
        th.Tensor.backward = real_backward
        th.nn.utils.clip_grad_norm_ = real_clip
    assert feed["i"] == H and len(rec["action"]) == H and len(rec["clip"]) == 1 and len(rec["params"]) == 1, (feed, len(rec["action"]))
    done = np.stack(rec["done"]).astype(bool)
    gnorm = float(np.linalg.norm(rec["clip"][0][0]))
    print(f"{name}: N={N} H={H} dones={int(done.sum())} actor_loss={rec['loss'][0]:.6f} |g|={gnorm:.4f} (clip {rec['clip'][0][1]})")
    assert done.sum() > 0 and rec["clip"][0][1] == 0.5
    save.update(
        eps=eps_all, fs_init=fs_init, seed=np.int32(seed), max_episode_steps=np.int32(max_steps), spawn=np.asarray(repr(spawn)),
        dyn_kw=np.asarray(repr(dict(G.ENV_DYN))), H=np.int32(H), gamma=np.float64(0.99), lr=np.float64(lr), max_grad_norm=np.float64(0.5),
        log_std_min=np.float32(-10), log_std_max=np.float32(2),
        action=np.stack(rec["action"]), reward=np.stack(rec["reward"]), done=np.stack(rec["done"]),
        actor_loss=np.float64(rec["loss"][0]), actor_grad=rec["clip"][0][0], actor_params1=rec["params"][0],
        label=np.asarray("cr-oracle; the reference's own BPTT.learn loop (BPTT.py:100-134), MTDPolicy.actor = td_policies.Actor, StateExtractor, "
                         "create_mlp; SB3 base classes restated (oracle/gen_shac.py)"),
        **{"c_" + k: v for k, v in consts.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, choices=[None, "shac_hover", "shac_hover_shared", "bptt_loop_hover"])
    a = ap.parse_args()
    if a.only == "bptt_loop_hover":
        gen_bptt_loop()
    elif a.only == "shac_hover":
        gen_shac()
    elif a.only == "shac_hover_shared":
        gen_shac(name="shac_hover_shared", share=True)
    else:          # separate interpreters: both patch module-level state of the imported reference
        import subprocess
        for n in ("shac_hover", "shac_hover_shared", "bptt_loop_hover"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--only", n])
