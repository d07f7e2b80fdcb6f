"""SHAC: short-horizon actor-critic through the differentiable simulator.

Restates ``TemporalDifferBase.learn`` of the reference (utils/algorithms/shac.py:185-326) with the reference's network shapes
(utils/policies/td_policies.py):

* **Actor** (:146-252): features extractor -> two trunks of identical shape, ``latent_pi -> mu`` (4) and ``log_latent_pi ->
  log_std`` (4, clamped to [-10, 2], :31-32,241-243); action = tanh(mu + eps * exp(log_std)) (SB3's squashed Gaussian).  One
  ``MlpPolicy`` layer table with two 4-wide heads; ``log_latent_pi`` starts as a copy of ``latent_pi`` (:205).
* **Critic** (:82-143): its OWN features extractor (``share_features_extractor=False``, the reference's default: trained by the critic loss;
  with ``True`` the critic runs the actor's extractor under no_grad and its optimiser leaves it alone -- r06, fixture ``shac_hover_shared``),
  ``th.cat([features, actions])`` -> two Q MLPs ``qf0`` / ``qf1`` -> 1.  One layer table: extractor branch + a frozen identity
  branch that appends the 4 action columns to the features, two trunks with 1-wide heads; a second instance is the target.

Per iteration (shac.py:215-278):

1. H env steps with the stochastic actor; ``actor_loss = mean_i sum_t [-reward_t disc_t - next_value_t disc_t gamma cut_t]``,
   ``cut = (done | t == H-1) & ~episode_done``, ``next_value = min(Q1', Q2')`` of the TARGET critic on the detached next
   observation and a freshly sampled, detached next action (:242-257) -- a constant of the actor objective, so the actor
   gradient is the BPTT reverse sweep (adjoint env kernel ``vf_env_step_bwd`` + MLP backward) while the loss VALUE carries the
   bootstrap term (``vf_shac_accumulate``).
2. clip_grad_norm_(0.5) + Adam on the actor (:271-276 of the actor part), ``env.detach()``.
3. TD-lambda returns of the horizon buffer (``vf_td_returns``; the buffer passes the literal gamma 0.99,
   utils/algorithms/common.py:1232-1239).
4. ``gradient_steps`` x: ``mse_loss(returns, min(Q1, Q2))`` over the whole buffer (``vf_twin_q_loss``), backward through both Q
   trunks and the critic's extractor, clip 0.5, Adam, Polyak update of the target (``vf_polyak_update``).

Parity: pinned by ``tests/golden/shac_hover.npz`` -- ONE iteration of the reference's own ``learn()`` loop with its own Actor /
ContinuousCritic / StateExtractor / create_mlp / SimpleRolloutBuffer / compute_td_returns over the differentiable HoverEnv
(``oracle/gen_shac.py``; SB3's base classes restated there, reference defect C-10 repaired: the buffer's observations are
flattened like its actions): horizon buffer, next values, actor loss incl. bootstrap, flat actor gradient, actor parameters after
the step, returns, and per critic step loss / gradient / parameters / target parameters (``tests/test_shac_gpu.py``).
"""
import os
import ctypes as C
from typing import Optional

import torch as th

from . import _lib, checkpoint, parallel
from .bptt import BPTT, LOG_STD_MAX, LOG_STD_MIN
from .ppo import MlpPolicy, _ptr


class SHAC(BPTT):
    def __init__(self, env, policy=None, policy_kwargs: Optional[dict] = None, learning_rate: float = 1e-3, logger_kwargs=None,
                 comment=None, save_path=None, dump_step=1e4, horizon: int = 32, tau: float = 0.005, gamma: float = 0.99,
                 gradient_steps: int = 5, buffer_size: int = int(1e6), batch_size: int = int(2e5), clip_range_vf: float = 0.1,
                 pre_stop: float = 0.1, policy_noise: float = 0.0, device=None, seed: int = 42, lamda: float = 0.95,
                 max_grad_norm: float = 0.5, **kw):
        # argument list of TemporalDifferBase.__init__ (shac.py:50-72); buffer_size / batch_size / clip_range_vf / pre_stop /
        # policy_noise are stored but unused by the reference's loop as well
        if policy not in (None, "MultiInputPolicy", "MlpPolicy", "MTDPolicy"):
            raise NotImplementedError(f"policy {policy}: vector-observation MTDPolicy only")
        self._critic_arch = None
        self.tau, self.gradient_steps, self.lamda = tau, gradient_steps, lamda
        self.fused_critic = os.environ.get("VISFLY_AMD_FUSED_CRITIC", "1") != "0"   # a critic update's per-row part as one launch (vf_twin_q_update); A/B
        self.batch_targets = os.environ.get("VISFLY_AMD_BATCH_TARGETS", "1") != "0"  # the horizon's target-critic passes as one launch (vf_mlp_forward_steps); A/B
        kw.pop("policy", None)
        super().__init__(env, horizon=horizon, gamma=gamma, learning_rate=learning_rate, max_grad_norm=max_grad_norm,
                         policy_kwargs=policy_kwargs, seed=seed, **kw)
        pol, dev = self.policy, self.device
        obs_dims = {k: pol.obs_dims[k] for k in self._ext_keys}
        mk = lambda s: MlpPolicy({**obs_dims, "action": 4}, self._extractor, self._critic_arch, self._critic_arch, dev, seed=s,
                                 ortho_init=False, head_dims=(1, 1), passthrough=("action",), log_std_param=False)
        self.critic, self.critic_target = mk(seed + 101), mk(seed + 101)
        parallel.broadcast_(self.critic.flat)
        # share_features_extractor=True (td_policies.py:127; SB3 SACPolicy._build): critic.features_extractor IS the actor's -- run under
        # no_grad, left alone by the critic's optimiser; the TARGET keeps its own copy (initialised from it, Polyak-averaged with the rest).
        # Here: the critic's extractor block of `flat` mirrors the actor's (same layer order and offsets: extractor layers lead both tables)
        self._share_extractor = bool(getattr(self, "_share_extractor", False))
        self._ext_end = sum(ly.K * ly.No + ly.No for ly in self._ext_layers()) if self._share_extractor else 0
        self._stale_ext = th.zeros(self._ext_end, device=dev) if self._share_extractor else None
        if self._share_extractor:
            assert [(ly.K, ly.No, ly.w_off) for ly in self.critic.layers[:len(self._ext_layers())]] == \
                   [(ly.K, ly.No, ly.w_off) for ly in pol.layers[:len(self._ext_layers())]], "extractor blocks of actor and critic differ"
            self._sync_shared_extractor()
        self.critic_target.flat.copy_(self.critic.flat)                              # critic_target.load_state_dict(critic.state_dict())
        for net in (self.critic, self.critic_target):
            net.lazy_pack = True
            net.mark_updated()
        n = self.critic.n_params
        self.c_exp_avg, self.c_exp_avg_sq = th.zeros(n, device=dev), th.zeros(n, device=dev)
        self._c_sumsq = th.zeros(1, device=dev)
        self._critic_step = 0
        self._buf = None
        # _eps_override (tests): (2H, N, 4) noise feed, [2t] = action of step t, [2t+1] = next action

    def _ext_layers(self):
        return [ly for ly in self.critic.layers if not ly.frozen and (ly.dst == "feat" or ly.dst.startswith("x:"))]

    def _sync_shared_extractor(self):
        """share_features_extractor: the critic's extractor parameters ARE the actor's (after every actor step)"""
        self.critic.flat[:self._ext_end].copy_(self.policy.flat[:self._ext_end])
        self.critic.mark_updated()

    def _after_actor_step(self):
        """share_features_extractor: the shared extractor's .grad still holds the actor loss's gradient, scaled by the actor's clip
        (clip_grad_norm_ works in place; the critic's optimizer.zero_grad() does not touch parameters it does not own), and
        clip_grad_norm_(critic.parameters()) (shac.py:274) counts it; the critic sees the actor's NEW extractor"""
        if self._share_extractor:
            coef = th.clamp(self.max_grad_norm / (self._sumsq.sqrt() + 1e-6), max=1.0)
            self._stale_ext.copy_(self.policy.grad[:self._ext_end] * coef)             # (policy.grad: summed over the ranks by _apply)
            self._sync_shared_extractor()

    # ---- networks ---------------------------------------------------------------------------------------------------------
    # the Actor (extractor -> latent_pi -> mu | log_latent_pi -> log_std) and its action head are BPTT's reference-actor path
    # (bptt.py::_make_reference_actor / _head_fwd): the reference's BPTT and SHAC share td_policies.Actor
    reference_actor = True

    def _q(self, net, obs, action, save=False, slot=0):
        q0, q1 = net.forward({**{k: obs[k] for k in self._ext_keys}, "action": action}, save_activations=save, slot=slot)
        return q0.view(-1), q1.view(-1)

    # ---- actor: horizon roll-out + reverse sweep ------------------------------------------------------------------------------
    def _grad_reverse_sweep(self):
        """shac.py:215-266 forward, then the reverse sweep t = H-1 .. 0: adjoint env step -> action head -> both actor trunks ->
        gradient w.r.t. the observation of step t, which step t-1 returned.  Like BPTT's reference-actor sweep the activations of
        every step stay in their slot and the actor's weight gradient is reduced once over the H N rows of the horizon.  Where the
        library has the kernels (vf_bptt_rollout / vf_bptt_reverse, actor class (b)) the closed loop policy -> env -> policy and its
        reverse are ONE persistent launch each; the terms of the loss VALUE that are constants of the actor objective -- next action
        and target critics on the detached next observation (:234-239) -- follow over the recorded horizon: the next observation of
        step t IS the input of slot t + 1, so its actor heads are already there (same weights), only the last step's are computed.
        Leaves what the launch-by-launch loop leaves, bit for bit (tests/test_shac_gpu.py)."""
        env, pol, N, H = self.env, self.policy, self.env.num_envs, self.H
        L, st, dev = _lib.lib(), _lib.current_stream(self.device), self.device
        keys = self.obs_keys
        pol.grad.zero_()
        pol.reserve_slots(N, H)
        if self._defer_wgrad is None:
            self._defer_wgrad = pol.backward_data_supported(N, both_heads=True)
        defer = self._defer_wgrad
        blk = pol._slot_blocks[N][1]
        disc, loss_vec = th.ones(N, device=dev), th.zeros(N, device=dev)
        f = dict(dtype=th.float32, device=dev)
        u8 = dict(dtype=th.uint8, device=dev)
        b = self._buf = dict(obs={k: th.empty((H, N, pol.obs_dims[k]), **f) for k in keys}, action=th.empty((H, N, 4), **f),
                             reward=th.empty((H, N), **f), done=th.empty((H, N), **u8), ep_done=th.empty((H, N), **u8),
                             next_value=th.empty((H, N), **f))
        eps = self._eps_override if self._eps_override is not None else th.randn((2 * H, N, 4), device=dev, generator=self._gen)
        assert eps.shape == (2 * H, N, 4)
        eps_a = eps[0::2].contiguous()                                                 # [t] = noise of the action of step t
        drews, ls_rows = th.empty((H, N), **f), blk["value"]
        scale = 1.0 / (N * self.world)
        t0 = env._tape_t
        fused = False
        if defer and self.fused_rollout:
            flag_rows = th.empty((H, N), **u8)
            fused = env.rollout_policy(pol, keys, eps_a, b["action"], drews, th.zeros(N, **f), th.ones(N, **f), float(self.gamma), scale,
                                       reward_rows=b["reward"], ep_flag_rows=flag_rows)
        if fused:
            for k in keys:
                b["obs"][k].copy_(blk["obs:" + k][:H])                                 # rollout_buffer.add(obs=pre_obs ...) :259
            b["done"].copy_(env._tape_done[t0:t0 + H])
            last = env.get_observation()
            o_last = {k: last[k].detach().contiguous() for k in keys}
            mu2, ls2, nxt = th.empty((H, N, 4), **f), th.empty((H, N, 4), **f), th.empty((H, N, 4), **f)
            if H > 1:
                mu2[:H - 1].copy_(blk["mean"][1:H])
                ls2[:H - 1].copy_(blk["value"][1:H])
            m, l = pol.forward(o_last, save_activations=False, slot=H)
            mu2[H - 1].copy_(m)
            ls2[H - 1].copy_(l)
            self._head_fwd(mu2.view(-1, 4), ls2.view(-1, 4), eps[1::2].contiguous().view(-1, 4), nxt.view(-1, 4))
            # target critics on (obs', a') of every step and the loss / discount recurrence: ONE forward launch over the H N rows with the
            # rows-per-wave choice of an N-row launch (vf_mlp_forward_steps: every row as the per-step launch computes it) + ONE
            # accumulate launch, where the library has them; else H launches of each
            qs = None
            if self.batch_targets:
                if getattr(self, "_obs2", None) is None or self._obs2[keys[0]].shape[:2] != (H, N):
                    self._obs2 = {k: th.empty((H, N, pol.obs_dims[k]), **f) for k in keys}
                for k in keys:
                    if H > 1:
                        self._obs2[k][:H - 1].copy_(blk["obs:" + k][1:H])
                    self._obs2[k][H - 1].copy_(o_last[k])
                tg = self.critic_target
                qs = tg.forward_steps({**{k: self._obs2[k].view(H * N, -1) for k in self._ext_keys}, "action": nxt.view(H * N, 4)}, N, H)
            if qs is not None:
                _lib.check(L.vf_shac_accumulate_horizon(_ptr(b["reward"]), b["done"].data_ptr(), flag_rows.data_ptr(), _ptr(qs[0]), _ptr(qs[1]),
                                                        _ptr(disc), _ptr(loss_vec), _ptr(drews), _ptr(b["next_value"]),
                                                        b["ep_done"].data_ptr(), float(self.gamma), scale, H, N, st))
            for t in range(H if qs is not None else 0, H):      # target critics per step: N rows per launch, the row count (-> kernel choice) of the loop
                o2 = {k: blk["obs:" + k][t + 1] for k in keys} if t + 1 < H else o_last
                q0, q1 = self._q(self.critic_target, o2, nxt[t], slot=0)
                _lib.check(L.vf_shac_accumulate(_ptr(b["reward"][t]), b["done"][t].data_ptr(), flag_rows[t].data_ptr(), _ptr(q0), _ptr(q1),
                                                _ptr(disc), _ptr(loss_vec), _ptr(drews[t]), _ptr(b["next_value"][t]),
                                                b["ep_done"][t].data_ptr(), float(self.gamma), scale, 1 if t == H - 1 else 0, N, st))
        else:
            nxt = th.empty((N, 4), **f)
            obs = env.get_observation()
        for t in range(0 if not fused else H, H):
            for k in keys:
                b["obs"][k][t].copy_(obs[k].detach())                                  # rollout_buffer.add(obs=pre_obs ...) :259
            o = {k: b["obs"][k][t] for k in keys}                                      # rows that outlive the reverse sweep
            mu, ls = pol.forward(o, slot=t)                                            # actor.action_log_prob(obs) :219
            action = b["action"][t]
            self._head_fwd(mu, ls, eps_a[t], action)
            obs, reward, done, _ = env._step_no_grad(action, False, record=True, borrow=True)      # :225
            # next action of the stochastic actor on the new observation, target critics on (obs', a'), all detached :234-239
            o2 = {k: obs[k].detach().contiguous() for k in keys}
            mu2, ls2 = pol.forward(o2, save_activations=False, slot=H)
            self._head_fwd(mu2, ls2, eps[2 * t + 1], nxt)
            q0, q1 = self._q(self.critic_target, o2, nxt, slot=0)
            b["reward"][t].copy_(reward)
            b["done"][t].copy_(done)
            _lib.check(L.vf_shac_accumulate(_ptr(reward), done.data_ptr(), env._ep_flags.data_ptr(), _ptr(q0), _ptr(q1), _ptr(disc),
                                            _ptr(loss_vec), _ptr(drews[t]), _ptr(b["next_value"][t]), b["ep_done"][t].data_ptr(),
                                            float(self.gamma), scale, 1 if t == H - 1 else 0, N, st))
        g_obs = None
        d_mus, d_lss = th.empty((H, N, 4), **f), th.empty((H, N, 4), **f)
        rev = False
        if fused and self.fused_reverse:
            rev = env.reverse_policy(pol, H, eps_a, b["action"], drews, d_mus, None, d_log_stds=d_lss)
        for t in reversed(range(0 if not rev else H, H)):
            d_action = env.backward_step(t0 + t, g_obs, drews[t])
            _lib.check(L.vf_shac_head_bwd(_ptr(d_action), _ptr(b["action"][t]), _ptr(ls_rows[t]), _ptr(eps_a[t]), _ptr(d_mus[t]),
                                          _ptr(d_lss[t]), N, LOG_STD_MIN, LOG_STD_MAX, st))
            if defer:
                d_in = pol.backward_data(d_mus[t], t, d_value=d_lss[t])
            else:
                d_in = pol.backward(d_mus[t], d_lss[t], None, accumulate=True, need_input_grad=t > 0, slot=t)
            g_obs = d_in.get("state") if t > 0 else None
        if defer:
            pol.weight_grad_slots(N, H, d_mus, accumulate=True, d_value_all=d_lss)
        return loss_vec.mean() / self.world

    # ---- one iteration ------------------------------------------------------------------------------------------------------
    def _update(self):
        loss = self._grad_reverse_sweep()
        out = self._apply(loss)                                                        # clip 0.5 + Adam + env.detach() :271-277
        self._after_actor_step()
        b, N, H = self._buf, self.env.num_envs, self.H
        L, st = _lib.lib(), _lib.current_stream(self.device)
        returns = th.empty((H, N), device=self.device)
        # SimpleRolloutBuffer.compute_returns passes the literal 0.99 (common.py:1232-1239)
        _lib.check(L.vf_td_returns(_ptr(b["reward"]), b["done"].data_ptr(), b["ep_done"].data_ptr(), _ptr(b["next_value"]),
                                   _ptr(returns), H, N, 0.99, float(self.lamda), st))
        b["returns"] = returns
        obs = {k: v.view(H * N, -1) for k, v in b["obs"].items()}
        self._last_losses = (out, self._train_critics(obs, b["action"].view(H * N, 4), returns.view(-1)))
        return out

    def flush_logs(self):
        """device scalars of the last iteration -> self.logs (one host sync; the loop itself never waits for the GPU)"""
        if getattr(self, "_last_losses", None) is not None:
            a, c = self._last_losses
            c = c.clone()
            parallel.allreduce_sum_(c)
            self.logs["train/actor_loss"], self.logs["train/critic_loss"] = float(a), float(c)
        return self.logs

    def learn(self, total_timesteps: int, log_interval: Optional[int] = None):
        out = super().learn(total_timesteps, log_interval)
        self.flush_logs()
        return out

    def _critic_step_once(self, obs, action, target):
        """one critic update (shac.py:267-274) -> loss (0-dim device tensor, this rank's share of the global mean)"""
        L, st, M = _lib.lib(), _lib.current_stream(self.device), action.shape[0]
        c, dev = self.critic, self.device
        if getattr(self, "_dq", None) is None or self._dq[0].numel() != M:
            self._dq = (th.empty(M, device=dev), th.empty(M, device=dev), th.empty(1, device=dev),
                        th.empty(int(L.vf_twin_q_loss_scratch_doubles(M)), dtype=th.float64, device=dev))
        dq0, dq1, loss, scr = self._dq
        # forward + loss + reverse chain of the update as ONE launch where the library has the class (vf_twin_q_update), else three
        if not (self.fused_critic and c.twin_q_update({**{k: obs[k] for k in self._ext_keys}, "action": action}, target, loss,
                                                      M * self.world)):
            q0, q1 = self._q(c, obs, action, save=True)
            _lib.check(L.vf_twin_q_loss(_ptr(q0), _ptr(q1), _ptr(target), _ptr(dq0), _ptr(dq1), _ptr(loss), scr.data_ptr(), M,
                                        M * self.world, st))
            c.backward(dq0.view(M, 1), dq1.view(M, 1), None)
        parallel.allreduce_sum_(c.grad)
        if getattr(self, "_critic_grad_override", None) is not None:      # tests: the optimiser trajectory from recorded gradients
            c.grad.copy_(self._critic_grad_override)
        if self._share_extractor:          # no gradient reaches the shared extractor from the critic loss; what its .grad holds is the actor's (above)
            c.grad[:self._ext_end].copy_(self._stale_ext)
        self._last_critic_grad = c.grad
        _lib.check(L.vf_sumsq(_ptr(c.grad), c.n_params, _ptr(self._c_sumsq), _ptr(self._scratch), st))
        self._critic_step += 1
        pmap, packed = c.pack_map()
        cfg = _lib.AdamCfg(self.lr, self.betas[0], self.betas[1], self.adam_eps, 0.0, self.max_grad_norm, self._critic_step, 0,
                           _ptr(pmap), _ptr(packed))
        _lib.check(L.vf_adam_step(_ptr(c.flat), _ptr(c.grad), _ptr(self.c_exp_avg), _ptr(self.c_exp_avg_sq), c.n_params,
                                  _ptr(self._c_sumsq), C.byref(cfg), st))
        c.mark_updated(packed_current=pmap is not None)
        if self._share_extractor:
            # the critic's optimiser does not own the extractor: its parameters stay the actor's, its moments never existed; the clip
            # rescaled the stale gradient in place
            e = self._ext_end
            self._stale_ext.mul_(th.clamp(self.max_grad_norm / (self._c_sumsq.sqrt() + 1e-6), max=1.0))
            self.c_exp_avg[:e].zero_()
            self.c_exp_avg_sq[:e].zero_()
            self._sync_shared_extractor()
        tg = self.critic_target
        _lib.check(L.vf_polyak_update(_ptr(tg.flat), _ptr(c.flat), c.n_params, float(self.tau), st))    # trainable part only
        tg.mark_updated()
        return loss[0]

    def _train_critics(self, obs, action, target):
        loss = th.full((), float("nan"), device=self.device)
        for _ in range(self.gradient_steps):
            loss = self._critic_step_once(obs, action, target)
        return loss

    # ---- inference -------------------------------------------------------------------------------------------------------------
    def predict(self, obs, state=None, episode_start=None, deterministic: bool = False):
        """shac.py:334-343 / MTDPolicy.predict: -> (action, None); deterministic: tanh(mu)"""
        N = obs[self.obs_keys[0]].shape[0]
        mu, ls = self.policy.forward({k: obs[k].detach().contiguous() for k in self.obs_keys}, save_activations=False, slot=self.H + 1)
        if deterministic:
            return th.tanh(mu), None
        eps = th.randn((N, 4), device=self.device, generator=self._gen)
        action = th.empty((N, 4), device=self.device)
        self._head_fwd(mu, ls.contiguous(), eps, action)
        return action, None

    def _state(self):
        d = super()._state()
        d.update({"critic": self.critic.flat.cpu(), "critic_target": self.critic_target.flat.cpu(), "c_exp_avg": self.c_exp_avg.cpu(),
                  "c_exp_avg_sq": self.c_exp_avg_sq.cpu(), "critic_step": int(self._critic_step)})
        d["spec"].update(tau=self.tau, gradient_steps=self.gradient_steps, lamda=self.lamda)
        return d

    def _load_state(self, d, load_optimizer=True):
        super()._load_state(d, load_optimizer)
        for net, key in ((self.critic, "critic"), (self.critic_target, "critic_target")):
            assert d[key].numel() == net.flat.numel(), f"{key}: the archive holds a different network"
            net.flat.copy_(d[key])
            net.mark_updated()
        if load_optimizer and "c_exp_avg" in d:
            self.c_exp_avg.copy_(d["c_exp_avg"])
            self.c_exp_avg_sq.copy_(d["c_exp_avg_sq"])
            self._critic_step = int(d["critic_step"])
