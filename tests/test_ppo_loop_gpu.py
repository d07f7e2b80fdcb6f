"""The PPO update LOOP against the reference's own ``PPO.train()`` (tests/golden/ppo_loop_nav*.npz, oracle/gen_ppo_loop.py): the reference's
train() (utils/algorithms/PPO.py:177-337) was run, unmodified, on its own CustomMultiInputActorCriticPolicy / StateTargetExtractor /
DictRolloutBuffer for 2 epochs x (3 full minibatches + a partial one); ``visfly_amd.ppo.PPO.train`` replays the same minibatches (the
recorded permutations) from the same rollout and initial parameters and must reproduce, per optimiser step, the loss, the value loss, the
flat gradient before clipping, its norm and the parameters after clip + Adam -- and per run the epoch / minibatch at which ``target_kl``
stops the loop, the number of optimiser steps and everything train() hands to its logger (equal-weight means over minibatches, the last
minibatch's loss, explained variance, std, n_updates = epochs, learning rate and clip ranges).  Third case: learning_rate / clip_range /
clip_range_vf are callables of SB3's progress_remaining, train() entered with 40 % of the run remaining.  Tolerances: those of the SHAC / BPTT loop fixtures (fp32 MFMA chains
vs MKL sgemm: gradients 2e-5 of the block scale, Adam 2e-7 absolute)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["ppo_loop_nav", "ppo_loop_nav_kl", "ppo_loop_nav_sched"]


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.mark.parametrize("name", CASES)
def test_fixture_is_a_run_of_the_references_train(name):
    """what the fixture must contain to pin the loop: both clip_grad_norm_ branches, the trailing partial minibatch, and (second case) an
    early stop inside the second epoch that costs an evaluated minibatch but no optimiser step"""
    fx = load(name)
    T, N, bs = int(fx["T"]), int(fx["N"]), int(fx["batch_size"])
    assert "the reference's own PPO.train" in str(fx["label"])
    assert fx["params0"].size == 43977 == fx["grad"].shape[1] == fx["params"].shape[1]
    per_epoch = [bs] * (T * N // bs) + ([T * N % bs] if T * N % bs else [])
    assert fx["mb_rows"].tolist() == (per_epoch * int(fx["n_epochs"]))[:len(fx["mb_rows"])]
    assert fx["perms"].shape == (int(fx["epochs_started"]), T * N) and all(sorted(p) == list(range(T * N)) for p in fx["perms"].tolist())
    n_opt = len(fx["loss"])
    if name == "ppo_loop_nav":
        assert fx["grad_norm"].min() < float(fx["max_grad_norm"]) < fx["grad_norm"].max(), "both branches of the gradient clip"
    if float(fx["target_kl"]) > 0:
        assert bool(fx["early_stop"]) and len(fx["value_loss"]) == n_opt + 1 and int(fx["epochs_started"]) == 2 and n_opt > len(per_epoch)
    else:
        assert not bool(fx["early_stop"]) and len(fx["value_loss"]) == n_opt == len(per_epoch) * int(fx["n_epochs"])


def _trainer_on_fixture(name):
    """the trainer with the fixture's hyper-parameters, initial parameters and rollout buffer -> (env, ppo, fx, perms, device tensor maker)"""
    import torch
    from visfly_amd.envs import NavigationEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    dev = "cuda:0"
    fx = load(name)
    T, N, bs = int(fx["T"]), int(fx["N"]), int(fx["batch_size"])
    opt = lambda k: None if float(fx[k]) < 0 else float(fx[k])
    lr_arg, clip_arg, clip_vf_arg = float(fx["lr"]), float(fx["clip_range"]), opt("clip_range_vf")
    if bool(fx["sched"]):        # the generator's schedules (oracle/gen_ppo_loop.py: sched_lr, sched_clip), passed as callables
        lr_arg, clip_arg, clip_vf_arg = (lambda p, v=lr_arg: v * p), (lambda p, v=clip_arg: v * (0.5 + 0.5 * p)), (lambda p, v=clip_vf_arg: v * (0.5 + 0.5 * p))
    env = NavigationEnv(num_agent_per_scene=N, seed=1, dynamics_kwargs=dict(ENV_DYN), device=dev, max_episode_steps=64, tensor_output=True)
    ppo = PPO(env, n_steps=T, batch_size=bs, n_epochs=int(fx["n_epochs"]), gamma=float(fx["gamma"]), gae_lambda=float(fx["gae_lambda"]),
              clip_range=clip_arg, ent_coef=float(fx["ent_coef"]), vf_coef=float(fx["vf_coef"]),
              max_grad_norm=float(fx["max_grad_norm"]), learning_rate=lr_arg, weight_decay=float(fx["weight_decay"]),
              adam_eps=float(fx["adam_eps"]), target_kl=opt("target_kl"), clip_range_vf=clip_vf_arg, seed=0,
              policy_kwargs=dict(activation_fn="relu"))        # the fixture's networks (oracle/gen_ppo_loop.py: activation_fn=nn.ReLU in both places)
    ppo._current_progress_remaining = float(fx["progress_remaining"])          # what learn() sets before train() (PPO.py:150-152)
    pol = ppo.policy
    n = pol.n_params
    assert n == fx["params0"].size
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    pol.flat[:n].copy_(d(fx["params0"]))
    pol.mark_updated()
    b = ppo.buf
    b.obs["state"].copy_(d(fx["obs_state"]))
    b.obs["target"].copy_(d(fx["obs_target"]))
    b.actions.copy_(d(fx["actions"]))
    for dst, key in ((b.values, "values"), (b.log_probs, "log_probs"), (b.advantages, "advantages"), (b.returns, "returns")):
        dst.copy_(d(fx[key]))
    # the reference's rows are env-major (SB3 swap_and_flatten: row = env * T + step), the trainer's step-major
    perms = [(p % T) * N + (p // T) for p in fx["perms"].astype(np.int64)]
    return env, ppo, fx, perms, d


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_train_replays_the_references_train(name):
    import torch
    env, ppo, fx, perms, d = _trainer_on_fixture(name)
    pol = ppo.policy
    n = pol.n_params
    rec = {"loss": [], "value_loss": [], "grad": [], "params": [], "stepped": []}
    real = ppo._minibatch_update

    def spy(mb, stats_acc=None):
        rows = mb["adv"].numel()
        r = real(mb, stats_acc)
        s = ppo._stats[:9].double().cpu().numpy() / rows
        # PPO.py:257-261 with entropy_loss = -mean(-log_prob) (the squashed Gaussian has no analytical entropy, :249-251)
        rec["loss"].append(s[0] + float(fx["ent_coef"]) * s[2] + float(fx["vf_coef"]) * s[1])
        rec["value_loss"].append(s[1])
        rec["stepped"].append(r is not None)
        if r is not None:
            rec["grad"].append(pol.grad[:n].cpu().numpy().copy())
            rec["params"].append(pol.flat[:n].cpu().numpy().copy())
        return r
    ppo._minibatch_update = spy
    ppo.train(permutations=perms)
    torch.cuda.synchronize()

    n_opt = len(fx["loss"])
    assert rec["stepped"] == [True] * n_opt + [False] * (len(fx["value_loss"]) - n_opt), "same minibatches, same early stop"
    assert ppo._opt_step == n_opt and bool(ppo.logs["train/early_stop"]) == bool(fx["early_stop"])
    assert np.allclose(rec["value_loss"], fx["value_loss"], rtol=2e-5, atol=1e-7), (rec["value_loss"], fx["value_loss"])
    assert np.allclose(rec["loss"][:n_opt], fx["loss"], rtol=0, atol=2e-6 * max(1.0, np.abs(fx["loss"]).max())), (rec["loss"], fx["loss"])
    # per-layer blocks of the flat gradient, each against its own scale
    offs = [(ly.w_off, ly.w_off + ly.K * ly.No) for ly in pol.layers] + [(ly.b_off, ly.b_off + ly.No) for ly in pol.layers] + [(pol.log_std_off, n)]
    for i in range(n_opt):
        g, w = rec["grad"][i], fx["grad"][i]
        for lo, hi in offs:
            scale = max(np.abs(w[lo:hi]).max(), 1e-3 * np.abs(w).max())
            # (from the second step on the two runs no longer hold the same parameters to the last bit: the bound grows with the step)
            assert np.abs(g[lo:hi] - w[lo:hi]).max() <= 2e-5 * (i + 1) * scale, f"step {i}: gradient block [{lo}, {hi})"
        assert abs(np.linalg.norm(g.astype(np.float64)) - fx["grad_norm"][i]) <= 2e-5 * (i + 1) * fx["grad_norm"][i]
    # clip + Adam (torch.optim.Adam with weight_decay = L2 term in the gradient, PPO.py:284-292).  An Adam step is lr * m^ / (sqrt(v^) + eps):
    # for the handful of parameters whose gradient is ~1e-8 (units active in a row or two) a 1e-10 difference of the gradient is a
    # visible fraction of lr, whatever computes it -- so (a) the trainer's Adam is checked tightly against a float64 replay from the
    # trainer's OWN gradients, (b) against the reference's parameters the bulk must agree to the SHAC loop's 2e-7 per step and no
    # parameter may be off by more than 2 % of a full-size step
    lr, wd, eps, b1, b2 = float(fx["log_learning_rate"]), float(fx["weight_decay"]), float(fx["adam_eps"]), 0.9, 0.999
    p, m, v = fx["params0"].astype(np.float64), np.zeros(n), np.zeros(n)
    for i in range(n_opt):
        g = rec["grad"][i].astype(np.float64)
        g = g * min(1.0, float(fx["max_grad_norm"]) / (np.linalg.norm(g) + 1e-6)) + wd * p
        m, v = b1 * m + (1 - b1) * g, b2 * v + (1 - b2) * g * g
        p = p - lr / (1 - b1 ** (i + 1)) * m / (np.sqrt(v) / np.sqrt(1 - b2 ** (i + 1)) + eps)
        assert np.abs(rec["params"][i] - p).max() <= 3e-7 * (i + 1), f"Adam step {i} vs a float64 replay of the trainer's own gradients"
        diff = np.abs(rec["params"][i] - fx["params"][i])
        assert np.quantile(diff, 0.99) <= 2e-7 * (i + 1) and diff.max() <= 0.02 * lr * (i + 1), (i, np.quantile(diff, 0.99), diff.max())
    # what train() logs (PPO.py:322-336): equal-weight means over the evaluated minibatches (approx_kl: those of the last epoch started),
    # the last minibatch's loss, explained variance of the buffer, std, n_updates (+1 per epoch), the schedules' current values
    lg = ppo.logs
    for key, name, tol in (("train/value_loss", "log_value_loss", 2e-5), ("train/policy_gradient_loss", "log_policy_gradient_loss", 2e-5),
                           ("train/entropy_loss", "log_entropy_loss", 2e-6), ("train/approx_kl", "log_approx_kl", 2e-5),
                           ("train/clip_fraction", "log_clip_fraction", 1e-9), ("train/loss", "log_loss", 2e-5),
                           ("train/explained_variance", "log_explained_variance", 1e-5), ("train/std", "log_std_mean", 1e-6),
                           ("train/learning_rate", "log_learning_rate", 1e-12), ("train/clip_range", "log_clip_range", 1e-9)):
        assert abs(lg[key] - float(fx[name])) <= tol * max(1.0, abs(float(fx[name]))), (key, lg[key], float(fx[name]))
    assert lg["train/n_updates"] == int(fx["log_n_updates"]) == int(fx["n_updates"])
    if float(fx["clip_range_vf"]) > 0:
        assert abs(lg["train/clip_range_vf"] - float(fx["log_clip_range_vf"])) <= 1e-9
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_every_optimiser_step_from_the_references_own_state(name):
    """the replay above lets the two runs drift apart (its bounds grow with the step index: r05 verdict).  Here every optimiser step starts
    from the REFERENCE's state -- its parameters after the previous step and the Adam moments a float64 replay of its recorded gradients
    gives -- so that each step is held to the bounds of the first one: gradient 2e-5 of the block scale, norm 2e-5, parameters after clip +
    Adam 99 % within 2e-7 and none off by more than 2 % of a full-size step"""
    import torch
    env, ppo, fx, perms, d = _trainer_on_fixture(name)
    pol = ppo.policy
    n = pol.n_params
    lr, wd, eps, b1, b2, mx = float(fx["log_learning_rate"]), float(fx["weight_decay"]), float(fx["adam_eps"]), 0.9, 0.999, float(fx["max_grad_norm"])
    n_opt = len(fx["loss"])
    # the reference's optimiser state before step k, from its own gradients (float64)
    m, v, states = np.zeros(n), np.zeros(n), []
    for k in range(n_opt):
        states.append((m.copy(), v.copy()))
        p_before = (fx["params0"] if k == 0 else fx["params"][k - 1]).astype(np.float64)
        g = fx["grad"][k].astype(np.float64)
        g = g * min(1.0, mx / (np.linalg.norm(g) + 1e-6)) + wd * p_before
        m, v = b1 * m + (1 - b1) * g, b2 * v + (1 - b2) * g * g
    rec = {"grad": [], "params": []}
    real = ppo._minibatch_update
    offs = [(ly.w_off, ly.w_off + ly.K * ly.No) for ly in pol.layers] + [(ly.b_off, ly.b_off + ly.No) for ly in pol.layers] + [(pol.log_std_off, n)]

    def spy(mb, stats_acc=None):
        k = len(rec["grad"])
        if 0 < k < n_opt:          # continue from where the REFERENCE stood
            pol.flat[:n].copy_(d(fx["params"][k - 1]))
            pol.mark_updated()
            ppo.exp_avg[:n].copy_(d(states[k][0].astype(np.float32)))
            ppo.exp_avg_sq[:n].copy_(d(states[k][1].astype(np.float32)))
        r = real(mb, stats_acc)
        if r is not None:
            rec["grad"].append(pol.grad[:n].cpu().numpy().copy())
            rec["params"].append(pol.flat[:n].cpu().numpy().copy())
        return r
    ppo._minibatch_update = spy
    ppo.train(permutations=perms)
    torch.cuda.synchronize()
    assert len(rec["grad"]) == n_opt
    for i in range(n_opt):
        g, w = rec["grad"][i], fx["grad"][i]
        for lo, hi in offs:
            scale = max(np.abs(w[lo:hi]).max(), 1e-3 * np.abs(w).max())
            assert np.abs(g[lo:hi] - w[lo:hi]).max() <= 2e-5 * scale, f"step {i}: gradient block [{lo}, {hi})"
        assert abs(np.linalg.norm(g.astype(np.float64)) - fx["grad_norm"][i]) <= 2e-5 * fx["grad_norm"][i]
        diff = np.abs(rec["params"][i] - fx["params"][i])
        assert np.quantile(diff, 0.99) <= 2e-7 and diff.max() <= 0.02 * lr, (i, np.quantile(diff, 0.99), diff.max())
    env.close()
