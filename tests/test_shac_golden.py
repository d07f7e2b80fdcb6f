"""CPU side of the SHAC golden (tests/golden/shac_hover.npz from the reference's own learn() loop, oracle/gen_shac.py):
the oracle's TD-lambda is bit-identical to the buffer's returns, the actor-loss recurrence incl. the bootstrap term
(shac.py:247-257) and the twin-Q objective (:267-270) restated in a few lines of numpy / torch reproduce the recorded
scalars and the recorded critic gradient from the recorded parameters -- i.e. the fixture's parameter layout and every
quantity the GPU test (tests/test_shac_gpu.py) compares against mean what that test assumes."""
import numpy as np
import torch

import oracle
from _golden import load


def test_oracle_td_lambda_is_the_buffers_returns():
    fx = load("shac_hover")
    ret = oracle.td_returns(fx["buf_reward"], fx["buf_done"], fx["buf_next_value"], fx["buf_episode_done"], 0.99, float(fx["lamda"]))
    assert np.array_equal(ret.view(np.uint32), fx["buf_returns"].view(np.uint32))
    assert fx["buf_done"].sum() > fx["buf_episode_done"].sum() > 0            # truncations AND true episode ends in the horizon


def test_actor_loss_recurrence_with_bootstrap():
    fx = load("shac_hover")
    H, N = fx["buf_reward"].shape
    g = np.float32(float(fx["gamma"]))
    loss, disc = np.zeros(N, np.float32), np.ones(N, np.float32)
    for t in range(H):
        done, epd = fx["buf_done"][t].astype(bool), fx["buf_episode_done"][t].astype(bool)
        loss = loss - fx["buf_reward"][t] * disc
        cut = (done | (t == H - 1)) & ~epd
        loss = loss - fx["buf_next_value"][t] * disc * g * cut.astype(np.float32)
        disc = disc * g * (~done).astype(np.float32) + done.astype(np.float32)
    assert abs(float(loss.mean()) - float(fx["actor_loss"])) <= 1e-6
    # without the bootstrap term the value is a different one: the recorded scalar DOES include it
    plain = -(fx["buf_reward"] * 1.0).sum(0).mean()
    assert abs(float(plain) - float(fx["actor_loss"])) > 1e-3


def _critic(params, obs, act):
    """extractor 13 -> 128 -> 64, cat action, two Q MLPs 68 -> 64 -> 64 -> 1; layout: per layer weight [No][K] then bias"""
    p, off = torch.from_numpy(params).double().requires_grad_(True), 0

    def lin(x, K, No, relu=True):
        nonlocal off
        w, b = p[off:off + K * No].view(No, K), p[off + K * No:off + K * No + No]
        off += K * No + No
        y = x @ w.T + b
        return torch.relu(y) if relu else y

    f = lin(lin(obs, 13, 128), 128, 64)
    x = torch.cat([f, act], dim=1)
    qs = []
    for _ in range(2):
        qs.append(lin(lin(lin(x, 68, 64), 64, 64), 64, 1, relu=False))
    assert off == params.size
    return p, qs


def test_twin_q_objective_and_gradient_from_recorded_parameters():
    fx = load("shac_hover")
    H, N = fx["buf_reward"].shape
    obs = torch.from_numpy(fx["buf_obs"].reshape(H * N, 13)).double()
    act = torch.from_numpy(fx["buf_action"].reshape(H * N, 4)).double()
    target = torch.from_numpy(fx["buf_returns"].reshape(-1)).double()
    for i in range(int(fx["gradient_steps"])):
        params = fx["critic_params0"] if i == 0 else fx["critic_params"][i - 1]
        p, (q0, q1) = _critic(params, obs, act)
        values = torch.cat([q0, q1], dim=1).min(dim=1)[0]
        loss = torch.nn.functional.mse_loss(target, values)
        assert abs(float(loss.detach()) - float(fx["critic_loss"][i])) <= 2e-6
        loss.backward()
        want = fx["critic_grad"][i]
        assert np.abs(p.grad.numpy() - want).max() <= 2e-5 * np.abs(want).max()
    # Polyak: target_i = (1 - tau) target_{i-1} + tau critic_i
    tau = float(fx["tau"])
    t = fx["critic_params0"].astype(np.float64)
    for i in range(int(fx["gradient_steps"])):
        t = (1 - tau) * t + tau * fx["critic_params"][i]
        assert np.abs(t - fx["target_params"][i]).max() <= 2e-7


def test_adam_step_of_the_actor_from_the_recorded_gradient():
    fx = load("shac_hover")
    g = fx["actor_grad"].astype(np.float64)
    norm = np.linalg.norm(g)
    g = g * min(1.0, 0.5 / (norm + 1e-6))                                   # clip_grad_norm_(0.5)
    m, v = 0.1 * g, 0.001 * g * g
    step = float(fx["lr"]) * (m / 0.1) / (np.sqrt(v / 0.001) + 1e-8)
    assert np.abs(fx["actor_params0"] - step - fx["actor_params1"]).max() <= 2e-7


# ---- tests/golden/bptt_loop_hover.npz: ONE iteration of the reference's own BPTT.learn with its own actor (gen_shac.py::gen_bptt_loop) ----
def test_bptt_loop_fixture_first_episodes_are_the_oracle_envs():
    """the recorded (clipped) actions driven through the CPU oracle's HoverEnv from the recorded spawn states: reward and done of
    every agent's FIRST episode (5 steps; the re-spawns that follow come from the reference's RNG stream, which only the replay
    mode on the GPU reproduces) are the fixture's, bit for bit -- the env side of the loop is the pinned env"""
    from _golden import assert_bits_equal, consts_of
    fx = load("bptt_loop_hover")
    H, N = fx["reward"].shape
    env = oracle.OracleEnv(consts_of(fx), N, "hover", int(fx["max_episode_steps"]))
    env.reset_full_state(fx["fs_init"])
    alive = np.ones(N, bool)
    checked = 0
    for t in range(H):
        env.step(fx["action"][t])
        assert_bits_equal(env.a["reward"][alive], fx["reward"][t][alive], f"reward of the first episodes @ {t}")
        assert np.array_equal(env.a["done"][alive], fx["done"][t][alive]), f"done of the first episodes @ {t}"
        checked += int(alive.sum())
        alive &= fx["done"][t] == 0
        if not alive.any():
            break
    assert t == int(fx["max_episode_steps"]) - 1 and checked >= 4 * N          # everybody is truncated at step 5 at the latest
    assert (np.abs(fx["action"]) <= 1.0).all()                                   # tanh-squashed: the clip of BPTT.py:114-116 is the identity


def test_bptt_loop_loss_recurrence_and_adam_step():
    """actor_loss = mean_i sum_t -reward_t disc_t, disc <- disc gamma ~done + done (BPTT.py:123-127) from the recorded rewards / done
    flags; the recorded parameters after the step = clip_grad_norm_(0.5) + Adam(lr) from the recorded gradient (:130-133)"""
    fx = load("bptt_loop_hover")
    H, N = fx["reward"].shape
    g = np.float32(float(fx["gamma"]))
    loss, disc = np.zeros(N, np.float32), np.ones(N, np.float32)
    for t in range(H):
        done = fx["done"][t].astype(bool)
        loss = loss + np.float32(-1) * fx["reward"][t] * disc
        disc = disc * g * (~done).astype(np.float32) + done.astype(np.float32)
    assert abs(float(loss.mean()) - float(fx["actor_loss"])) <= 1e-6
    assert fx["done"].sum() > N                                                   # truncations of everybody + true episode ends
    gr = fx["actor_grad"].astype(np.float64)
    gr = gr * min(1.0, 0.5 / (np.linalg.norm(gr) + 1e-6))
    m, v = 0.1 * gr, 0.001 * gr * gr
    step = float(fx["lr"]) * (m / 0.1) / (np.sqrt(v / 0.001) + 1e-8)
    assert np.abs(fx["actor_params0"] - step - fx["actor_params1"]).max() <= 2e-7
    assert fx["actor_params0"].size == 13 * 128 + 128 + 128 * 64 + 64 + 2 * (64 * 64 + 64) * 2 + 2 * (64 * 4 + 4)


def test_shared_extractor_fixture_means_what_the_gpu_test_assumes():
    """tests/golden/shac_hover_shared.npz (oracle/gen_shac.py --only shac_hover_shared: MTDPolicy(share_features_extractor=True)): the critic's
    extractor block IS the actor's -- before and after the actor step, untouched by the critic steps --, the extractor block of the gradient
    clip_grad_norm_(critic.parameters()) saw is the actor loss's gradient (x the clip coefficients applied so far), the q-network part is the
    twin-Q gradient with the features detached, and the target's own extractor follows by Polyak averaging"""
    fx = load("shac_hover_shared")
    e = 13 * 128 + 128 + 128 * 64 + 64
    assert np.array_equal(fx["critic_params0"][:e], fx["actor_params0"][:e])
    steps = int(fx["gradient_steps"])
    for i in range(steps):
        assert np.array_equal(fx["critic_params"][i][:e], fx["actor_params1"][:e]), "the critic's optimiser never touches the shared extractor"
    # stale extractor gradient: the actor's (clip coefficient 1 here: |g_actor| << 0.5), rescaled in place by every critic clip
    na = float(np.linalg.norm(fx["actor_grad"].astype(np.float64)))
    assert na < 0.5
    stale = fx["actor_grad"][:e].astype(np.float64)
    for i in range(steps):
        assert np.abs(fx["critic_grad"][i][:e] - stale).max() <= 1e-7 * max(np.abs(stale).max(), 1e-30) + 1e-12
        n = float(np.linalg.norm(fx["critic_grad"][i].astype(np.float64)))
        stale = stale * min(1.0, 0.5 / (n + 1e-6))
    # q-network gradient at the recorded parameters, features detached
    H, N = fx["buf_reward"].shape
    obs = torch.from_numpy(fx["buf_obs"].reshape(H * N, 13)).double()
    act = torch.from_numpy(fx["buf_action"].reshape(H * N, 4)).double()
    ret = torch.from_numpy(fx["buf_returns"].reshape(-1)).double()
    prm = fx["critic_params0"].copy()
    prm[:e] = fx["actor_params1"][:e]
    p, qs = _critic(prm, obs, act)
    loss = torch.nn.functional.mse_loss(ret, torch.minimum(qs[0], qs[1]).view(-1))
    loss.backward()
    assert abs(float(loss) - float(fx["critic_loss"][0])) <= 1e-6
    g = p.grad.numpy()
    assert np.abs(g[e:] - fx["critic_grad"][0][e:]).max() <= 2e-6 * np.abs(g[e:]).max()
    # Polyak over ALL critic parameters, the shared extractor included (shac.py:277)
    tau = float(fx["tau"])
    t = fx["critic_params0"].astype(np.float64)
    for i in range(steps):
        t = (1 - tau) * t + tau * fx["critic_params"][i].astype(np.float64)
        assert np.abs(t - fx["target_params"][i]).max() <= 2e-7
    assert not np.array_equal(fx["target_params"][-1][:e], fx["critic_params"][-1][:e])
