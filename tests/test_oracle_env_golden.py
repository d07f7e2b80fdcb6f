"""Pins the oracle's env layer (bbox collision, reward, counters, masks) bit-for-bit against
per-step traces captured from the stub-imported reference HoverEnv / NavigationEnv
(oracle/gen_golden.py; resets are scripted from the captured reset events)."""
import numpy as np
import pytest

import oracle
from _golden import assert_bits_equal, assert_env_trace_close_to_unpatched_reference, consts_of, decode_actions, load

ENVS = ["env_hover", "env_hover_256", "env_nav", "env_nav_close", "env_racing", "env_hover2", "env_nav2"]


def run_env_fixture(name, make_env, step_fn, reset_fn, state_fn, gates_fn=None):
    """drives any env implementation through the fixture; shared with the GPU parity test"""
    fx = load(name)
    acts = decode_actions(fx)
    env = make_env(fx)
    steps = acts.shape[0]
    keep = list(fx["keep_steps"])
    for k in range(steps):
        out = step_fn(env, acts[k])
        # NavigationEnv's reward goes through acos (NavigationEnv.py:91): CR-trig reference, "cr" mode of the oracle -> bit level
        assert_bits_equal(out["reward"], fx["reward"][k], f"{name} reward @ {k}")
        assert np.array_equal(out["done"].astype(np.uint8), fx["done"][k]), f"{name} done @ {k}"
        assert np.array_equal(out["step_count"], fx["step_count"][k]), f"{name} step_count @ {k}"
        assert np.array_equal(out["is_collision"].astype(np.uint8), fx["is_collision"][k]), f"{name} is_collision @ {k}"
        assert np.array_equal(out["is_out_bounds"].astype(np.uint8), fx["is_out_bounds"][k]), f"{name} oob @ {k}"
        assert np.array_equal(out["success"].astype(np.uint8), fx["success"][k]), f"{name} success @ {k}"
        assert_bits_equal(out["col_dis"], fx["col_dis"][k], f"{name} collision_dis @ {k}")
        if k in keep:
            assert_bits_equal(out["ext"], fx["ext_pre_keep"][keep.index(k)], f"{name} extend_state(pre-reset) @ {k}")
        sel = fx["ev_step"] == k
        assert np.array_equal(np.nonzero(out["done"])[0], fx["ev_agent"][sel]), f"{name} reset set @ {k}"
        if sel.any():
            reset_fn(env, fx["ev_agent"][sel], fx["ev_fs"][sel])
        if k in keep:
            assert_bits_equal(state_fn(env), fx["obs_state_keep"][keep.index(k)], f"{name} obs(post-reset) @ {k}")
        if str(fx["kind"]) == "racing":   # gate bookkeeping after the auto-reset, as observed
            g, p = gates_fn(env)
            assert np.array_equal(g, fx["gate"][k]), f"{name} next gate @ {k}"
            assert np.array_equal(p, fx["past"][k]), f"{name} passed gates @ {k}"


@pytest.mark.parametrize("name", ENVS)
def test_oracle_env_trace(name):
    def make_env(fx):
        racing = str(fx["kind"]) == "racing"
        env = oracle.OracleEnv(consts_of(fx), fx["fs_init"].shape[0], str(fx["kind"]),
                               int(fx["max_episode_steps"]), target=fx["target"],
                               success_radius=0.3 if racing else 0.5, gates=fx["gates"] if racing else None)
        env.reset_full_state(fx["fs_init"])
        if racing:
            assert np.array_equal(env.a["next_gate"], fx["gate0"])
        return env

    def step_fn(env, a):
        env.step(a)
        out = {k: v.copy() for k, v in env.a.items()}
        out["ext"] = env.dyn.extend_state
        return out

    def state_fn(env):
        return env.obs_state

    run_env_fixture(name, make_env, step_fn, lambda env, idx, fs: env.reset_agents(idx, fs), state_fn,
                    lambda env: (env.a["next_gate"], env.a["past_gates"]))


@pytest.mark.parametrize("name", ENVS + ["env_racing2"])
def test_env_fixture_stays_close_to_the_unpatched_reference(name):
    """what the CR patches cost at the env level: the fixture's reward / done trace (CR-patched reference, pinned to the bit above)
    against the reference run exactly as torch runs it, all 256 steps"""
    fx = load(name)
    assert fx["reward"].shape[0] == 256
    assert_env_trace_close_to_unpatched_reference(name, fx, fx["reward"], fx["done"])


def test_run_steps_equals_step_loop():
    """the chunked multi-step driver bench.py times (vfo_env_run_steps, one OpenMP region) is the same arithmetic as the
    per-step entry points the golden fixtures pin"""
    import oracle
    from _golden import consts_of, load
    fx = load("env_hover")
    c = consts_of(fx)
    N = 5000                       # > 4096: the per-step path takes its OpenMP branch too
    rng = np.random.default_rng(3)
    fs = np.zeros((N, 22), np.float32)
    fs[:, 0:3] = (np.array([1, 0, 1.5]) + rng.uniform(-1, 1, (N, 3))).astype(np.float32)
    fs[:, 3] = 1.0
    fs[:, 13:17], fs[:, 17:21] = c["w_init"], c["T_init"]
    acts = np.clip(rng.uniform(-.5, .5, (5, N, 4)), -1, 1).astype(np.float32)
    for threads in (1, 3, oracle.max_threads()):
        oracle.set_threads(threads)
        a, b = (oracle.OracleEnv(c, N, "hover", 256) for _ in range(2))
        a.reset_full_state(fs)
        b.reset_full_state(fs)
        for k in range(13):
            a.step(acts[k % 5])
        b.run_steps(acts, 13)
        assert np.array_equal(a.dyn.S.view(np.uint32), b.dyn.S.view(np.uint32))
        assert np.array_equal(a.dyn.Q.view(np.uint32), b.dyn.Q.view(np.uint32)) and a.dyn.tick.value == b.dyn.tick.value
        for k in ("reward", "rewards", "done", "step_count", "col_dis", "once_collided"):
            assert np.array_equal(a.a[k], b.a[k]), k
    oracle.set_threads(oracle.max_threads())
