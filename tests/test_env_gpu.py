"""Parity of the fused HIP env step (HoverEnv / NavigationEnv classes -> C-ABI -> kernel) with the
per-step traces captured from the reference envs:
  * scripted resets: reward / done / step_count / collision flags / collision_dis / pre-reset state
    bit-exact at every step (Nav reward to the acos tolerance stated in test_oracle_env_golden.py);
  * replay spawn mode: the env runs on its own from seed 42 -- spawn states, auto-resets and
    returned observations are bit-identical to the reference's run."""
import numpy as np
import pytest
import torch

from _golden import (ENV_DYN, ENV_KW, RACING_DYN, assert_bits_equal, assert_env_trace_close_to_unpatched_reference, consts_of,
                     decode_actions, load)
from test_oracle_env_golden import ENVS, run_env_fixture

pytestmark = pytest.mark.gpu


def make(name, fx, **kw):
    from visfly_amd.envs import HoverEnv, HoverEnv2, NavigationEnv, NavigationEnv2, RacingEnv
    kind = str(fx["kind"])
    cls = {"hover": HoverEnv, "nav": NavigationEnv, "racing": RacingEnv, "hover2": HoverEnv2, "nav2": NavigationEnv2}[kind]
    if kind == "racing":
        kw["gates"] = fx["gates"].tolist()
    return cls(num_agent_per_scene=fx["fs_init"].shape[0], num_scene=1, seed=int(fx["seed"]), visual=False,
               dynamics_kwargs=dict(RACING_DYN if kind == "racing" else ENV_DYN), device="cuda:0", tensor_output=True,
               constants=consts_of(fx), **ENV_KW[name], **kw)


@pytest.mark.parametrize("name", ENVS)
def test_env_trace_scripted_resets(name):
    def make_env(fx):
        env = make(name, fx)
        env.reset(state=torch.from_numpy(fx["fs_init"]))
        return env

    def step_fn(env, a):
        obs, reward, done, info = env.step(torch.from_numpy(a).cuda(), is_test=True)
        n = lambda t: t.cpu().numpy()
        return dict(reward=n(reward), done=n(done), step_count=n(env._step_count), is_collision=n(env.is_collision),
                    is_out_bounds=n(env.is_out_bounds), success=n(env.success), col_dis=n(env.collision_dis),
                    ext=n(env.extend_state))

    def reset_fn(env, idx, fs):
        env.reset_agent_by_id(torch.from_numpy(idx.astype(np.int64)), state=torch.from_numpy(fs))

    run_env_fixture(name, make_env, step_fn, reset_fn, lambda env: env.get_observation()["state"].cpu().numpy(),
                    lambda env: (env._next_target_i.cpu().numpy(), env._past_targets_num.cpu().numpy())
                    if str(load(name)["kind"]) == "racing" else None)


@pytest.mark.parametrize("name", ENVS)
def test_env_replay_mode_matches_reference_run(name):
    fx = load(name)
    acts = decode_actions(fx)
    env = make(name, fx, spawn="replay", replay_trig="cr")
    obs0 = env.reset()
    assert_bits_equal(env.full_state.cpu().numpy(), fx["fs_init"], f"{name} spawn states")
    assert_bits_equal(obs0["state"].cpu().numpy(), fx["obs0_state"], f"{name} reset() observation")
    if "obs0_cv" in fx:
        assert_bits_equal(obs0["collision_vector"].cpu().numpy(), fx["obs0_cv"], f"{name} reset() collision_vector entry")
    if str(fx["kind"]) == "racing":
        assert np.array_equal(obs0["gate"].cpu().numpy(), fx["obs0_gate"][:, 0]), f"{name} reset() gate entry (the stale one)"
    keep = list(fx["keep_steps"])
    trace_r, trace_d = [], []
    for k in range(acts.shape[0]):
        obs, reward, done, info = env.step(torch.from_numpy(acts[k]).cuda())
        r = reward.cpu().numpy()
        assert_bits_equal(r, fx["reward"][k], f"{name} reward @ {k}")      # Navigation incl.: acos in "cr" mode vs the CR-trig reference
        d = done.cpu().numpy()
        assert np.array_equal(d.astype(np.uint8), fx["done"][k]), f"{name} done @ {k}"
        trace_r.append(r)
        trace_d.append(d.astype(np.uint8))
        sel = fx["ev_step"] == k
        if sel.any():
            idx = fx["ev_agent"][sel]
            assert_bits_equal(env.full_state.cpu().numpy()[idx], fx["ev_fs"][sel], f"{name} auto-reset states @ {k}")
            i0 = int(idx[0])
            assert info[i0]["episode"]["l"] == fx["step_count"][k][i0]
            assert info[i0]["TimeLimit.truncated"] == bool(fx["step_count"][k][i0] >= int(fx["max_episode_steps"]))
            # everything collect_info (droneGymEnv.py:238-275) put into the info dict of EVERY episode that ended in this step
            for j in np.nonzero(sel)[0]:
                d = info[int(fx["ev_agent"][j])]
                ep = d["episode"]
                assert_bits_equal(np.float32(ep["r"]), fx["ev_r"][j], f"{name} episode r @ {k}")
                assert int(ep["l"]) == int(fx["ev_l"][j])
                assert_bits_equal(np.float32(ep["t"]), fx["ev_t"][j], f"{name} episode t @ {k}")
                got = (int(bool(d["is_success"])) | (int(bool(d["TimeLimit.truncated"])) << 1) | (int(bool(d["episode_done"])) << 2)
                       | (int(bool(ep["extra"]["collision"])) << 3))
                assert got == int(fx["ev_flags"][j]), f"{name} info flags @ {k}: {got} vs {int(fx['ev_flags'][j])}"
                tob = d["terminal_observation"]["state"]
                tob = tob.cpu().numpy() if hasattr(tob, "cpu") else np.asarray(tob)
                assert_bits_equal(tob, fx["ev_tobs"][j], f"{name} terminal observation @ {k}")
        if k in keep:
            assert_bits_equal(obs["state"].cpu().numpy(), fx["obs_state_keep"][keep.index(k)], f"{name} obs @ {k}")
        if "obs_cv" in fx:      # NavigationEnv2: the collision_vector entry of the returned observation
            assert_bits_equal(obs["collision_vector"].cpu().numpy(), fx["obs_cv"][k], f"{name} collision_vector obs @ {k}")
        if str(fx["kind"]) == "racing":
            # the index INSIDE the returned observation (pre-pass unless some agent ended its episode in the step), and the env's own
            assert np.array_equal(obs["gate"].cpu().numpy(), fx["obs_gate"][k][:, 0]), f"{name} gate obs @ {k}"
            assert np.array_equal(env._next_target_i.cpu().numpy(), fx["gate"][k]), f"{name} _next_target_i @ {k}"
            if sel.any():
                tg = [int(info[int(i)]["terminal_observation"]["gate"]) for i in fx["ev_agent"][sel]]
                assert tg == list(fx["ev_tgate"][np.nonzero(sel)[0]]), f"{name} terminal gate @ {k}"
    # the whole 256-step run against the reference EXACTLY as torch runs it (no CR patches): done flags identical, rewards <= 5e-7
    assert acts.shape[0] == 256
    assert_env_trace_close_to_unpatched_reference(name, fx, np.stack(trace_r), np.stack(trace_d))


def test_racing2_replay_matches_reference_run():
    """RacingEnv2 (envs/RacingEnv.py:218-267; repaired-oracle fixture env_racing2): the 16-column gate-relative observation and
    the (N,1) gate entry, bit for bit over the reference's own seed-42 run incl. auto-resets; reward / done as RacingEnv"""
    from visfly_amd.envs import RacingEnv2
    fx = load("env_racing2")
    acts = decode_actions(fx)
    env = RacingEnv2(num_agent_per_scene=fx["fs_init"].shape[0], num_scene=1, seed=int(fx["seed"]), visual=False,
                     dynamics_kwargs=dict(RACING_DYN), device="cuda:0", tensor_output=True, constants=consts_of(fx),
                     gates=fx["gates"].tolist(), spawn="replay", replay_trig="cr", **ENV_KW["env_racing2"])
    assert env.observation_space["state"].shape == (16,)
    obs0 = env.reset()
    assert_bits_equal(env.full_state.cpu().numpy(), fx["fs_init"], "racing2 spawn states")
    assert_bits_equal(obs0["state"].cpu().numpy(), fx["obs0_state"], "racing2 reset() observation")
    assert np.array_equal(obs0["gate"].cpu().numpy(), fx["obs0_gate"])
    keep = list(fx["keep_steps"])
    checked_terminal = 0
    for k in range(acts.shape[0]):
        obs, reward, done, info = env.step(torch.from_numpy(acts[k]).cuda())
        assert_bits_equal(reward.cpu().numpy(), fx["reward"][k], f"racing2 reward @ {k}")
        assert np.array_equal(done.cpu().numpy().astype(np.uint8), fx["done"][k]), f"racing2 done @ {k}"
        assert obs["gate"].shape == (fx["fs_init"].shape[0], 1)
        assert np.array_equal(obs["gate"].cpu().numpy(), fx["obs_gate"][k]), f"racing2 gate obs @ {k}"
        if k in keep:
            assert_bits_equal(obs["state"].cpu().numpy(), fx["obs_state_keep"][keep.index(k)], f"racing2 obs @ {k}")
        sel = fx["ev_step"] == k
        if sel.any():
            for j in np.nonzero(sel)[0]:
                d = info[int(fx["ev_agent"][j])]
                t = d["terminal_observation"]
                assert t["state"].shape == (16,) and t["gate"].shape == (1,)
                assert int(t["gate"][0]) == int(fx["ev_tgate"][j]), f"racing2 terminal gate @ {k}"
                assert_bits_equal(t["state"].cpu().numpy(), fx["ev_tobs"][j], f"racing2 terminal observation @ {k}")
                assert_bits_equal(np.float32(d["episode"]["r"]), fx["ev_r"][j], f"racing2 episode r @ {k}")
                assert int(d["episode"]["l"]) == int(fx["ev_l"][j])
            checked_terminal += 1
    assert checked_terminal > 0
    with pytest.raises(Exception):
        env.step_n(torch.zeros((2, fx["fs_init"].shape[0], 4), device="cuda"))


def test_racing2_without_done_list_and_with_reassigned_targets():
    """RacingEnv2 after enable_done_list(False) steps on the torch expressions (same rows, bit for bit, as the one-launch
    path that needs the step's done count), and assigning `targets` refreshes the host copy vf_race_obs reads: one-launch rows == torch rows"""
    from visfly_amd.envs import RacingEnv2
    N = 512
    g = torch.Generator().manual_seed(3)
    acts = [(torch.tensor([-0.8333] * 4) + (torch.rand((N, 4), generator=g) * 2 - 1) * 0.2).clamp(-1, 1).cuda() for _ in range(40)]
    rows = []
    for on in (True, False):
        env = RacingEnv2(num_agent_per_scene=N, seed=4, dynamics_kwargs=dict(RACING_DYN), device="cuda:0", max_episode_steps=12, tensor_output=True)
        if not on:
            env.enable_done_list(False)
        env.reset()
        out = []
        for a in acts:
            obs, r, d, _ = env.step(a)
            out.append((obs["state"].clone(), obs["gate"].clone(), r.clone(), d.clone()))
        rows.append(out)
        if on:      # new gates for the observation: the launch (N > 1 rows) and the torch expressions (row by row) read the same ones
            env.targets = [[3., 3., 1.], [7., 1., 2.], [4., -3., 1.], [0., -2., 1.5]]
            raw, gate = env.envs.dynamics.state.detach().contiguous(), env._gate
            fast = env._race_state(raw, gate)
            slow = torch.cat([env._race_state(raw[i:i + 1], gate[i:i + 1]) for i in range(8)])
            assert torch.equal(fast[:8], slow)
            assert torch.equal(fast[:, 0:3], (env.targets[gate.long() % 4] - raw[:, 0:3]) / torch.full((1,), 10.0, device="cuda:0"))
        env.close()
    assert any(bool(x[3].any()) for x in rows[0]), "episodes ended"
    for k, (a, b) in enumerate(zip(*rows)):
        for x, y in zip(a, b):
            assert torch.equal(x, y), k


def test_device_spawn_statistics_and_conventions():
    """device (Philox) spawn mode: one launch per step; spawn box respected, counters reset, reward/done
    are the pre-reset values, obs the post-reset ones, episode stats consistent."""
    from visfly_amd.envs import HoverEnv
    N = 4096
    env = HoverEnv(num_agent_per_scene=N, seed=3, dynamics_kwargs=dict(ENV_DYN), device="cuda:0",
                   max_episode_steps=32, tensor_output=True)
    obs = env.reset()
    p = obs["state"][:, :3]
    lo, hi = torch.tensor([0., -1., 1.], device="cuda"), torch.tensor([2., 1., 2.], device="cuda")
    assert (p >= lo).all() and (p <= hi).all()
    assert p.std(dim=0).min() > 0.2  # actually random
    assert (obs["state"][:, 3:7] == torch.tensor([1., 0, 0, 0], device="cuda")).all()
    g = torch.Generator(device="cuda").manual_seed(0)
    seen_done = 0
    for k in range(70):
        a = (torch.rand((N, 4), device="cuda", generator=g) * 2 - 1) * 0.3 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda")
        obs, reward, done, info = env.step(a.clamp(-1, 1))
        sc = env._step_count
        assert (sc[done] == 0).all()          # counters of reset agents cleared
        assert (sc <= 32).all()
        if done.any():
            seen_done += int(done.sum())
            i = int(torch.where(done)[0][0])
            d = info[i]
            assert d["episode"]["l"] >= 1 and ("terminal_observation" in d)
            pr = obs["state"][done][:, :3]
            assert (pr >= lo).all() and (pr <= hi).all()   # post-reset observation is a fresh spawn
            assert (env.t[done] <= 6.28).all()
    assert seen_done >= 2 * N  # every agent timed out at least twice
    # different seeds / agents give different spawns; same seed reproduces
    env2 = HoverEnv(num_agent_per_scene=N, seed=3, dynamics_kwargs=dict(ENV_DYN), device="cuda:0", tensor_output=True)
    env3 = HoverEnv(num_agent_per_scene=N, seed=4, dynamics_kwargs=dict(ENV_DYN), device="cuda:0", tensor_output=True)
    a, b, c = env2.reset()["state"], HoverEnv(num_agent_per_scene=N, seed=3, dynamics_kwargs=dict(ENV_DYN),
                                             device="cuda:0", tensor_output=True).reset()["state"], env3.reset()["state"]
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_numpy_output_convention():
    from visfly_amd.envs import HoverEnv
    env = HoverEnv(num_agent_per_scene=64, dynamics_kwargs=dict(ENV_DYN), device="cuda:0")  # tensor_output=False default
    obs = env.reset()
    assert isinstance(obs["state"], np.ndarray)
    obs, r, d, info = env.step(np.zeros((64, 4), np.float32))
    assert isinstance(r, np.ndarray) and r.dtype == np.float32 and d.dtype == np.int32   # droneGymEnv.py:218
    with pytest.raises(AssertionError):
        HoverEnv(num_agent_per_scene=4, dynamics_kwargs=dict(ENV_DYN), device="cuda:0").step(np.zeros((4, 4)))


def test_config3_navigation_rk4_ctrl_delay_drag_randomisation():
    """BASELINE configs[2]: NavigationEnv, RK4 + ctrl_delay + drag domain randomisation.  The RK4 path is
    the documented repair; per-agent drag coefficients drawn on the device are fed to
    the CPU oracle, after which dynamics AND env outputs must agree bit-for-bit (reward to acos tolerance)."""
    import oracle
    from visfly_amd.envs import NavigationEnv
    N = 1000
    dkw = dict(action_type="bodyrate", integrator="rk4", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True, drag_random=0.1)
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
    env = NavigationEnv(num_agent_per_scene=N, seed=9, dynamics_kwargs=dkw, random_kwargs=spawn, device="cuda:0",
                        max_episode_steps=40, tensor_output=True)
    obs = env.reset()
    dyn = env.envs.dynamics
    kl, kq = dyn.drag_coefficients
    c = dyn.constants
    for k, mean in ((kl, c["k_lin"]), (kq, c["k_quad"])):
        f = (k / torch.tensor(mean, device="cuda")).cpu()
        assert (f >= 0.9 - 1e-6).all() and (f <= 1.1 + 1e-6).all() and f.std(dim=0).min() > 0.02
    ref = oracle.OracleEnv(c, N, "nav", 40, target=[9., 0., 1.])
    ref.dyn.klin = np.ascontiguousarray(kl.cpu().numpy().T)
    ref.dyn.kquad = np.ascontiguousarray(kq.cpu().numpy().T)
    ref.reset_full_state(env.full_state.cpu().numpy())
    g = torch.Generator().manual_seed(0)
    for k in range(30):   # < max_episode_steps: no timeouts; collisions may end episodes -> is_test keeps both in step
        a = ((torch.rand((N, 4), generator=g) * 2 - 1) * 0.5 + torch.tensor([-0.3, 0, 0, 0])).clamp(-1, 1)
        o, r, d, _ = env.step(a.cuda(), is_test=True)
        ro, rr, rd = ref.step(a.numpy())
        assert_bits_equal(o["state"].cpu().numpy(), ro, f"rk4+drag state @ {k}")
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), rd)
        assert_bits_equal(r.cpu().numpy(), rr, f"rk4+drag reward @ {k}")    # shared acos kernel: bit-identical to the oracle
    # auto-reset redraws the drag factors of the re-spawned agents
    env2 = NavigationEnv(num_agent_per_scene=256, seed=9, dynamics_kwargs=dkw, random_kwargs=spawn, device="cuda:0",
                         max_episode_steps=5, tensor_output=True)
    env2.reset()
    k0 = env2.envs.dynamics.drag_coefficients[0].clone()
    for _ in range(5):
        env2.step(torch.zeros((256, 4), device="cuda"))
    assert (env2.envs.dynamics.drag_coefficients[0] != k0).any(dim=1).all()


@pytest.mark.parametrize("mode", ["velocity", "position"])
def test_env_with_geometric_controller(mode):
    """HoverEnv driven through the velocity / position action types: same fused launch, the
    geometric controller in front of the sub-steps.  State within 1e-4 of the oracle's column scale over 30
    free-running steps (same bound as the golden fixtures), masks and counters exact."""
    import oracle
    from visfly_amd.envs import HoverEnv
    N = 777
    dkw = dict(action_type=mode, integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
    env = HoverEnv(num_agent_per_scene=N, seed=3, dynamics_kwargs=dkw, device="cuda:0", max_episode_steps=64,
                   tensor_output=True)
    env.reset()
    c = env.envs.dynamics.constants
    assert int(c["action_type"]) == {"velocity": 2, "position": 3}[mode]
    ref = oracle.OracleEnv(c, N, "hover", 64, target=[1., 0., 1.5])
    ref.reset_full_state(env.full_state.cpu().numpy())
    g = torch.Generator().manual_seed(1)
    worst = 0.0
    for k in range(30):
        a = (torch.rand((N, 4), generator=g) * 2 - 1) * 0.2
        o, r, d, _ = env.step(a.cuda(), is_test=True)
        ro, rr, rd = ref.step(a.numpy())
        scale = np.maximum(np.abs(ro).max(0), 1e-3)
        dev = (np.abs(o["state"].cpu().numpy() - ro) / scale).max()
        worst = max(worst, dev)
        assert dev <= 1e-4, f"{mode} state @ {k}: {dev:.2e}"
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), rd)
        assert np.allclose(r.cpu().numpy(), rr, rtol=0, atol=1e-5)
    print(f"{mode}: worst state deviation / column scale = {worst:.2e}")


def test_edge_cases_empty_reset_single_agent_action_validation():
    import oracle
    from visfly_amd.envs import HoverEnv
    from visfly_amd import Dynamics
    # empty index list: no-op
    d = Dynamics(num=70, device="cuda:0", action_type="bodyrate", dt=0.0025, ctrl_dt=0.02)
    before = d.full_state.clone()
    d.reset(indices=torch.zeros(0, dtype=torch.int64))
    assert torch.equal(before, d.full_state)
    # one agent (a single lane of a padded wave), euler orientation output shape
    env = HoverEnv(num_agent_per_scene=1, dynamics_kwargs=dict(ENV_DYN), device="cuda:0", tensor_output=True, seed=11)
    obs = env.reset()
    ref = oracle.OracleEnv(env.envs.dynamics.constants, 1, "hover", 256)
    ref.reset_full_state(env.full_state.cpu().numpy())
    for k in range(20):
        a = torch.tensor([[-0.3, 0.1 * np.sin(k), 0.05, 0.0]])
        o, r, dn, _ = env.step(a.cuda())
        ro, rr, rd = ref.step(a.numpy())
        assert_bits_equal(o["state"].cpu().numpy(), ro, f"single agent @ {k}")
        assert_bits_equal(r.cpu().numpy(), rr, f"single agent reward @ {k}")
    de = Dynamics(num=5, device="cuda:0", ori_output_type="euler")
    assert de.orientation.shape == (5, 3) and de.state.shape == (5, 12)
    # |action| > 1 is an AssertionError when validation is on (droneGymEnv.py:144)
    env2 = HoverEnv(num_agent_per_scene=8, dynamics_kwargs=dict(ENV_DYN), device="cuda:0", tensor_output=True,
                    validate_actions=True)
    env2.reset()
    with pytest.raises(AssertionError):
        env2.step(torch.full((8, 4), 1.5, device="cuda"))
    with pytest.raises(NotImplementedError):
        HoverEnv(num_agent_per_scene=8, visual=True, device="cuda:0")


def test_million_agent_invariants():
    """maximum practical size: 1 048 576 agents through the fused env step (plain kernel shape)"""
    from visfly_amd.envs import HoverEnv
    N = 1 << 20
    env = HoverEnv(num_agent_per_scene=N, dynamics_kwargs=dict(ENV_DYN), device="cuda:0", tensor_output=True,
                   max_episode_steps=16, seed=2)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    total_done = 0
    for k in range(20):
        a = ((torch.rand((N, 4), device="cuda", generator=g) * 2 - 1) * 0.2 + torch.tensor([-1 / 3, 0, 0, 0], device="cuda"))
        obs, r, d, _ = env.step(a)
        total_done += int(d.sum())
    assert torch.isfinite(obs["state"]).all() and torch.isfinite(r).all()
    assert total_done >= N                       # everyone timed out once at step 16
    assert int(env._step_count.max()) <= 16
    assert ((obs["state"][:, 3:7].norm(dim=1) - 1).abs() < 1e-5).all()


def test_scene_reset_and_stack_recover():
    from visfly_amd.envs import HoverEnv
    env = HoverEnv(num_agent_per_scene=8, num_scene=4, dynamics_kwargs=dict(ENV_DYN), device="cuda:0", tensor_output=True, seed=1)
    env.reset()
    for _ in range(5):
        env.step(torch.zeros((32, 4), device="cuda"))
    before = env.full_state.clone()
    env.reset_env_by_id(torch.tensor([1, 3]))                       # droneGymEnv.py:329-337
    after, sc = env.full_state, env._step_count
    changed = (after != before).any(dim=1).cpu()
    assert changed.tolist() == [False] * 8 + [True] * 8 + [False] * 8 + [True] * 8
    assert sc.cpu().tolist() == [5] * 8 + [0] * 8 + [5] * 8 + [0] * 8
    env.envs.stack()                                                # droneEnv.py:387-396
    q, v = env.orientation.clone(), env.velocity.clone()
    for _ in range(3):
        env.step(torch.rand((32, 4), device="cuda") * 0.2 - 0.4)
    env.envs.recover()
    assert torch.equal(env.orientation, q) and torch.equal(env.velocity, v) and int(env._step_count.max()) == 0


@pytest.mark.parametrize("kind", ["hover", "nav", "racing"])
def test_full_size_batch_vs_oracle(kind):
    """BASELINE-size batch (65 536 agents, configs[1..2]): the fused launch against the OpenMP oracle on the very same
    spawn states, 6 control steps, bit-exact state / masks (Nav reward to the acos tolerance)."""
    import oracle
    from visfly_amd.envs import HoverEnv, NavigationEnv, RacingEnv
    N = 65536
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
    if kind == "hover":
        env = HoverEnv(num_agent_per_scene=N, seed=11, dynamics_kwargs=dict(ENV_DYN), device="cuda:0", max_episode_steps=256,
                       tensor_output=True)
        ref = oracle.OracleEnv(env.envs.dynamics.constants, N, "hover", 256, target=[1., 0., 1.5])
    elif kind == "nav":
        env = NavigationEnv(num_agent_per_scene=N, seed=11, dynamics_kwargs=dict(ENV_DYN), random_kwargs=spawn, device="cuda:0",
                            max_episode_steps=256, tensor_output=True)
        ref = oracle.OracleEnv(env.envs.dynamics.constants, N, "nav", 256, target=[9., 0., 1.])
    else:
        env = RacingEnv(num_agent_per_scene=N, seed=11, dynamics_kwargs=dict(RACING_DYN), device="cuda:0", max_episode_steps=256,
                        tensor_output=True)
        ref = oracle.OracleEnv(env.envs.dynamics.constants, N, "racing", 256, success_radius=0.3,
                               gates=[[4, 4, 1.], [8, 0, 2.], [5, -4, 1.], [1, -1, 1.]])
    env.reset()
    ref.reset_full_state(env.full_state.cpu().numpy())
    g = torch.Generator().manual_seed(3)
    hover = torch.tensor([-0.8333] * 4 if kind == "racing" else [-1 / 3, 0, 0, 0])
    for k in range(6):
        a = (hover + (torch.rand((N, 4), generator=g) * 2 - 1) * 0.3).clamp(-1, 1)
        o, r, d, _ = env.step(a.cuda(), is_test=True)
        ro, rr, rd = ref.step(a.numpy())
        assert_bits_equal(o["state"].cpu().numpy(), ro, f"{kind} state @ {k}")
        assert np.array_equal(d.cpu().numpy().astype(np.uint8), rd), f"{kind} done @ {k}"
        assert_bits_equal(r.cpu().numpy(), rr, f"{kind} reward @ {k}")     # nav too: the acos kernel is shared with the oracle


def test_racing_info_reports_gates_passed_in_the_finished_episode():
    """info["episode"]["extra"]["past_gate"] (RacingEnv.collect_info, RacingEnv.py:113-116) is the count of the episode
    that just ended -- captured by the step kernel before the auto-reset clears it; a pass pays +20 (:215)"""
    from visfly_amd.envs import RacingEnv
    gates = load("env_racing")["gates"].tolist()          # gates next to the spawn boxes: random flight passes them
    N = 512
    env = RacingEnv(num_agent_per_scene=N, seed=4, dynamics_kwargs=dict(RACING_DYN), device="cuda:0", max_episode_steps=48,
                    tensor_output=True, gates=gates)
    env.reset()
    g = torch.Generator().manual_seed(2)
    count = np.zeros(N, np.int64)
    seen = 0
    for k in range(120):
        a = (torch.tensor([-0.8333] * 4) + (torch.rand((N, 4), generator=g) * 2 - 1) * 0.08).clamp(-1, 1)
        obs, r, d, info = env.step(a.cuda())
        count += (r.cpu().numpy() > 10.0)
        for i in np.nonzero(d.cpu().numpy())[0]:
            assert info[i]["episode"]["extra"]["past_gate"] == count[i], (k, i)
            seen += int(count[i] > 0)
            count[i] = 0
    assert seen > 0


def test_imu_noise_replay_mode_matches_reference():
    """random_kwargs["noise_kwargs"]["IMU"] with non-zero amplitude (droneEnv.py:99-125): in replay-spawn mode the env draws the
    reference's th.rand(N,13) at every update_observation, so sensor_obs["IMU"] = state + (u - 0.5) * half + mean with the
    quaternion re-normalised is the reference's, through resets"""
    from visfly_amd.envs import HoverEnv
    fx = load("env_hover_imu")
    acts = decode_actions(fx)
    N = fx["imu"].shape[1]
    noise = {"IMU": {"model": "UniformNoiseModel", "kwargs": {"mean": fx["noise_mean"], "half": fx["noise_half"]}}}
    rk = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [1.0, 1.0, 0.5]}}]},
          "noise_kwargs": noise}
    env = HoverEnv(num_agent_per_scene=N, seed=int(fx["seed"]), dynamics_kwargs=dict(ENV_DYN), device="cuda:0", tensor_output=True,
                   max_episode_steps=int(fx["max_episode_steps"]), random_kwargs=rk, spawn="replay", replay_trig="cr", constants=consts_of(fx))
    env.reset()
    keep = list(fx["keep"])
    for k in range(acts.shape[0] + 1):
        if k in keep:
            j = keep.index(k)
            assert_bits_equal(env.state.cpu().numpy(), fx["state"][j], f"state @ {k}")
            imu = env.sensor_obs["IMU"].cpu().numpy()
            # th.nn.functional.normalize on the GPU vs the reference's CPU: one division by max(norm, eps) per element
            assert np.abs(imu - fx["imu"][j]).max() <= 2.4e-7, (k, np.abs(imu - fx["imu"][j]).max())
            same = (imu.view(np.uint32) == fx["imu"][j].view(np.uint32)).mean()
            assert same > 0.93, same      # measured 0.965: the quaternion columns go through a GPU normalise
        if k < acts.shape[0]:
            env.step(torch.from_numpy(acts[k]).cuda())
    # device-spawn mode: same distribution, own generator
    env2 = HoverEnv(num_agent_per_scene=4096, dynamics_kwargs=dict(ENV_DYN), device="cuda:0", tensor_output=True, random_kwargs=rk)
    env2.reset()
    env2.step(torch.zeros((4096, 4), device="cuda"))
    d = (env2.sensor_obs["IMU"] - env2.state).cpu().numpy()
    mean, half = fx["noise_mean"], fx["noise_half"]
    cols = [0, 1, 2, 7, 8, 9, 10, 11, 12]                                  # quaternion columns are re-normalised
    assert np.abs(d[:, cols].mean(0) - mean[cols]).max() < 0.02 and (np.abs(d[:, cols] - mean[cols]) <= half[cols] / 2 + 1e-6).all()
    assert np.allclose(np.linalg.norm(env2.sensor_obs["IMU"][:, 3:7].cpu().numpy(), axis=1), 1.0, atol=1e-6)
    assert env2.sensor_obs["IMU"] is env2.sensor_obs["IMU"]                # one draw per step
    with pytest.raises(NotImplementedError):
        HoverEnv(num_agent_per_scene=4, dynamics_kwargs=dict(ENV_DYN), device="cuda:0",
                 random_kwargs=dict(rk, noise_kwargs={"IMU": {"model": "GaussianNoiseModel", "kwargs": {"mean": 0, "std": 1}}}))


def test_racing_gate_entry_switch():
    """obs_gate_exact=False (what trainers set when their policy does not read "gate"): the returned entry is the env's
    current index, no per-step bookkeeping; True (default): the reference's returned entry (checked against the fixture above)"""
    from visfly_amd.envs import RacingEnv
    fx = load("env_racing")
    env = RacingEnv(num_agent_per_scene=256, seed=3, dynamics_kwargs=dict(RACING_DYN), device="cuda:0", tensor_output=True,
                    max_episode_steps=64, gates=fx["gates"].tolist())
    env.obs_gate_exact = False
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(40):
        a = ((torch.rand((256, 4), device="cuda", generator=g) * 2 - 1) * 0.08 - 0.8333).clamp(-1, 1)
        obs, _, _, _ = env.step(a)
        assert torch.equal(obs["gate"], env._next_target_i)
