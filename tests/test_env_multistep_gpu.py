"""Multi-step launch paths of the env boundary (vf_env_step_n, vf_env_graph_*, the output ring of step()) against the
per-call path: same seeds, same actions -> every returned array and the final slab bit-identical, through auto-resets.
Pose hand-off (vf_env_export_pose) against the full_state columns, after steps and after auto-resets."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DYN = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)


def bits(t):
    return t.detach().cpu().contiguous().view(torch.uint8 if t.dtype == torch.bool else torch.int32).numpy()


def same(a, b, what):
    assert np.array_equal(bits(a), bits(b)), what


def make(cls_name, N, seed=7, **kw):
    import visfly_amd.envs as E
    dyn = dict(DYN, action_type="thrust") if cls_name == "RacingEnv" else dict(DYN)
    env = getattr(E, cls_name)(num_agent_per_scene=N, seed=seed, dynamics_kwargs=dyn, device="cuda:0", tensor_output=True,
                               max_episode_steps=12, **kw)          # short episodes: auto-resets inside every run
    env.reset()
    return env


def actions(N, K, seed=0, wide=0.6):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.rand((K, N, 4), device="cuda", generator=g) * 2 - 1).mul_(wide).contiguous()


@pytest.mark.parametrize("cls_name,N", [("HoverEnv", 1000), ("NavigationEnv", 4099), ("RacingEnv", 777), ("HoverEnv", 40000)])
def test_step_n_equals_k_steps(cls_name, N):
    K = 40
    A = actions(N, K)
    ref, env = make(cls_name, N), make(cls_name, N)
    outs = [ref.step(A[k]) for k in range(K)]
    obs, reward, done = env.step_n(A)
    assert obs.shape == (K, N, 13) and reward.shape == (K, N) and done.shape == (K, N) and done.dtype == torch.bool
    assert bool(done.any()), "the run must contain auto-resets"
    for k in range(K):
        same(obs[k], outs[k][0]["state"], f"obs @ {k}")
        same(reward[k], outs[k][1], f"reward @ {k}")
        same(done[k], outs[k][2], f"done @ {k}")
    same(env.state_slab, ref.state_slab, "slab after the rollout")
    same(env._ep_return, ref._ep_return, "episode returns")
    same(env._ep_length, ref._ep_length, "episode lengths")
    same(env.get_observation()["state"], outs[-1][0]["state"], "get_observation() after step_n")
    same(env.reward, outs[-1][1], "env.reward after step_n")
    if cls_name == "RacingEnv":
        same(env.get_observation()["gate"], ref.get_observation()["gate"], "gate observation")
    # a second rollout continues from the same state as further step() calls
    B = actions(N, K, seed=1)
    outs2 = [ref.step(B[k]) for k in range(K)]
    obs2, reward2, done2 = env.step_n(B)
    same(obs2[K - 1], outs2[-1][0]["state"], "second rollout: last obs")
    same(env.state_slab, ref.state_slab, "slab after the second rollout")


@pytest.mark.parametrize("cls_name,N", [("HoverEnv", 1000), ("NavigationEnv", 4099), ("RacingEnv", 777), ("HoverEnv", 70000)])
def test_fused_rollout_equals_k_steps(cls_name, N):
    """vf_env_rollout_fused: K steps inside one launch, agents held in registers between the steps -- bit-identical to K
    step() calls through auto-resets (ring / racing / spawn state goes through memory inside the launch), for row counts
    whose per-step output rows are and are not 16-byte aligned, and a second rollout continues where per-step calls would"""
    K = 40
    A = actions(N, K)
    ref, env = make(cls_name, N), make(cls_name, N)
    outs = [ref.step(A[k]) for k in range(K)]
    obs, reward, done = env.step_n(A, fused=True)
    assert bool(done.any())
    for k in range(K):
        same(obs[k], outs[k][0]["state"], f"obs @ {k}")
        same(reward[k], outs[k][1], f"reward @ {k}")
        same(done[k], outs[k][2], f"done @ {k}")
    same(env.state_slab, ref.state_slab, "slab after the fused rollout")
    same(env._ep_return, ref._ep_return, "episode returns")
    same(env._terminal_obs, ref._terminal_obs, "terminal observations")
    if cls_name == "RacingEnv":
        same(env.get_observation()["gate"], ref.get_observation()["gate"], "gate observation")
    B = actions(N, 7, seed=1)
    outs2 = [ref.step(B[k]) for k in range(7)]
    obs2, _, _ = env.step_n(B, fused=True)
    same(obs2[6], outs2[-1][0]["state"], "second fused rollout")
    x = actions(N, 1, seed=2)[0]
    same(env.step(x)[0]["state"], ref.step(x)[0]["state"], "step() after fused rollouts (ring phase, counters)")
    same(env.state_slab, ref.state_slab, "slab")


def test_fused_rollout_rk4_drag_randomisation():
    import visfly_amd.envs as E
    N, K = 2048, 30
    dyn = dict(DYN, integrator="rk4", drag_random=0.1)
    mk = lambda: E.NavigationEnv(num_agent_per_scene=N, seed=3, dynamics_kwargs=dict(dyn), device="cuda:0", tensor_output=True,
                                 max_episode_steps=9)
    ref, env = mk(), mk()
    ref.reset(), env.reset()
    A = actions(N, K)
    outs = [ref.step(A[k]) for k in range(K)]
    obs, reward, done = env.step_n(A, fused=True)
    for k in range(K):
        same(obs[k], outs[k][0]["state"], f"obs @ {k}")
        same(reward[k], outs[k][1], f"reward @ {k}")
    same(env.state_slab, ref.state_slab, "slab (re-drawn per-agent drag granules included)")


def test_step_n_is_test_keeps_done_agents():
    N, K = 512, 30
    A = actions(N, K)
    ref, env = make("HoverEnv", N), make("HoverEnv", N)
    outs = [ref.step(A[k], is_test=True) for k in range(K)]
    obs, reward, done = env.step_n(A, is_test=True)
    for k in (0, 11, 12, K - 1):
        same(obs[k], outs[k][0]["state"], f"obs @ {k}")
        same(done[k], outs[k][2], f"done @ {k}")
    same(env.state_slab, ref.state_slab, "slab")


def test_graph_replay_equals_k_steps():
    N, K = 3000, 24
    ref, env = make("NavigationEnv", N), make("NavigationEnv", N)
    buf = torch.empty((K, N, 4), device="cuda")
    for rnd in range(3):                      # the SAME captured graph replays with refilled actions
        A = actions(N, K, seed=rnd)
        buf.copy_(A)
        outs = [ref.step(A[k]) for k in range(K)]
        obs, reward, done = env.step_n(buf, graph=True)
        for k in range(K):
            same(obs[k], outs[k][0]["state"], f"round {rnd} obs @ {k}")
            same(reward[k], outs[k][1], f"round {rnd} reward @ {k}")
            same(done[k], outs[k][2], f"round {rnd} done @ {k}")
        same(env.state_slab, ref.state_slab, f"round {rnd} slab")
    assert len(env._rollouts[K]["graphs"]) == 1          # K is a multiple of the 3-slot delay ring: one graph serves every replay
    env.close()


def test_graph_replay_ring_phases_and_interleaved_steps():
    """K not a multiple of delay_steps: the ring phase at the start of a replay cycles, one graph per phase is captured;
    single step() calls in between shift the phase too"""
    from visfly_amd import _lib
    N, K = 1000, 20
    ref, env = make("HoverEnv", N), make("HoverEnv", N)
    buf = torch.empty((K, N, 4), device="cuda")
    for rnd in range(5):
        A = actions(N, K, seed=10 + rnd)
        buf.copy_(A)
        outs = [ref.step(A[k]) for k in range(K)]
        obs, reward, done = env.step_n(buf, graph=True)
        for k in (0, 1, 2, K - 1):
            same(obs[k], outs[k][0]["state"], f"round {rnd} obs @ {k}")
        same(env.state_slab, ref.state_slab, f"round {rnd} slab")
        if rnd == 2:
            extra = actions(N, 1, seed=99)[0]
            same(env.step(extra)[0]["state"], ref.step(extra)[0]["state"], "interleaved step()")
    assert len(env._rollouts[K]["graphs"]) == 3
    L = _lib.lib()
    phase = int(L.vf_env_ring_phase(env._h))
    wrong = next(g for key, g in env._rollouts[K]["graphs"].items() if key[2] != phase)
    assert L.vf_env_graph_launch(wrong, _lib.current_stream(env.device)) == -3 and b"phase" in L.vf_last_error()
    same(env.state_slab, ref.state_slab, "a refused replay leaves the env untouched")


def test_output_ring_equals_fresh_tensors():
    N, K, R = 2049, 30, 3
    A = actions(N, K)
    ref, env = make("RacingEnv", N), make("RacingEnv", N, out_buffers=R)
    held = []
    for k in range(K):
        o0, r0, d0, i0 = ref.step(A[k])
        o1, r1, d1, i1 = env.step(A[k])
        same(o1["state"], o0["state"], f"obs @ {k}")
        same(o1["gate"], o0["gate"], f"gate @ {k}")
        same(r1, r0, f"reward @ {k}")
        same(d1, d0, f"done @ {k}")
        idx = torch.nonzero(d0).flatten().cpu().numpy()
        if len(idx):
            j = int(idx[0])
            assert i1[j]["episode"]["l"] == i0[j]["episode"]["l"] and i1[j]["TimeLimit.truncated"] == i0[j]["TimeLimit.truncated"]
            assert i1[j]["episode"]["extra"]["past_gate"] == i0[j]["episode"]["extra"]["past_gate"]
        held.append((o1["state"], o0["state"].clone()))
        if k >= R - 1:                         # what step t returned is still intact R-1 steps later
            same(held[k - (R - 1)][0], held[k - (R - 1)][1], f"slot of step {k - (R - 1)} still valid at step {k}")
    assert held[0][0].data_ptr() == held[R][0].data_ptr(), "ring re-uses its buffers"
    with pytest.raises(ValueError):
        make("HoverEnv", 64, out_buffers=1)


def test_step_n_rejects_replay_and_tape():
    from visfly_amd._lib import VisflyError
    env = make("HoverEnv", 128, spawn="replay", replay_trig="cr")
    with pytest.raises(VisflyError):
        env.step_n(actions(128, 2))
    env2 = make("HoverEnv", 128, requires_grad=True)
    with pytest.raises(VisflyError):
        env2.step_n(actions(128, 2))
    env3 = make("HoverEnv", 128)
    with pytest.raises(ValueError):
        env3.step_n(actions(64, 2))


def test_step_n_c_abi_strides_and_errors():
    """direct C-ABI call: stride 0 = one constant action / outputs that keep only the last step"""
    from visfly_amd import _lib
    L = _lib.lib()
    N, K = 640, 9
    ref, env = make("HoverEnv", N), make("HoverEnv", N)
    a = actions(N, 1)[0]
    for _ in range(K):
        o, r, d, _i = ref.step(a)
    obs, rew, done = torch.empty((N, 13), device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, dtype=torch.bool, device="cuda")
    ro = _lib.EnvRollout()
    ro.out = env._out(obs, rew, done)
    ro.actions, ro.K, ro.auto_reset = a.data_ptr(), K, 1          # every stride 0
    _lib.check(L.vf_env_step_n(env._h, C.byref(ro), _lib.current_stream(env.device)))
    same(obs, o["state"], "last obs")
    same(rew, r, "last reward")
    same(env.state_slab, ref.state_slab, "slab")
    ro.K = 0
    assert L.vf_env_step_n(env._h, C.byref(ro), None) == -1 and b"K must be > 0" in L.vf_last_error()
    ro.K, ro.actions = 2, None
    assert L.vf_env_step_n(env._h, C.byref(ro), None) == -1
    g = _lib._vp()
    assert L.vf_env_graph_create(env._h, C.byref(ro), C.byref(g)) == -1
    assert L.vf_env_graph_launch(None, None) == -1


@pytest.mark.parametrize("cls_name", ["HoverEnv", "NavigationEnv"])
def test_export_pose_matches_full_state(cls_name):
    N = 1500
    env = make(cls_name, N)
    A = actions(N, 20)
    pose = None
    saw_reset = False
    for k in range(20):
        obs, reward, done, info = env.step(A[k])
        saw_reset |= bool(done.any())
        pose = env.export_pose(pose)
        fs = env.full_state
        same(pose["position"], fs[:, 0:3].contiguous(), f"position @ {k}")
        same(pose["rotation"], fs[:, 3:7].contiguous(), f"rotation @ {k}")
        same(pose["velocity"], fs[:, 7:10].contiguous(), f"velocity @ {k}")
        same(pose["angular_velocity"], fs[:, 10:13].contiguous(), f"angular velocity @ {k}")
        same(pose["position"], obs["state"][:, 0:3].contiguous(), "pose == returned observation (post auto-reset)")
        same(pose["rotation"], env.envs.dynamics.quaternion.contiguous(), "Dynamics.quaternion")
    assert saw_reset

    class Scene:                               # the stub an external renderer plugs in
        def set_pose(self, position, rotation, velocity=None):
            self.got = (position, rotation, velocity)

    sc = Scene()
    p = env.export_pose()
    sc.set_pose(p["position"], p["rotation"], p["velocity"])
    assert sc.got[1].shape == (N, 4)


def test_export_pose_wind_and_partial_outputs():
    from visfly_amd import _lib
    import visfly_amd.envs as E
    N = 300
    env = E.HoverEnv(num_agent_per_scene=N, dynamics_kwargs=dict(DYN, wind_settings=(0.5, -0.25, 0.125)), device="cuda:0",
                     tensor_output=True)
    env.reset()
    obs, _r, _d, _i = env.step(actions(N, 1)[0])
    pos, vel = torch.empty((N, 3), device="cuda"), torch.empty((N, 3), device="cuda")
    _lib.check(_lib.lib().vf_env_export_pose(env._h, pos.data_ptr(), None, vel.data_ptr(), None, _lib.current_stream(env.device)))
    same(pos, env.position.contiguous(), "position only")
    same(vel, obs["state"][:, 7:10].contiguous(), "velocity includes the wind (dynamics.py:751-752)")
    same(vel, env.velocity.contiguous(), "Dynamics.velocity")
    assert _lib.lib().vf_env_export_pose(None, None, None, None, None, None) == -1


@pytest.mark.parametrize("kind", ["hover", "nav_dr", "racing", "hover_many"])
def test_prefetched_respawn_is_bit_identical(kind):
    """spawn_prefetch (helper blocks of the step launch draw every agent's NEXT re-spawn state ahead of time into the slab; a
    wave that ends an episode only loads it) against the in-place draw: same Philox keys, so every output of every step --
    observations after auto-reset, rewards, done flags, terminal rows, the full state -- is bit-identical; short episodes and
    U(-1,1) actions, so that re-spawns happen in every step, incl. episodes of length one (which fall back to the in-place draw)"""
    import visfly_amd.envs as E
    N, steps = 65536 + 192, 70
    if kind == "hover_many":     # more than two waves per SIMD: the kernel variant whose ending lanes load the copy lazily (k_env_step LAZY_SLOT)
        N, steps = 2 * 65536 + 320, 24
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 0.6], "half": [1., 1., 0.58]},
                                                                "orientation": {"mean": [0., 0., 0.], "half": [0.3, 0.3, 3.0]},
                                                                "velocity": {"mean": [0., 0., 0.], "half": [1., 1., 1.]}}]}}
    dkw = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
    cls, kw = E.HoverEnv, dict(random_kwargs=spawn)
    if kind == "nav_dr":
        cls, dkw = E.NavigationEnv, dict(dkw, drag_random=0.1)
    elif kind == "racing":
        cls, kw, dkw = E.RacingEnv, {}, dict(dkw, action_type="thrust")
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = torch.rand((steps, N, 4), device="cuda:0", generator=g) * 2 - 1
    outs = []
    for pf in (True, False):
        env = cls(num_agent_per_scene=N, seed=11, dynamics_kwargs=dict(dkw), device="cuda:0", max_episode_steps=7, tensor_output=True,
                  spawn_prefetch=pf, **kw)
        assert bool(env._ecfg.spawn_prefetch) == pf
        env.reset()
        rec = []
        for t in range(steps):
            obs, r, d, _ = env.step(acts[t])
            rec.append((obs["state"].clone(), r.clone(), d.clone(), env._terminal_obs.clone(), env._ep_return.clone()))
        rec.append((env.full_state.clone(),))
        # step_n (the launch loop in C) takes the same path
        o2, r2, d2 = env.step_n(acts[:16].contiguous())
        rec.append((o2.clone(), r2.clone(), d2.clone(), env.full_state.clone()))
        outs.append(rec)
        env.close()
    n_done = 0
    for t, (a, b) in enumerate(zip(*outs)):
        for x, y in zip(a, b):
            if x.dtype == torch.bool or x.dtype == torch.uint8:
                assert torch.equal(x, y), f"step {t}"
            else:
                assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f"step {t}"
        if t < steps:
            n_done += int(a[2].sum())
    assert n_done > N * steps // 8           # every agent ends an episode at least every 7 steps


def test_compacted_done_list_matches_the_done_flags():
    """vf_env_out.done_list / done_count: the unordered set of indices equals where(done), through resets"""
    env = make("HoverEnv", 70000)
    env.enable_done_list()
    A = actions(70000, 30, seed=5, wide=1.0)
    seen = 0
    for t in range(30):
        _, _, done, _ = env.step(A[t])
        idx = env.done_indices()
        want = torch.where(done)[0].to(torch.int32)
        assert idx.numel() == want.numel() and torch.equal(torch.sort(idx).values, want), f"step {t}"
        seen += int(want.numel())
    assert seen > 70000
    ref = make("HoverEnv", 70000)
    for t in range(30):
        ref.step(A[t])
    same(env.state_slab, ref.state_slab, "the list is an extra output: the state does not depend on it")
