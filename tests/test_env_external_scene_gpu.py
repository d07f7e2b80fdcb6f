"""the step split around an external scene manager (step_begin -> scene -> step_finish), the hand-off
DroneEnvsBase.step performs with visual=True (/root/reference/envs/base/droneEnv.py:330-342,364-367,374-379).  No Habitat
scene exists here, so the checks are: (1) with the bounding-box answer (None, or the same closest point computed outside)
the split step equals the fused step bit for bit over whole episodes; (2) with a synthetic scene (a sphere obstacle) the
collision outputs follow the reference's arithmetic on the scene's closest point."""
import pytest
import torch as th

pytestmark = pytest.mark.gpu

from visfly_amd.envs import HoverEnv, NavigationEnv

KW = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
NAV_RK = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.0, 2., 1.]}}]}}


def _mk(cls, n, **kw):
    env = cls(num_agent_per_scene=n, seed=7, visual=False, dynamics_kwargs=dict(KW), device="cuda:0", tensor_output=True,
              max_episode_steps=48, **kw)
    env.reset()
    return env


def _bbox_point(env):
    p = env.position
    lo, hi = th.tensor([-30., -30., 0.], device=p.device), th.tensor([30., 30., 8.], device=p.device)
    v, idx = th.hstack([p - lo, hi - p]).min(dim=1)
    cp = p.clone()
    flat = th.cat([lo, hi])
    cp[th.arange(p.shape[0]), idx % 3] = flat[idx]
    return cp, ((p < lo).any(dim=1) | (p > hi).any(dim=1))


@pytest.mark.parametrize("cls,kw", [(HoverEnv, {}), (NavigationEnv, {"random_kwargs": NAV_RK})])
@pytest.mark.parametrize("answer", ["none", "bbox"])
def test_split_step_equals_fused_step(cls, kw, answer):
    n = 1000
    a, b = _mk(cls, n, **kw), _mk(cls, n, **kw)
    g = th.Generator(device="cuda").manual_seed(3)
    for t in range(120):     # > 2 episodes: truncations, crashes (wide actions) and re-spawns on both sides
        act = (th.rand((n, 4), device="cuda", generator=g) * 2 - 1) * (1.0 if t % 3 else 0.05) + th.tensor([-1 / 3, 0, 0, 0], device="cuda") * (t % 3 == 0)
        act = act.clamp(-1, 1).contiguous()
        o1, r1, d1, _ = a.step(act)
        pose = b.step_begin(act)
        assert set(pose) == {"position", "rotation", "velocity", "angular_velocity"}
        if answer == "bbox":
            cp, oob = _bbox_point(b)
            o2, r2, d2, _ = b.step_finish(cp, oob)
        else:
            o2, r2, d2, _ = b.step_finish()
        assert th.equal(r1, r2) and th.equal(d1, d2), t
        for k in o1.keys():
            assert th.equal(o1[k], o2[k]), (t, k)
        assert th.equal(a.is_collision, b.is_collision) and th.equal(a.collision_dis, b.collision_dis)
    assert th.equal(a.state, b.state)


def test_scene_answer_drives_collision_and_done():
    n = 512
    env = _mk(HoverEnv, n)
    centre, radius = th.tensor([1.0, 0.0, 1.5], device="cuda"), 0.6          # a ball in the middle of the spawn box
    hover = th.tensor([-1 / 3, 0, 0, 0], device="cuda").repeat(n, 1).contiguous()
    hits = 0
    for t in range(20):
        pose = env.step_begin(hover)
        p = pose["position"]
        d = p - centre
        dist = d.norm(dim=1, keepdim=True).clamp_min(1e-6)
        cp = centre + d / dist * radius                                       # closest point of the ball's surface
        obs, reward, done, info = env.step_finish(cp, th.zeros(n, dtype=th.bool, device="cuda"))
        vec = cp - p
        dis = (vec - 0).norm(dim=1)                                           # droneEnv.py:365-366
        hit = dis < 0.1                                                       # uav_radius, :367
        # is_collision_reset is the default: a hit ends the episode in this step (droneGymEnv.py:189-190)
        assert bool((done >= hit).all())                                      # every agent the ball touched is done
        alive = ~done
        assert th.equal(env.is_collision[alive], hit[alive])
        assert th.allclose(env.collision_dis[alive], dis[alive], rtol=0, atol=1e-6)
        assert th.allclose(env.collision_point[alive], cp[alive])
        hits += int(hit.sum())
    assert hits > 0                                                           # the ball was actually hit


def test_half_step_is_guarded():
    from visfly_amd.envs.base import VisflyError
    env = _mk(HoverEnv, 64)
    a = th.zeros((64, 4), device="cuda")
    with pytest.raises(VisflyError):
        env.step_finish()
    env.step_begin(a)
    with pytest.raises(VisflyError):
        env.step(a)
    env.step_finish()
    env.step(a)
