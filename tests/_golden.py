"""helpers shared by the parity tests: fixture loading + the int8 action decode"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def consts_of(fx):
    return {k[2:]: fx[k] for k in fx if k.startswith("c_")}


def decode_actions(fx):
    """int8 fixture -> fp32 actions, identical to oracle/gen_golden.py::decode_actions"""
    a = fx["actions_q"].astype(np.float32) * np.float32(float(fx["scale"]) / 127.0) + fx["hover"].astype(np.float32)
    return np.clip(a, np.float32(-1), np.float32(1)).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ne = bits(a) != bits(b)
    # +0 / -0 are distinct bit patterns; report them explicitly if that's the only difference
    if ne.any():
        idx = np.argwhere(ne)[0]
        raise AssertionError(f"{what}: {int(ne.sum())} of {a.size} fp32 words differ; first at {tuple(idx)}: "
                             f"{a[tuple(idx)]!r} vs {b[tuple(idx)]!r}; max|diff|={np.nanmax(np.abs(a - b)):.3e}")


# velocity / position action types.  Their controller evaluates sin / cos (torch: closed-source MKL VML); the
# fixtures are generated with those patched to the fp64 result rounded once (oracle/gen_golden.py::use_cr_trig), which the
# oracle and the kernels reproduce in their "cr" transcendental mode -> held to the bit like every other fixture
GEOMETRIC = ["dyn_velocity_euler", "dyn_position_euler"]


# constructor kwargs of the env fixtures (same as oracle/gen_golden.py::ENV_CASES)
ENV_DYN = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
RACING_DYN = dict(action_type="thrust", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
_NAV_CLOSE_SPAWN = {"state_generator": {
    "class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.5, 1., 0.5]},
                                    "orientation": {"mean": [0., 0., 0.], "half": [0.2, 0.2, 1.0]},
                                    "velocity": {"mean": [1., 0., 0.], "half": [1., .5, .5]}}]}}
ENV_KW = {
    "env_hover2": dict(max_episode_steps=64, random_kwargs={"state_generator": {"class": "Uniform", "kwargs": [
        {"position": {"mean": [1., 0., 1.5], "half": [1.0, 1.0, 0.5]}}]}}),
    "env_nav2": dict(max_episode_steps=96, target=[2.5, 0., 1.5], random_kwargs=_NAV_CLOSE_SPAWN),
    "env_racing": dict(max_episode_steps=48),
    "env_racing2": dict(max_episode_steps=48),
    "env_hover": dict(max_episode_steps=64),
    "env_hover_256": dict(max_episode_steps=256),
    "env_nav": dict(max_episode_steps=64, random_kwargs={"state_generator": {"class": "Uniform", "kwargs": [
        {"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}),
    "env_nav_close": dict(max_episode_steps=96, target=[2.5, 0., 1.5], random_kwargs={"state_generator": {
        "class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0.5, 1., 0.5]},
                                        "orientation": {"mean": [0., 0., 0.], "half": [0.2, 0.2, 1.0]},
                                        "velocity": {"mean": [1., 0., 0.], "half": [1., .5, .5]}}]}}),
}
