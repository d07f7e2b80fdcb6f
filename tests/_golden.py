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


GEOMETRIC = ["dyn_velocity_euler", "dyn_position_euler"]


def assert_geometric_close(got, want_all, want, what=""):
    """velocity / position action types: the controller evaluates sin / cos / atan2.  torch.atan2 is SLEEF's
    atan2f_u10 and is restated bit for bit; torch.sin / torch.cos run Intel MKL VML (closed source), so the oracle and
    the kernels use SLEEF's sinf_u10 / cosf_u10 (restated bit for bit, the closest published algorithms), which differ
    from MKL by one ulp in ~2 % of the calls (oracle/vf_sleef.h, tests/test_sleef_restatement.py).  After ONE control
    step >= 98 % of the state words are bit-identical; the closed attitude loop then amplifies the one-ulp
    differences.  Tolerance per extend_state column: |diff| <= 5e-5 * max|column| over the 256-step fixture -- looser
    than north_star's 1e-5, which these two action types cannot meet without MKL's algorithm."""
    got = np.asarray(got, np.float32)
    scale = np.abs(want_all).reshape(-1, want_all.shape[-1]).max(0)
    lim = 5e-5 * np.maximum(scale, 1e-3)
    d = np.abs(got - want)
    bad = d > lim
    if bad.any():
        i = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {int(bad.sum())} components beyond tolerance; first at {tuple(i)}: "
                             f"{got[tuple(i)]!r} vs {want[tuple(i)]!r} (limit {lim[i[-1]]:.2e})")
    return float((d / lim).max())


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
