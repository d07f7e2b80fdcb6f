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


# Distance of the CR-patched reference (= what the fixtures pin to the bit) to the reference EXACTLY as this torch build runs it
# (MKL sin / cos / acos, its not-correctly-rounded sqrt), max abs over the 13 state components after all 256 steps: the fixtures
# store the unpatched run's final state (`raw_ext_last`).  Measured (oracle/gen_golden.py prints them): bodyrate 5.3e-5, dt=0.005
# 1.1e-4, wide actions 9.2e-5, no ctrl_delay 0, RK4 6.1e-5, thrust 2.5e-4 (0.25 m/s^2-scale motion over 5 s), velocity 6.2e-5,
# position 1.5e-5, wind functions 4.6e-5.  north_star's 1e-5 is below the fp32 noise floor of a 256-step closed loop;
# these bounds are what a regression must not exceed -- asserted by the oracle test on the CPU and by the HIP test on the GPU.
RAW_DRIFT_ABS = {"dyn_bodyrate_euler": 1.0e-4, "dyn_bodyrate_euler_wide": 1.5e-4, "dyn_thrust_euler": 4.0e-4, "dyn_bodyrate_nodelay": 1.0e-6,
                 "dyn_bodyrate_dt005": 2.0e-4, "dyn_bodyrate_rk4": 1.0e-4, "dyn_velocity_euler": 1.0e-4, "dyn_position_euler": 5.0e-5,
                 "dyn_wind_functions": 1.0e-4}
RAW_DRIFT_REL = 3.0e-5        # ... and relative to the largest magnitude the column reaches in the fixture


def assert_close_to_unpatched_reference(name, fx, ext_last, what="HIP"):
    """|state - unpatched reference's state| after the fixture's last step: within the stated bounds"""
    d = np.abs(np.asarray(ext_last)[:, :13] - fx["raw_ext_last"][:, :13])
    scale = np.maximum(np.abs(fx["ext"][..., :13]).reshape(-1, 13).max(0), 1e-3)
    rel = float((d.max(0) / scale).max())
    assert d.max() <= RAW_DRIFT_ABS[name], f"{name}: |{what} - unpatched reference| = {d.max():.3e} > {RAW_DRIFT_ABS[name]:.1e}"
    assert rel <= RAW_DRIFT_REL, f"{name}: |{what} - unpatched reference| = {rel:.3e} of the column scale > {RAW_DRIFT_REL:.1e}"
    return float(d.max()), rel


def assert_env_trace_close_to_unpatched_reference(name, fx, reward, done):
    """env fixtures also store the UNPATCHED reference's reward / done of every step (`raw_reward`, `raw_done`): the done flags of all
    256 steps must be identical (measured: 0 mismatches in all eight fixtures) and the rewards within 5e-7 (measured <= 1.4e-7)"""
    reward, done = np.asarray(reward), np.asarray(done).astype(np.uint8)
    assert reward.shape == fx["raw_reward"].shape
    assert np.array_equal(done, fx["raw_done"]), f"{name}: done flags differ from the unpatched reference's in {int((done != fx['raw_done']).sum())} places"
    err = float(np.abs(reward - fx["raw_reward"]).max())
    assert err <= 5e-7, f"{name}: |reward - unpatched reference's| = {err:.3e}"
    return err


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
