"""Host-side constant derivation (visfly_amd/constants.py) reproduces the reference's bits
as captured in the golden fixtures (derived constants are part of the contract)."""
import numpy as np
import pytest

from _golden import consts_of, load
from visfly_amd.constants import derive_constants
from visfly_amd.drone_params import available, load_params

CASES = {
    "dyn_bodyrate_euler": dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True),
    "dyn_thrust_euler": dict(action_type="thrust", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True),
    "dyn_bodyrate_nodelay": dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=False, comm_delay=0.0),
    "dyn_bodyrate_dt005": dict(action_type="bodyrate", dt=0.005, ctrl_dt=0.03, ctrl_delay=True,
                               wind_settings=[0.5, -0.25, 0.125]),
    "dyn_bodyrate_rk4": dict(action_type="bodyrate", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True, integrator="rk4"),
    "dyn_velocity_euler": dict(action_type="velocity", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True),
    "dyn_position_euler": dict(action_type="position", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True),
}


@pytest.mark.parametrize("name", list(CASES))
def test_constant_bits(name):
    want = consts_of(load(name))
    got = derive_constants(**CASES[name])
    # fixtures generated before the geometric-controller constants existed lack those keys
    from visfly_amd._lib import GEOMETRIC_FIELDS
    assert set(want) <= set(got) and set(got) - set(want) <= set(GEOMETRIC_FIELDS)
    if "velocity" in name or "position" in name:
        assert set(got) == set(want)
    for k in want:
        a, b = np.asarray(got[k]), np.asarray(want[k])
        assert a.dtype == b.dtype and a.shape == b.shape, (k, a.dtype, b.dtype, a.shape, b.shape)
        assert a.tobytes() == b.tobytes(), (k, a, b)


def test_bad_ctrl_dt_raises():
    with pytest.raises(ValueError):  # dynamics.py:71-72
        derive_constants(dt=0.003, ctrl_dt=0.02)


def test_airframes():
    assert "drone_state" in available()
    for name in ("drone_state", "drone_d435i_jetson_orin_nx", "drone_d435i_jetson_orin_nx_fast"):
        derive_constants(cfg=name)
    with pytest.raises(KeyError):  # these airframes lack max_acc / THRUST_PID in the reference too (SURVEY C-7)
        derive_constants(cfg="drone_d435i")
    assert load_params("drone_state")["BODYRATE_PID"]["p"][1][1] == 50.6
