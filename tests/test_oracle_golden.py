"""Pins the CPU oracle (oracle/vf_oracle.c) bit-for-bit against golden vectors generated
from the imported reference (oracle/gen_golden.py, CR-sqrt oracle)."""
import numpy as np
import pytest

import oracle
from _golden import GEOMETRIC, assert_bits_equal, assert_geometric_close, bits, consts_of, decode_actions, load

DYN = ["dyn_bodyrate_euler", "dyn_bodyrate_euler_wide", "dyn_thrust_euler", "dyn_bodyrate_nodelay",
       "dyn_bodyrate_dt005", "dyn_bodyrate_rk4"]


@pytest.mark.parametrize("name", GEOMETRIC)
def test_geometric_controller_vs_reference(name):
    """velocity / position action types (dynamics.py:414-496): tolerance-level pin (torch's sin / cos are closed-source
    MKL VML, see _golden.assert_geometric_close); after ONE control step >= 98 % of all state words are still
    bit-identical, which pins the operation order of the restated controller and the SLEEF atan2 restatement."""
    fx = load(name)
    acts = decode_actions(fx)
    N = fx["fs0"].shape[0]
    od = oracle.OracleDynamics(consts_of(fx), N)
    od.set_full_state(fx["fs0"])
    cps = list(fx["checkpoints"])
    for k in range(acts.shape[0]):
        obs = od.step(acts[k])
        if (k + 1) in cps:
            j = cps.index(k + 1)
            assert_geometric_close(od.extend_state, fx["ext"], fx["ext"][j], f"{name} extend_state @ step {k + 1}")
            assert_geometric_close(obs, fx["obs"], fx["obs"][j], f"{name} obs @ step {k + 1}")
            if k == 0:
                same = (bits(od.extend_state) == bits(fx["ext"][j])).mean()
                assert same >= 0.98, same


@pytest.mark.parametrize("name", DYN)
def test_dyn_step_bit_exact(name):
    fx = load(name)
    acts = decode_actions(fx)
    N = fx["fs0"].shape[0]
    od = oracle.OracleDynamics(consts_of(fx), N)
    od.set_full_state(fx["fs0"])
    cps = list(fx["checkpoints"])
    for k in range(acts.shape[0]):
        obs = od.step(acts[k])
        if (k + 1) in cps:
            j = cps.index(k + 1)
            assert_bits_equal(od.extend_state, fx["ext"][j], f"{name} extend_state @ step {k + 1}")
            assert_bits_equal(obs, fx["obs"][j], f"{name} obs @ step {k + 1}")
