"""Pins the CPU oracle (oracle/vf_oracle.c) bit-for-bit against golden vectors generated
from the imported reference (oracle/gen_golden.py, CR-sqrt oracle)."""
import numpy as np
import pytest

import oracle
from _golden import GEOMETRIC, RAW_DRIFT_ABS, assert_bits_equal, assert_close_to_unpatched_reference, consts_of, decode_actions, load

# incl. the velocity / position action types (dynamics.py:414-496): since r03 their fixtures come from the CR-trig reference
# (sin / cos = fp64 result rounded once) and are held to the bit for all 256 steps, like bodyrate / thrust
DYN = ["dyn_bodyrate_euler", "dyn_bodyrate_euler_wide", "dyn_thrust_euler", "dyn_bodyrate_nodelay",
       "dyn_bodyrate_dt005", "dyn_bodyrate_rk4"] + GEOMETRIC


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


@pytest.mark.parametrize("name", sorted(RAW_DRIFT_ABS))
def test_cr_oracle_stays_close_to_the_unpatched_reference(name):
    """the bit-level pins are against the CR-patched reference (correctly rounded sqrt / sin / cos / acos); this bounds what the
    patch costs against the reference as torch runs it, for every dynamics fixture (ADVICE r03 / VERDICT r03 item 7)"""
    fx = load(name)
    assert_close_to_unpatched_reference(name, fx, fx["ext"][-1], "CR fixture")


@pytest.mark.parametrize("name", GEOMETRIC)
def test_sleef_mode_stays_close_to_the_unpatched_reference(name):
    """transcendentals="sleef" (no fp64 arithmetic) against the unpatched reference: the same bounds (measured 1.2e-5 abs)"""
    fx = load(name)
    consts = dict(consts_of(fx), trig_mode=np.int32(0))
    acts = decode_actions(fx)
    od = oracle.OracleDynamics(consts, fx["fs0"].shape[0])
    od.set_full_state(fx["fs0"])
    for k in range(acts.shape[0]):
        od.step(acts[k])
    assert_close_to_unpatched_reference(name, fx, od.extend_state, "sleef-mode oracle")
