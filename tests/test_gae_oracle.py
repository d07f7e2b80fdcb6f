"""GAE: the oracle (oracle/vf_oracle.c::vfo_gae) against a line-by-line numpy transcription of the
reference's recurrence (utils/algorithms/common.py:119-132 == SB3 2.2.1
RolloutBuffer.compute_returns_and_advantage) and a hand-computed T=3 case.  The reference's own
tests pin nothing here and SB3 is not installed: parity unpinned by reference tests."""
import numpy as np

import oracle


def gae_numpy(rewards, values, episode_starts, last_values, dones, gamma, gae_lambda):
    """transcription of common.py:119-132 with numpy fp32 arrays and python-float gamma/lambda"""
    T = rewards.shape[0]
    adv = np.zeros_like(rewards)
    last_gae_lam = 0
    for step in reversed(range(T)):
        if step == T - 1:
            next_non_terminal = 1.0 - dones
            next_values = last_values
        else:
            next_non_terminal = 1.0 - episode_starts[step + 1]
            next_values = values[step + 1]
        delta = rewards[step] + gamma * next_values * next_non_terminal - values[step]
        last_gae_lam = delta + gamma * gae_lambda * next_non_terminal * last_gae_lam
        adv[step] = last_gae_lam
    return adv, adv + values


def make(T, N, seed):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.normal(size=s).astype(np.float32)
    es = (rng.uniform(size=(T, N)) < 0.1).astype(np.float32)
    return f(T, N), f(T, N), es, f(N), (rng.uniform(size=N) < 0.2).astype(np.float32)


def test_gae_matches_numpy_transcription_bitwise():
    for T, N, seed in [(16, 5, 0), (256, 33, 1), (1, 7, 2)]:
        r, v, es, lv, d = make(T, N, seed)
        a0, r0 = gae_numpy(r, v, es, lv, d, 0.99, 0.95)
        a1, r1 = oracle.gae(r, v, es, lv, d, 0.99, 0.95)
        assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))
        assert np.array_equal(r0.view(np.uint32), r1.view(np.uint32))


def test_gae_hand_computed():
    # T=3, N=1, gamma=0.5, lambda=0.5, no terminations: exact binary fractions
    r = np.array([[1.0], [2.0], [4.0]], np.float32)
    v = np.array([[0.5], [1.0], [2.0]], np.float32)
    es = np.zeros((3, 1), np.float32)
    lv, d = np.array([8.0], np.float32), np.array([0.0], np.float32)
    # delta2 = 4 + .5*8 - 2 = 6 ; A2 = 6
    # delta1 = 2 + .5*2 - 1 = 2 ; A1 = 2 + .25*6 = 3.5
    # delta0 = 1 + .5*1 - .5 = 1 ; A0 = 1 + .25*3.5 = 1.875
    a, ret = oracle.gae(r, v, es, lv, d, 0.5, 0.5)
    assert a[:, 0].tolist() == [1.875, 3.5, 6.0]
    assert ret[:, 0].tolist() == [2.375, 4.5, 8.0]
    # an episode start at t=2 cuts the bootstrap from step 1
    es[2, 0] = 1.0
    a, _ = oracle.gae(r, v, es, lv, d, 0.5, 0.5)
    assert a[:, 0].tolist() == [1 + .25 * 1.0, 2.0 - 1.0, 6.0]


def test_td_lambda_oracle_bit_exact_vs_reference_function():
    """vfo_td_returns against golden vectors produced by the reference's compute_td_returns
    (utils/algorithms/common.py:893-923, imported through the stubs by oracle/gen_golden.py::gen_td)"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _golden import load
    fx = load("td_lambda")
    for tag, ed in (("", None), ("_ep", fx["episode_done"])):
        out = oracle.td_returns(fx["r"], fx["done"], fx["next_value"], ed, float(fx["gamma"]), float(fx["lamda"]))
        assert np.array_equal(out.view(np.uint32), fx["returns" + tag].view(np.uint32))
