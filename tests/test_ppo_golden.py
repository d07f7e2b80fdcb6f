"""PPO-side goldens (tests/golden/ppo_nav*.npz, oracle/gen_golden.py::gen_ppo): advantages / returns produced by the
reference's own RolloutBuffer.compute_returns_and_advantage, and the loss scalars + flat parameter gradient of one
PPO.train minibatch (utils/algorithms/PPO.py:210-263) through a network assembled from the reference's own
StateTargetExtractor / create_mlp modules.  CPU: the oracle's GAE is bit-identical to the reference's.  GPU: the chain
vf_gae -> vf_adv_normalize_segments -> vf_ppo_update -> vf_mlp_weight_grad reproduces the frozen numbers."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["ppo_nav", "ppo_nav_vclip"]


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.mark.parametrize("name", CASES)
def test_oracle_gae_is_the_references_gae(name):
    fx = load(name)
    a, r = oracle.gae(fx["rewards"], fx["values"], fx["episode_starts"], fx["last_values"], fx["dones"],
                      float(fx["gamma"]), float(fx["gae_lambda"]))
    assert np.array_equal(a.view(np.uint32), fx["advantages"].view(np.uint32)), np.abs(a - fx["advantages"]).max()
    assert np.array_equal(r.view(np.uint32), fx["returns"].view(np.uint32))
    assert fx["params"].size == 43977 and fx["grad"].size == 43977          # StateTarget MLP of BASELINE configs[3]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_ppo_chain_matches_reference_golden(name):
    import torch
    from visfly_amd import _lib
    from visfly_amd.ppo import MlpPolicy
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = lambda: _lib.current_stream(dev)
    fx = load(name)
    T, N = fx["rewards"].shape
    M = T * N
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # ---- vf_gae: bit-identical to the reference's RolloutBuffer ----
    adv, ret = torch.empty((T, N), device=dev), torch.empty((T, N), device=dev)
    rw, vl, es, lv, dn = (d(fx[k]) for k in ("rewards", "values", "episode_starts", "last_values", "dones"))
    _lib.check(L.vf_gae(rw.data_ptr(), vl.data_ptr(), es.data_ptr(), lv.data_ptr(), dn.data_ptr(), adv.data_ptr(), ret.data_ptr(),
                        T, N, float(fx["gamma"]), float(fx["gae_lambda"]), st()))
    assert np.array_equal(adv.cpu().numpy().view(np.uint32), fx["advantages"].view(np.uint32))
    assert np.array_equal(ret.cpu().numpy().view(np.uint32), fx["returns"].view(np.uint32))
    # ---- advantage normalisation of the (single, full) minibatch: PPO.py:215-220 ----
    advn = torch.empty(M, device=dev)
    sums = torch.empty((1, 2), dtype=torch.float64, device=dev)
    _lib.check(L.vf_adv_normalize_segments(adv.data_ptr(), advn.data_ptr(), 1, M, M, sums.data_ptr(), 2, st()))
    want = fx["adv_normalized"]
    assert np.abs(advn.cpu().numpy() - want).max() <= 4e-6 * np.abs(want).max()      # fp64 two-pass sums vs torch's fp32 mean / std
    # ---- the policy with the fixture's weights ----
    pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], dev)
    assert pol.n_params == fx["params"].size
    pol.flat.copy_(d(fx["params"]))
    pol.mark_updated()
    obs = {"state": d(fx["obs_state"].reshape(M, 13)), "target": d(fx["obs_target"].reshape(M, 3))}
    mean, value = pol.forward(obs, save_activations=False)
    assert np.abs(mean.cpu().numpy() - fx["mean"]).max() <= 2e-5 and np.abs(value.cpu().numpy().reshape(-1) - fx["value"]).max() <= 2e-5
    # ---- one fused PPO minibatch step: forward + loss + reverse chain + weight gradients ----
    vclip = float(fx["clip_range_vf"])
    old_v = d(fx["values"].reshape(M))
    stats, scratch = torch.zeros(16, device=dev), torch.zeros(16 * 1024 + 4096, device=dev)
    cfg = _lib.PpoLossCfg(float(fx["clip_range"]), float(fx["ent_coef"]), float(fx["vf_coef"]), 1.0 / M,
                          pol.grad.data_ptr() + 4 * pol.log_std_off, None,
                          old_v.data_ptr() if vclip > 0 else None, vclip if vclip > 0 else 0.0, 0)
    act, olp, advn_ref, ret_flat = d(fx["actions"]), d(fx["old_log_prob"]), d(want), ret.reshape(M).contiguous()
    res = pol.ppo_update(obs, act, olp, advn_ref, ret_flat, cfg, stats, scratch)
    assert res is True, "the StateTarget network must run on the register-chained kernels"
    s = (stats[:5] / M).cpu().numpy()
    gold = np.array([fx["policy_loss"], fx["value_loss"], fx["entropy_loss"], fx["approx_kl"], fx["clip_fraction"]], np.float32)
    assert np.allclose(s, gold, rtol=3e-5, atol=2e-6), (s, gold)
    g = pol.grad.cpu().numpy()
    scale = np.abs(fx["grad"]).max()
    assert np.abs(g - fx["grad"]).max() <= 3e-5 * scale, (np.abs(g - fx["grad"]).max(), scale)
    # per-layer: every block of the flat gradient individually close (a wrong layer cannot hide behind a large one)
    for ly in pol.layers:
        for off, n in ((ly.w_off, ly.K * ly.No), (ly.b_off, ly.No)):
            ref = fx["grad"][off:off + n]
            assert np.abs(g[off:off + n] - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-7), (ly.dst, off)
    assert np.allclose(g[pol.log_std_off:], fx["grad"][pol.log_std_off:], rtol=1e-4, atol=1e-7)
    # ---- the stand-alone loss kernel + block-tile backward give the same answer (fallback path) ----
    d_mean, d_value = torch.empty((M, 4), device=dev), torch.empty(M, device=dev)
    stats2 = torch.zeros(16, device=dev)
    mean, value = pol.forward(obs)
    _lib.check(L.vf_ppo_loss(mean.data_ptr(), value.data_ptr(), pol.log_std.data_ptr(), act.data_ptr(), olp.data_ptr(),
                             advn_ref.data_ptr(), ret_flat.data_ptr(), d_mean.data_ptr(), d_value.data_ptr(), stats2.data_ptr(), M,
                             C.byref(cfg), scratch.data_ptr(), st()))
    assert np.allclose((stats2[:5] / M).cpu().numpy(), gold, rtol=3e-5, atol=2e-6)
