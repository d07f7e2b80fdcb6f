"""PPO inner loop kernels against plain PyTorch fp32 references (and the CPU oracle for GAE).
Floating-point tolerances are stated per test; they cover fp32 summation-order differences only."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def L():
    from visfly_amd import _lib
    return _lib, _lib.lib()


def st():
    return torch.cuda.current_stream().cuda_stream


def test_gae_kernel_bit_exact_vs_oracle():
    _lib, lib = L()
    for T, N in [(16, 5), (256, 4099), (64, 65536)]:
        g = torch.Generator().manual_seed(T + N)
        r, v = torch.randn((T, N), generator=g), torch.randn((T, N), generator=g)
        es = (torch.rand((T, N), generator=g) < 0.05).float()
        lv, d = torch.randn(N, generator=g), (torch.rand(N, generator=g) < 0.2).float()
        dr, dv, des, dlv, dd = (x.to(DEV) for x in (r, v, es, lv, d))
        adv, ret = torch.empty_like(dr), torch.empty_like(dr)
        _lib.check(lib.vf_gae(dr.data_ptr(), dv.data_ptr(), des.data_ptr(), dlv.data_ptr(), dd.data_ptr(), adv.data_ptr(),
                              ret.data_ptr(), T, N, 0.99, 0.95, st()))
        a0, r0 = oracle.gae(r.numpy(), v.numpy(), es.numpy(), lv.numpy(), d.numpy(), 0.99, 0.95)
        assert np.array_equal(adv.cpu().numpy().view(np.uint32), a0.view(np.uint32))
        assert np.array_equal(ret.cpu().numpy().view(np.uint32), r0.view(np.uint32))


def test_td_lambda_kernel_bit_exact_vs_golden_and_oracle():
    from _golden import load
    _lib, lib = L()
    fx = load("td_lambda")
    for tag, ed in (("", None), ("_ep", fx["episode_done"])):
        H, N = fx["r"].shape
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
        r, d, e, nv = t(fx["r"]), t(fx["done"]), t(ed), t(fx["next_value"])
        out = torch.empty((H, N), device=DEV)
        _lib.check(lib.vf_td_returns(r.data_ptr(), d.data_ptr(), None if e is None else e.data_ptr(), nv.data_ptr(),
                                     out.data_ptr(), H, N, float(fx["gamma"]), float(fx["lamda"]), st()))
        assert np.array_equal(out.cpu().numpy().view(np.uint32), fx["returns" + tag].view(np.uint32))
    g = np.random.default_rng(0)
    H, N = 64, 70001
    r, nv = g.normal(size=(H, N)).astype(np.float32), g.normal(size=(H, N)).astype(np.float32)
    d = (g.uniform(size=(H, N)) < 0.05).astype(np.uint8)
    out = torch.empty((H, N), device=DEV)
    rd, dd, nvd = (torch.from_numpy(x).to(DEV) for x in (r, d, nv))
    _lib.check(lib.vf_td_returns(rd.data_ptr(), dd.data_ptr(), None, nvd.data_ptr(), out.data_ptr(), H, N, 0.99, 0.95, st()))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.td_returns(r, d, nv, None, 0.99, 0.95).view(np.uint32))


def test_adv_normalize_vs_torch():
    _lib, lib = L()
    for n in [2, 1000, 25600, 1 << 20]:
        a = torch.randn(n, device=DEV) * 3 + 1.5
        out, scratch = torch.empty_like(a), torch.zeros(4096, device=DEV)
        _lib.check(lib.vf_adv_normalize(a.data_ptr(), out.data_ptr(), n, n, None, scratch.data_ptr(), 2, st()))
        ref = (a - a.mean()) / (a.std() + 1e-8)          # PPO.py:217-220 (unbiased std)
        assert torch.allclose(out, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("K,No", [(13, 128), (128, 64), (3, 128), (64, 64), (64, 4), (64, 1), (128, 128), (7, 33)])
@pytest.mark.parametrize("M", [1, 63, 200, 25600])
def test_linear_layers_vs_torch(K, No, M):
    _lib, lib = L()
    g = torch.Generator(device=DEV).manual_seed(K * 1000 + No + M)
    X = torch.randn((M, K), device=DEV, generator=g)
    W = torch.randn((No, K), device=DEV, generator=g) / np.sqrt(K)
    b = torch.randn(No, device=DEV, generator=g)
    # asymmetric operands (transposes cannot hide), fp64 reference
    Yref = (X.double() @ W.double().T + b.double())
    for relu in (0, 1):
        Y = torch.full((M, No + 3), -7.0, device=DEV)     # ldy > No: columns beyond stay untouched
        _lib.check(lib.vf_linear_fwd(X.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), No + 3, M, K, No, relu, st()))
        ref = Yref.clamp_min(0) if relu else Yref
        assert torch.allclose(Y[:, :No].double(), ref, rtol=1e-5, atol=1e-5 * np.sqrt(K))
        assert (Y[:, No:] == -7.0).all()
    Ypost = Yref.clamp_min(0).float()
    dY = torch.randn((M, No), device=DEV, generator=g)
    dYm = (dY * (Ypost > 0)).double()
    # data gradient with mask, then accumulate
    dX = torch.zeros((M, K), device=DEV)
    _lib.check(lib.vf_linear_bwd_data(dY.data_ptr(), No, Ypost.data_ptr(), No, W.data_ptr(), dX.data_ptr(), K, M, K, No, 0, 1, st()))
    ref = dYm @ W.double()
    assert torch.allclose(dX.double(), ref, rtol=1e-5, atol=1e-5 * np.sqrt(No))
    _lib.check(lib.vf_linear_bwd_data(dY.data_ptr(), No, None, 0, W.data_ptr(), dX.data_ptr(), K, M, K, No, 1, 0, st()))
    assert torch.allclose(dX.double(), ref + dY.double() @ W.double(), rtol=1e-5, atol=2e-5 * np.sqrt(No))
    # weight / bias gradient
    need = int(lib.vf_linear_bwd_scratch_floats(M, K, No))
    scratch = torch.empty(need, device=DEV)
    dW, db = torch.empty((No, K), device=DEV), torch.empty(No, device=DEV)
    _lib.check(lib.vf_linear_bwd_weight(dY.data_ptr(), No, Ypost.data_ptr(), No, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(),
                                        M, K, No, scratch.data_ptr(), 1, st()))
    tol = 2e-5 * np.sqrt(M)
    assert torch.allclose(dW.double(), dYm.T @ X.double(), rtol=1e-5, atol=tol)
    assert torch.allclose(db.double(), dYm.sum(0), rtol=1e-5, atol=tol)


ACTS = {"relu": (1, torch.nn.ReLU), "tanh": (2, torch.nn.Tanh), "elu": (3, torch.nn.ELU), "leaky_relu": (4, torch.nn.LeakyReLU)}


@pytest.mark.parametrize("act", list(ACTS))
@pytest.mark.parametrize("K,No,M", [(13, 128, 200), (128, 64, 25600), (64, 64, 63), (7, 33, 1)])
def test_linear_layers_with_every_activation_vs_torch(K, No, M, act):
    """create_mlp's activation_fn (extractors.py:376-449; aliases policies.py:64-69) in the per-layer kernels: forward act(z), data and
    weight gradients through act'(y) formed from the saved OUTPUT, against torch fp64"""
    _lib, lib = L()
    kind, mod = ACTS[act]
    g = torch.Generator(device=DEV).manual_seed(K * 1000 + No + M + kind)
    X = torch.randn((M, K), device=DEV, generator=g)
    W = (torch.randn((No, K), device=DEV, generator=g) / np.sqrt(K)).requires_grad_(False)
    b = torch.randn(No, device=DEV, generator=g)
    dY = torch.randn((M, No), device=DEV, generator=g)
    Xr, Wr, br = X.double().requires_grad_(True), W.double().requires_grad_(True), b.double().requires_grad_(True)
    Yr = mod()(Xr @ Wr.T + br)
    (Yr * dY.double()).sum().backward()
    Y = torch.empty((M, No), device=DEV)
    _lib.check(lib.vf_linear_fwd(X.data_ptr(), K, W.data_ptr(), b.data_ptr(), Y.data_ptr(), No, M, K, No, kind, st()))
    assert torch.allclose(Y.double(), Yr.detach(), rtol=2e-6, atol=2e-6 * np.sqrt(K))
    dX = torch.zeros((M, K), device=DEV)
    _lib.check(lib.vf_linear_bwd_data(dY.data_ptr(), No, Y.data_ptr(), No, W.data_ptr(), dX.data_ptr(), K, M, K, No, 0, kind, st()))
    assert torch.allclose(dX.double(), Xr.grad, rtol=1e-5, atol=1e-5 * np.sqrt(No))
    scratch = torch.empty(int(lib.vf_linear_bwd_scratch_floats(M, K, No)), device=DEV)
    dW, db = torch.empty((No, K), device=DEV), torch.empty(No, device=DEV)
    _lib.check(lib.vf_linear_bwd_weight(dY.data_ptr(), No, Y.data_ptr(), No, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), M, K, No,
                                        scratch.data_ptr(), kind, st()))
    tol = 2e-5 * np.sqrt(M)
    assert torch.allclose(dW.double(), Wr.grad, rtol=1e-5, atol=tol) and torch.allclose(db.double(), br.grad, rtol=1e-5, atol=tol)


@pytest.mark.parametrize("acts", [("tanh", "relu"), ("elu", "leaky_relu"), ("leaky_relu", "leaky_relu"), ("tanh", "tanh")])
@pytest.mark.parametrize("M", [33, 25600])
def test_policy_with_other_activations_vs_torch(acts, M, monkeypatch):
    """the whole actor-critic with `activation_fn` = Tanh (the reference policy's DEFAULT, policies.py:108) / ELU / LeakyReLU in the trunks
    and / or the extractor MLPs against torch autograd in fp64: (a) the generated chain class of the shape + activations (forward, reverse
    chain + row-slab weight gradients; ("tanh", "relu") and ("elu", "leaky_relu") are pre-built, __graft_entry__.build), (b) the block-tile
    kernels (one launch each way), (c) layer by layer"""
    import warnings
    from visfly_amd.ppo import MlpPolicy
    _lib, lib = L()
    act, ext_act = acts
    dims = {"state": 13, "target": 3}
    prebuilt = acts in (("tanh", "relu"), ("elu", "leaky_relu"))
    if not prebuilt:
        monkeypatch.setenv("VISFLY_AMD_JIT", "0")          # (the other two combinations: the general kernels only)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pol = MlpPolicy(dims, {k: [128, 64] for k in dims}, [64, 64], [64, 64], DEV, seed=5, activation=act, extractor_activation=ext_act)
    assert pol.chain_jit == prebuilt and pol.spec["activation"] == act and pol.spec["extractor_activation"] == ext_act
    g = torch.Generator(device=DEV).manual_seed(M)
    obs = {k: torch.randn((M, d), device=DEV, generator=g) for k, d in dims.items()}
    d_mean, d_value = torch.randn((M, 4), device=DEV, generator=g) / M, torch.randn(M, device=DEV, generator=g) / M
    ref = pol.to_torch().double().to(DEV)
    xs = {k: v.double().requires_grad_(True) for k, v in obs.items()}
    m0, v0 = ref(xs)
    ((m0 * d_mean.double()).sum() + (v0.view(-1) * d_value.double()).sum()).backward()
    gref = ref.flat_grad().to(DEV)
    # ... and what plain fp32 PyTorch makes of the same network on the same device: the yardstick for Tanh (below)
    ref32 = pol.to_torch().to(DEV)
    x32 = {k: v.clone().requires_grad_(True) for k, v in obs.items()}
    m32, v32 = ref32(x32)
    ((m32 * d_mean).sum() + (v32.view(-1) * d_value).sum()).backward()
    g32 = ref32.flat_grad().to(DEV).double()
    sc = max(m0.abs().max().item(), v0.abs().max().item(), 1e-3)
    grads = []
    for fused in ((("chain", True), ("tile", True), ("layers", False)) if prebuilt else (("tile", True), ("layers", False))):
        fused, on = fused
        pol.fused = pol.fused_backward = on
        lib.vf_chain_plugin_set_enabled(1 if fused == "chain" else 0)
        n0 = lib.vf_chain_plugin_launches()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mean, value = pol.forward(obs)
            assert (lib.vf_chain_plugin_launches() > n0) == (fused == "chain"), fused
            assert (mean.double() - m0).abs().max().item() <= 4e-6 * sc and (value.view(-1).double() - v0.view(-1)).abs().max().item() <= 4e-6 * sc
            d_in = pol.backward(d_mean, d_value, None, need_input_grad=True)
        lib.vf_chain_plugin_set_enabled(1)
        # Tanh's derivative from the saved output, 1 - y^2 in fp32 (torch's tanh_backward forms it the same way), cancels where the unit
        # saturates: ~6e-5 relative per term against the fp64 reference, which the other three activations do not have
        # -- and LeakyReLU / ELU pass a gradient through EVERY unit, so a block of 25 600-row sums cancels further below the magnitude of its
        # terms than a ReLU block does.  Every block is held to the ReLU tests' 2e-5 of its scale OR three times the distance of torch's
        # own fp32 autograd from the fp64 reference, whichever is larger
        tol = 2e-5
        for ly in pol.layers:        # per parameter block, each against its own scale (as test_fused_backward_equals_layerwise_and_torch)
            for lo, hi in ((ly.w_off, ly.w_off + ly.K * ly.No), (ly.b_off, ly.b_off + ly.No)):
                bs = max(gref[lo:hi].abs().max().item(), 1e-3 * gref.abs().max().item())
                bound = max(tol * bs, 3.0 * (g32[lo:hi] - gref[lo:hi]).abs().max().item())
                assert (pol.grad[lo:hi].double() - gref[lo:hi]).abs().max().item() <= bound, (fused, ly.dst, bound / bs)
        for k in dims:
            bound = max(tol * max(xs[k].grad.abs().max().item(), 1e-12), 3.0 * (x32[k].grad.double() - xs[k].grad).abs().max().item())
            assert (d_in[k].double() - xs[k].grad).abs().max().item() <= bound, (fused, k)
        grads.append(pol.grad.clone())
    for gx in grads[1:]:
        assert (grads[0] - gx).abs().max().item() <= (3e-4 if "tanh" in acts else 2e-5) * gref.abs().max().item()


@pytest.mark.parametrize("acts", [("tanh", "relu"), ("elu", "leaky_relu")])
@pytest.mark.parametrize("B", [25600, 777])
def test_fused_ppo_step_of_a_tanh_or_elu_class_equals_separate_launches(acts, B):
    """vf_ppo_update on a generated class with Tanh / ELU / LeakyReLU layers (derivatives from the forward's own live tiles) + weight
    gradients vs forward / vf_ppo_loss / backward on the block-tile kernels: as test_fused_ppo_update_equals_separate_launches"""
    from visfly_amd.ppo import MlpPolicy
    _lib, lib = L()
    dims = {"state": 13, "target": 3}
    pol = MlpPolicy(dims, {k: [128, 64] for k in dims}, [64, 64], [64, 64], DEV, seed=5, log_std_init=-0.3, activation=acts[0], extractor_activation=acts[1])
    assert pol.chain_jit
    g = torch.Generator(device=DEV).manual_seed(B)
    obs = {k: torch.randn((B, d), device=DEV, generator=g) for k, d in dims.items()}
    mean, value = pol.forward(obs)
    actions = torch.tanh(mean + 0.7 * torch.randn((B, 4), device=DEV, generator=g)).contiguous()
    old_lp = sb3_squashed_log_prob(mean, pol.log_std, actions) + 0.3 * torch.randn(B, device=DEV, generator=g)
    adv, ret = torch.randn(B, device=DEV, generator=g), torch.randn(B, device=DEV, generator=g)
    scratch = torch.zeros(16 * max(1024, (B + 31) // 32), device=DEV)
    res = {}
    for fused in (True, False):
        stats = torch.zeros(16, device=DEV)
        pol.grad.fill_(3.0)
        cfg = _lib.PpoLossCfg(0.2, 0.01, 0.5, 1.0 / B, pol.grad.data_ptr() + 4 * pol.log_std_off, None)
        if fused:
            n0 = lib.vf_chain_plugin_launches()
            assert pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch) is True
            assert lib.vf_chain_plugin_launches() == n0 + 1, "the generated class's fused step ran"
        else:
            lib.vf_chain_plugin_set_enabled(0)
            m, v = pol.forward(obs)
            d_mean, d_value = torch.empty((B, 4), device=DEV), torch.empty(B, device=DEV)
            _lib.check(lib.vf_ppo_loss(m.data_ptr(), v.data_ptr(), pol.log_std.data_ptr(), actions.data_ptr(), old_lp.data_ptr(),
                                       adv.data_ptr(), ret.data_ptr(), d_mean.data_ptr(), d_value.data_ptr(), stats.data_ptr(), B,
                                       C.byref(cfg), scratch.data_ptr(), st()))
            pol.backward(d_mean, d_value, None)
            lib.vf_chain_plugin_set_enabled(1)
        res[fused] = (pol.grad.clone(), stats.clone())
    (g1, s1), (g0, s0) = res[True], res[False]
    assert torch.allclose(s1[:9], s0[:9], rtol=2e-5, atol=1e-6 * max(1.0, s0[:9].abs().max().item())), (s1, s0)
    scale = g0.abs().max().item()
    assert (g1 - g0).abs().max().item() <= (1e-4 if "tanh" in acts else 5e-6) * scale, ((g1 - g0).abs().max().item(), scale)


def test_ppo_without_policy_kwargs_builds_the_references_default_network_and_trains(monkeypatch):
    """`PPO(env)` with no policy_kwargs: CustomMultiInputActorCriticPolicy's defaults -- Tanh trunks (policies.py:108), ReLU extractor MLPs
    (extractors.py:666) -- until r06 this raised.  It trains (value loss falls) on the generated chain class of that network (pre-built:
    visfly_amd/_jit.py PREBUILD_ACT) -- fused step, persistent roll-out, no fallback warning; string / class spellings of activation_fn are
    accepted"""
    import warnings
    from visfly_amd.envs import NavigationEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
    mk = lambda: NavigationEnv(num_agent_per_scene=1024, seed=1, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=64,
                               tensor_output=True, random_kwargs=spawn)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ppo = PPO(mk(), n_steps=16, batch_size=4096, n_epochs=4, learning_rate=1e-3, seed=3)
        pol = ppo.policy
        assert [ly.relu for ly in pol.layers] == [1, 1, 1, 1, 2, 2, 0, 2, 2, 0]
        ppo.learn(16 * 1024)
        v0 = ppo.logs["train/value_loss"]
        ppo.learn(16 * 1024 * 6)
    torch.cuda.synchronize()
    assert ppo.logs["train/value_loss"] < v0 and np.isfinite(ppo.logs["train/loss"]) and bool(torch.isfinite(pol.flat).all())
    fb = [x for x in w if "register-chained" in str(x.message) or "not available" in str(x.message)]
    assert not fb and pol.chain_jit and ppo.fused_rollout and pol._fused_ppo is not False, [str(x.message)[:120] for x in w]
    monkeypatch.setenv("VISFLY_AMD_JIT", "0")          # (spellings only: no class is compiled for these)
    for spelled in ("Tanh", torch.nn.Tanh, "tanh"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p2 = PPO(mk(), n_steps=16, batch_size=4096, policy_kwargs=dict(activation_fn=spelled, features_extractor_kwargs=dict(activation_fn="elu")))
        assert (p2.policy.act, p2.policy.ext_act) == (2, 3)
    with pytest.raises(NotImplementedError):
        PPO(mk(), policy_kwargs=dict(activation_fn="gelu"))


def sb3_squashed_log_prob(mean, log_std, actions):
    """SB3 SquashedDiagGaussianDistribution.log_prob(actions) with gaussian_actions=None"""
    eps = torch.finfo(torch.float32).eps
    g = torch.atanh(actions.clamp(-1 + eps, 1 - eps))
    dist = torch.distributions.Normal(mean, log_std.exp().expand_as(mean))
    lp = dist.log_prob(g).sum(dim=1)
    return lp - torch.log(1 - actions ** 2 + 1e-6).sum(dim=1)


def torch_ppo_loss(ref, obs, actions, old_lp, adv, ret, clip, ent_coef, vf_coef):
    """PPO.train's loss (utils/algorithms/PPO.py:210-263) in plain torch"""
    mean, value = ref(obs)
    lp = sb3_squashed_log_prob(mean, ref.log_std, actions)
    ratio = torch.exp(lp - old_lp)
    pl = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
    vl = torch.nn.functional.mse_loss(ret, value.flatten())
    el = -torch.mean(-lp)
    return pl + ent_coef * el + vf_coef * vl, (pl, vl, el, ((ratio - 1) - (lp - old_lp)).mean(),
                                               ((ratio - 1).abs() > clip).float().mean())


@pytest.mark.parametrize("keys", [("state",), ("state", "target")])
def test_policy_loss_and_gradients_vs_torch_autograd(keys):
    from visfly_amd.ppo import MlpPolicy
    _lib, lib = L()
    dims = {"state": 13, "target": 3}
    pol = MlpPolicy({k: dims[k] for k in keys}, {k: [128, 64] for k in keys}, [64, 64], [64, 64], DEV, seed=5,
                    log_std_init=-0.3)
    assert pol.n_params == (43977 if len(keys) == 2 else 27017)          # SURVEY a19 [probe]
    B = 3000
    g = torch.Generator(device=DEV).manual_seed(11)
    obs = {k: torch.randn((B, dims[k]), device=DEV, generator=g) for k in keys}
    mean, value = pol.forward(obs)
    ref = pol.to_torch().double()
    obs64 = {k: v.cpu().double() for k, v in obs.items()}
    rmean, rvalue = ref(obs64)
    assert torch.allclose(mean.cpu().double(), rmean, rtol=1e-4, atol=2e-5)
    assert torch.allclose(value.cpu().double(), rvalue, rtol=1e-4, atol=2e-5)
    # rollout-like data: actions sampled near the policy, some far (clipping active), mixed-sign advantages
    actions = torch.tanh(mean + 0.7 * torch.randn((B, 4), device=DEV, generator=g)).contiguous()
    old_lp = sb3_squashed_log_prob(mean, pol.log_std, actions) + 0.3 * torch.randn(B, device=DEV, generator=g)
    adv = torch.randn(B, device=DEV, generator=g)
    ret = torch.randn(B, device=DEV, generator=g)
    clip, ent, vf = 0.2, 0.01, 0.5
    d_mean, d_value = torch.empty((B, 4), device=DEV), torch.empty(B, device=DEV)
    stats, scratch = torch.zeros(16, device=DEV), torch.zeros(16 * 1024, device=DEV)
    cfg = _lib.PpoLossCfg(clip, ent, vf, 1.0 / B)
    _lib.check(lib.vf_ppo_loss(mean.data_ptr(), value.data_ptr(), pol.log_std.data_ptr(), actions.data_ptr(), old_lp.data_ptr(),
                               adv.data_ptr(), ret.data_ptr(), d_mean.data_ptr(), d_value.data_ptr(), stats.data_ptr(), B,
                               C.byref(cfg), scratch.data_ptr(), st()))
    pol.backward(d_mean, d_value, stats[5:9])
    loss, parts = torch_ppo_loss(ref, obs64, actions.cpu().double(), old_lp.cpu().double(), adv.cpu().double(),
                                 ret.cpu().double(), clip, ent, vf)
    loss.backward()
    got = (stats[:5] / B).cpu().double()
    want = torch.stack([p.detach() for p in parts])
    assert torch.allclose(got, want, rtol=2e-4, atol=2e-6), (got, want)
    gref = ref.flat_grad().double()
    gk = pol.grad.cpu().double()
    scale = gref.abs().max()
    assert (gk - gref).abs().max() <= 2e-4 * scale, ((gk - gref).abs().max(), scale)
    assert torch.allclose(gk[pol.log_std_off:], gref[pol.log_std_off:], rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("M", [1, 63, 777, 25600])
@pytest.mark.parametrize("shape", ["reference", "other"])
def test_fused_forward_equals_layerwise(M, shape):
    """one-launch whole-network forward vs the per-layer kernels: heads, and every saved activation the backward reads.
    "other" (a shape the register-chained kernel is not instantiated for) runs the LDS kernel, which is bit-identical
    to the per-layer kernels; the reference-default shape runs the register-chained kernel (vf_mlp_chain.hip), whose
    reduction order over k differs -- equal to fp32 rounding"""
    from visfly_amd.ppo import MlpPolicy
    ext = {"state": [128, 64], "target": [128, 64]} if shape == "reference" else {"state": [128, 40], "target": [96, 64]}      # (40: off the 32 grid, so no generated chain class either)
    pol = MlpPolicy({"state": 13, "target": 3}, ext, [64, 64], [64, 64], DEV, seed=9)
    assert pol._plan is not None and pol._plan["total"] * 4 <= 160 * 1024
    g = torch.Generator(device=DEV).manual_seed(M)
    obs = {"state": torch.randn((M, 13), device=DEV, generator=g), "target": torch.randn((M, 3), device=DEV, generator=g)}
    pol.fused = True
    mean, value = pol.forward(obs, save_activations=True)
    fused = {k: v.clone() for k, v in pol._buffers(M).items() if not k.startswith(("g:", "obs:"))}
    for v in pol._buffers(M).values():
        if v.dtype == torch.float32 and v.data_ptr() not in (obs["state"].data_ptr(), obs["target"].data_ptr()):
            v.fill_(-123.0)
    pol.fused = False
    pol.forward(obs)
    for k, v in pol._buffers(M).items():
        if k in fused:
            if shape == "other":
                assert torch.equal(fused[k], v), k
            else:
                assert torch.allclose(fused[k], v, rtol=1e-5, atol=1e-6), (k, (fused[k] - v).abs().max())
    pol.fused = True
    m2, v2 = pol.forward(obs, save_activations=False)       # inference: only the heads leave the chip
    assert torch.equal(m2, fused["mean"]) and torch.equal(v2, fused["value"])


def test_chain_forward_hover_shape_and_ragged_rows():
    """register-chained forward, StateExtractor shape, row counts around the 32-row tile: vs torch fp32"""
    from visfly_amd.ppo import MlpPolicy
    pol = MlpPolicy({"state": 13}, {"state": [128, 64]}, [64, 64], [64, 64], DEV, seed=4)
    ref = pol.to_torch().to(DEV)
    for M in (1, 31, 32, 33, 95, 4097):
        obs = {"state": torch.randn((M, 13), device=DEV)}
        mean, value = pol.forward(obs, save_activations=True)
        with torch.no_grad():
            m0, v0 = ref(obs)
        assert torch.allclose(mean, m0, rtol=1e-5, atol=1e-6) and torch.allclose(value.view_as(v0), v0, rtol=1e-5, atol=1e-6)
        b = pol._buffers(M)
        with torch.no_grad():
            h1 = torch.relu(ref.lin[0](obs["state"]))
            feat = torch.relu(ref.lin[1](h1))
        assert torch.allclose(b["x:state:0"], h1, rtol=1e-5, atol=1e-6) and torch.allclose(b["feat"], feat, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M", [1, 63, 777, 25600, 40000])
@pytest.mark.parametrize("with_value", [True, False])
def test_fused_backward_equals_layerwise_and_torch(M, with_value):
    """one-launch whole-network backward (block-private tiles, dW kept in MFMA accumulators across a block's
    tiles) vs the per-layer kernels (data gradients bit-identical: same per-row arithmetic; weight gradients
    equal up to the order of the fp32 row sums) and vs torch autograd on the same weights"""
    from visfly_amd.ppo import MlpPolicy
    pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], DEV, seed=9)
    g = torch.Generator(device=DEV).manual_seed(M)
    obs = {"state": torch.randn((M, 13), device=DEV, generator=g), "target": torch.randn((M, 3), device=DEV, generator=g)}
    d_mean = torch.randn((M, 4), device=DEV, generator=g) / M
    d_value = torch.randn(M, device=DEV, generator=g) / M if with_value else None
    d_ls = torch.randn(4, device=DEV, generator=g)
    res = {}
    for fused in (False, True):
        pol.fused_backward = fused
        pol.grad.fill_(7.0)
        pol.forward(obs)
        d_in = pol.backward(d_mean, d_value, d_ls, need_input_grad=True)
        res[fused] = (pol.grad.clone(), {k: v.clone() for k, v in d_in.items()})
    (g0, in0), (g1, in1) = res[False], res[True]
    for k in in0:
        if with_value:      # both trunks + observation gradient: block-tile kernel, same per-row arithmetic as the layer kernels
            assert torch.equal(in0[k], in1[k]), k
        else:               # policy trunk + observation gradient: register-chained reverse sweep (other reduction order)
            assert torch.allclose(in0[k], in1[k], rtol=1e-4, atol=1e-6 * in0[k].abs().max().item()), k
    scale = g0.abs().max().item()
    assert (g0 - g1).abs().max().item() <= 2e-6 * scale
    if not with_value:                                           # the value trunk was skipped: its entries are untouched
        vf = [ly for ly in pol.layers if ly.dst == "value" or ly.dst.startswith("vf:")]
        assert all((g1[ly.w_off:ly.w_off + ly.K * ly.No] == 7.0).all() for ly in vf)
    # accumulate mode adds the same gradient once more
    pol.forward(obs)
    pol.backward(d_mean, d_value, d_ls, accumulate=True)
    live = g1 != 7.0
    assert torch.allclose(pol.grad[live], 2 * g1[live], rtol=1e-5, atol=1e-6 * scale)
    # torch autograd reference
    ref = pol.to_torch()
    o = {k: v.cpu().double() for k, v in obs.items()}
    ref = ref.double()
    mean, value = ref(o)
    loss = (mean * d_mean.cpu().double()).sum() + ((value.view(-1) * d_value.cpu().double()).sum() if with_value else 0.0)
    loss.backward()
    for ly, m in zip(pol.layers, ref.lin):
        if m.weight.grad is None:
            continue
        got = g1[ly.w_off:ly.w_off + ly.K * ly.No].view(ly.No, ly.K).cpu().double()
        assert (got - m.weight.grad).abs().max().item() <= 2e-5 * max(m.weight.grad.abs().max().item(), 1e-6), ly.dst
        gb = g1[ly.b_off:ly.b_off + ly.No].cpu().double()
        assert (gb - m.bias.grad).abs().max().item() <= 2e-5 * max(m.bias.grad.abs().max().item(), 1e-6), ly.dst


def test_adam_with_grad_clip_vs_torch():
    _lib, lib = L()
    n = 43977
    g = torch.Generator(device=DEV).manual_seed(1)
    p = torch.randn(n, device=DEV, generator=g)
    pr = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([pr], lr=1e-3, weight_decay=1e-5, eps=1e-8)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sumsq, scratch = torch.zeros(1, device=DEV), torch.zeros(4096, device=DEV)
    for step in range(1, 6):
        grad = torch.randn(n, device=DEV, generator=g) * (10.0 if step % 2 else 0.001)
        pr.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([pr], 0.5)
        opt.step()
        _lib.check(lib.vf_sumsq(grad.data_ptr(), n, sumsq.data_ptr(), scratch.data_ptr(), st()))
        cfg = _lib.AdamCfg(1e-3, 0.9, 0.999, 1e-8, 1e-5, 0.5, step, 0)
        _lib.check(lib.vf_adam_step(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, sumsq.data_ptr(),
                                    C.byref(cfg), st()))
        assert torch.allclose(p, pr.data, rtol=1e-5, atol=1e-6), (step, (p - pr.data).abs().max())


def test_head_sample_distribution_and_log_prob():
    _lib, lib = L()
    M = 200000
    mean = torch.zeros((M, 4), device=DEV) + torch.tensor([0.0, 0.5, -0.5, 1.0], device=DEV)
    log_std = torch.tensor([0.0, -0.5, -1.0, -2.0], device=DEV)
    a, lp = torch.empty((M, 4), device=DEV), torch.empty(M, device=DEV)
    _lib.check(lib.vf_head_sample(mean.data_ptr(), log_std.data_ptr(), a.data_ptr(), lp.data_ptr(), M, 123, 1, 0, st()))
    gz = torch.atanh(a.clamp(-1 + 1e-6, 1 - 1e-6))
    z = (gz - mean) / log_std.exp()
    ok = a.abs().max(dim=1).values < 0.999          # away from tanh saturation
    assert abs(float(z[ok].mean())) < 0.02 and abs(float(z[ok].std()) - 1.0) < 0.03
    assert torch.allclose(lp[ok], sb3_squashed_log_prob(mean, log_std, a)[ok], rtol=1e-4, atol=1e-3)
    a2, lp2 = torch.empty_like(a), torch.empty_like(lp)
    _lib.check(lib.vf_head_sample(mean.data_ptr(), log_std.data_ptr(), a2.data_ptr(), lp2.data_ptr(), M, 123, 1, 0, st()))
    assert torch.equal(a, a2)                                       # counter RNG: reproducible
    _lib.check(lib.vf_head_sample(mean.data_ptr(), log_std.data_ptr(), a2.data_ptr(), lp2.data_ptr(), M, 123, 2, 0, st()))
    assert not torch.equal(a, a2)
    _lib.check(lib.vf_head_sample(mean.data_ptr(), log_std.data_ptr(), a2.data_ptr(), lp2.data_ptr(), M, 123, 2, 1, st()))
    assert torch.allclose(a2, torch.tanh(mean), atol=1e-6)


def test_ppo_learn_runs_and_improves_value_fit():
    """end-to-end: rollouts from the fused env, GAE, minibatch updates; the value loss must drop"""
    from visfly_amd.envs import HoverEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    env = HoverEnv(num_agent_per_scene=2048, seed=1, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=64,
                   tensor_output=True)
    ppo = PPO(env, n_steps=32, batch_size=8192, n_epochs=4, learning_rate=3e-4, seed=3, policy_kwargs=dict(activation_fn="relu"))
    p0 = ppo.policy.flat.clone()
    ppo.learn(32 * 2048)
    v_first = ppo.logs["train/value_loss"]
    ppo.learn(32 * 2048 * 6)
    assert torch.isfinite(ppo.policy.flat).all() and not torch.equal(p0, ppo.policy.flat)
    assert ppo.logs["train/value_loss"] < v_first
    assert 0.0 <= ppo.logs["train/clip_fraction"] <= 1.0 and ppo.logs["time/fps"] > 0
    # rollout statistics (PPO._dump_logs): every agent times out after 64 steps, so 32-step rollouts finish one episode per two
    assert ppo.logs["rollout/episodes"] == 2048 and 0 < ppo.logs["rollout/ep_len_mean"] <= 64
    assert np.isfinite(ppo.logs["rollout/ep_rew_mean"]) and 0.0 <= ppo.logs["rollout/ep_success_rate"] <= 1.0
    # Adam refreshed the packed MFMA weight images incrementally (vf_adam_cfg.pack_map): they must equal a fresh pack
    pol = ppo.policy
    kept = pol._packed.clone()
    assert pol._packed_stamp == pol._stamp
    pol.lazy_pack = False
    pol._pack()
    assert torch.equal(kept, pol._packed)


def test_ppo_training_is_bitwise_reproducible():
    """every reduction on the path (MFMA partial folds, loss statistics, grad norm) has a fixed order: two runs from the
    same seeds end with bit-identical parameters"""
    from visfly_amd.envs import HoverEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    flats = []
    for _ in range(2):
        env = HoverEnv(num_agent_per_scene=1024, seed=1, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=64,
                       tensor_output=True)
        ppo = PPO(env, n_steps=16, batch_size=4096, n_epochs=2, learning_rate=3e-4, seed=3, policy_kwargs=dict(activation_fn="relu"))
        ppo.learn(16 * 1024 * 3)
        flats.append(ppo.policy.flat.clone())
    assert torch.equal(flats[0], flats[1])


def test_ppo_checkpoint_round_trip_and_reference_yaml_kwargs(tmp_path):
    """the `algorithm:` block of the reference's YAMLs passes as **kwargs; save -> load restores the
    policy, the Adam moments and the step count, and the loaded trainer acts identically"""
    from visfly_amd.envs import NavigationEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    alg = dict(policy="CustomMultiInputPolicy", verbose=1, device="cuda", gamma=0.99, n_steps=16, ent_coef=0.0,
               learning_rate=5e-5, vf_coef=0.5, max_grad_norm=0.5, batch_size=4096, gae_lambda=0.95, n_epochs=2, clip_range=0.2,
               policy_kwargs=dict(ortho_init=False, features_extractor_class="StateTargetExtractor",
                                  features_extractor_kwargs=dict(net_arch=dict(state=dict(layer=[128, 64]), target=dict(layer=[128, 64]))),
                                  net_arch=dict(pi=[64, 64], vf=[64, 64]), activation_fn="ReLU",
                                  optimizer_kwargs=dict(weight_decay=2e-5)))
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1.0, 0.0, 1.5], "half": [0.0, 2.0, 1.0]}}]}}

    def make():
        return NavigationEnv(num_agent_per_scene=1024, seed=1, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=64,
                             tensor_output=True, random_kwargs=spawn)
    ppo = PPO(make(), seed=3, **alg)
    assert ppo.weight_decay == 2e-5 and ppo.policy.spec["extractor"] == {"state": [128, 64], "target": [128, 64]}
    ppo.learn(16 * 1024 * 2)
    path = ppo.save(str(tmp_path / "PPO_std_1"))
    new = PPO.load(path, make(), seed=3, **alg)
    assert torch.equal(new.policy.flat, ppo.policy.flat), (int((new.policy.flat != ppo.policy.flat).sum()), int(torch.isnan(ppo.policy.flat).sum()), int(torch.isnan(new.policy.flat).sum()))
    assert torch.equal(new.exp_avg, ppo.exp_avg) and torch.equal(new.exp_avg_sq, ppo.exp_avg_sq)
    assert new._opt_step == ppo._opt_step and new.num_timesteps == ppo.num_timesteps
    obs = ppo.env.get_observation()
    a0 = ppo.policy.forward({k: obs[k] for k in ppo.obs_keys}, save_activations=False)
    a1 = new.policy.forward({k: obs[k] for k in ppo.obs_keys}, save_activations=False)
    assert torch.equal(a0[0], a1[0]) and torch.equal(a0[1], a1[1])


@pytest.mark.parametrize("M", [1, 33, 777, 25600])
@pytest.mark.parametrize("net,mode", [("nav", "ppo"), ("hover", "ppo"), ("nav", "bptt"), ("hover", "bptt")])
def test_chain_backward_vs_torch_and_block_tile_kernel(M, net, mode):
    """reverse chain in registers + row-slab weight gradients (vf_mlp_chain.hip, vf_mlp_wgrad.hip), the two variants the
    trainers use (PPO update: both trunks, no observation gradient; first-order optimisation: policy trunk + observation
    gradient), against torch autograd on the same weights and against the layer-by-layer kernels; deterministic"""
    from visfly_amd.ppo import MlpPolicy
    dims = {"state": 13, "target": 3} if net == "nav" else {"state": 13}
    pol = MlpPolicy(dims, {k: [128, 64] for k in dims}, [64, 64], [64, 64], DEV, seed=9)
    g = torch.Generator(device=DEV).manual_seed(M)
    obs = {k: torch.randn((M, d), device=DEV, generator=g) for k, d in dims.items()}
    d_mean = torch.randn((M, 4), device=DEV, generator=g) / M
    d_value = torch.randn(M, device=DEV, generator=g) / M if mode == "ppo" else None
    ig = mode == "bptt"
    ref = pol.to_torch().to(DEV)
    xs = {k: v.clone().requires_grad_(ig) for k, v in obs.items()}
    m0, v0 = ref(xs)
    loss = (m0 * d_mean).sum() + ((v0.view(-1) * d_value).sum() if d_value is not None else 0.0) + 0.0 * ref.log_std.sum()
    loss.backward()
    for mod in ref.lin:
        for prm in mod.parameters():
            if prm.grad is None:
                prm.grad = torch.zeros_like(prm)
    gref = ref.flat_grad().to(DEV)
    res = {}
    for fused in (True, False, True):
        pol.fused_backward = fused
        pol.grad.fill_(0.0)
        pol.forward(obs)
        d_in = pol.backward(d_mean, d_value, None, need_input_grad=ig)
        if fused and fused in res:                               # second run of the chain path: bit-identical
            assert torch.equal(res[True][0], pol.grad)
        res[fused] = (pol.grad.clone(), {k: v.clone() for k, v in d_in.items()})
    scale = gref.abs().max().item()
    for fused in (True, False):
        gk = res[fused][0].clone()
        gk[pol.log_std_off:] = 0
        assert (gk - gref).abs().max().item() <= 5e-6 * scale, (fused, (gk - gref).abs().max().item(), scale)
        for k, v in res[fused][1].items():
            assert torch.allclose(v, xs[k].grad, rtol=1e-4, atol=1e-6 * xs[k].grad.abs().max().item())
    # accumulate mode
    pol.fused_backward = True
    pol.forward(obs)
    pol.backward(d_mean, d_value, None, accumulate=True, need_input_grad=ig)
    assert torch.allclose(pol.grad, 2 * res[True][0], rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize("M", [1, 33, 777, 16384, 25600])
@pytest.mark.parametrize("net", ["nav", "hover"])
@pytest.mark.parametrize("ig", [True, False])
def test_sac_actor_chain_vs_torch_and_block_tile_kernel(M, net, ig):
    """r04: the reference's SAC-style Actor (td_policies.py:146-252: latent_pi -> mu, log_latent_pi -> log_std, two 4-wide heads; the
    actor of its BPTT and SHAC loops) on the register-chained kernels (vf_mlp_chain_sac.hip: forward in 32- and 16-row form, reverse
    chain of both trunks with and without the observation gradient) against torch on the same weights and against the block-tile
    kernels it ran on until r03; deterministic"""
    from visfly_amd import _lib
    from visfly_amd.ppo import MlpPolicy
    import ctypes as C
    dims = {"state": 13, "target": 3} if net == "nav" else {"state": 13}
    pol = MlpPolicy(dims, {k: [128, 64] for k in dims}, [64, 64], [64, 64], DEV, seed=9, ortho_init=False, head_dims=(4, 4), log_std_param=False)
    g = torch.Generator(device=DEV).manual_seed(M + 5)
    obs = {k: torch.randn((M, d), device=DEV, generator=g) for k, d in dims.items()}
    d_mu = torch.randn((M, 4), device=DEV, generator=g) / M
    d_ls = torch.randn((M, 4), device=DEV, generator=g) / M
    ref = pol.to_torch().double().to(DEV)          # fp64 reference: the bound below is this path's own rounding
    xs = {k: v.double().requires_grad_(ig) for k, v in obs.items()}
    m0, v0 = ref(xs)
    ((m0 * d_mu.double()).sum() + (v0 * d_ls.double()).sum()).backward()
    gref = ref.flat_grad().to(DEV).float()
    m0, v0 = m0.float(), v0.float()
    mu, ls = pol.forward(obs)
    assert mu.shape == ls.shape == (M, 4)
    sc = max(m0.abs().max().item(), v0.abs().max().item())
    assert (mu - m0).abs().max().item() <= 2e-6 * sc and (ls - v0).abs().max().item() <= 2e-6 * sc
    # the chain classes ARE what ran: the library's capability query answers for this layer table
    b = pol._buffers(M, 0)
    bd = pol._bwd_desc(b, M, d_mu, d_ls, ig)[0]
    assert _lib.lib().vf_mlp_backward_data_supported(C.byref(bd)) == 1
    res = {}
    for fused in (True, False, True):
        pol.fused_backward = fused
        pol.grad.fill_(0.0)
        pol.forward(obs)
        d_in = pol.backward(d_mu, d_ls, None, need_input_grad=ig)
        if fused and fused in res:
            assert torch.equal(res[True][0], pol.grad)
        res[fused] = (pol.grad.clone(), {k: v.clone() for k, v in d_in.items()})
    scale = gref.abs().max().item()
    # the two backward implementations start from the SAME saved activations (one forward kernel): they must agree to rounding ...
    assert (res[True][0] - res[False][0]).abs().max().item() <= 5e-6 * scale
    for fused in (True, False):
        # ... while against the fp64 network a ReLU mask can flip where a pre-activation is within fp32 rounding of zero: one flipped
        # unit in one row moves a gradient entry by ~|d| |w x| ~ 1 / M -- measured 1.5e-5 .. 2e-4 of the largest entry at M >= 16 384
        # (identically for both implementations), < 2e-7 below
        err = (res[fused][0] - gref).abs().max().item()
        assert err <= (1e-3 if M >= 16384 else 2e-6) * scale, (fused, err, scale)
        for k, v in res[fused][1].items():      # per-row quantity: a row with a flipped unit is off as a whole -- a handful of rows at most
            want = xs[k].grad.float()
            bad = ((v - want).abs() > 1e-4 * want.abs() + 1e-5 * want.abs().max()).any(dim=1)
            assert int(bad.sum()) <= (8 if M >= 16384 else 0), (k, int(bad.sum()))
    for k in res[True][1]:                      # ... and the two implementations agree on every row
        assert torch.allclose(res[True][1][k], res[False][1][k], rtol=1e-4, atol=1e-6 * res[False][1][k].abs().max().item())
    pol.fused_backward = True
    pol.forward(obs)
    pol.backward(d_mu, d_ls, None, accumulate=True, need_input_grad=ig)
    assert torch.allclose(pol.grad, 2 * res[True][0], rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize("M", [1, 33, 777, 16384, 40000])
@pytest.mark.parametrize("net", ["hover"])        # (over the two-branch extractor the concatenation is 132 wide: beyond the layer kernels' 128)
def test_twin_critic_chain_vs_torch_and_block_tile_kernel(M, net):
    """r04: the reference's twin ContinuousCritic (td_policies.py:82-143: own extractor, th.cat([features, actions]) -> qf0 / qf1 -> Q;
    68-wide first trunk layers, two 1-wide heads) on the register-chained kernels (vf_mlp_chain_sac.hip, ChainNet<.., PASS = 1>: the
    action columns are one more input tile, the frozen identity layer of the table is not executed) against torch on the same weights
    and against the block-tile kernels it ran on until r03; the saved feature rows carry the action columns (the weight gradients of
    the 68-wide layers read them); deterministic"""
    from visfly_amd import _lib
    from visfly_amd.ppo import MlpPolicy
    import ctypes as C
    dims = {"state": 13, "target": 3, "action": 4} if net == "nav" else {"state": 13, "action": 4}
    ext = {k: [128, 64] for k in dims if k != "action"}
    pol = MlpPolicy(dims, ext, [64, 64], [64, 64], DEV, seed=11, ortho_init=False, head_dims=(1, 1), passthrough=("action",), log_std_param=False)
    g = torch.Generator(device=DEV).manual_seed(M + 7)
    obs = {k: torch.randn((M, d), device=DEV, generator=g) for k, d in dims.items()}
    obs["action"] = torch.tanh(obs["action"])
    d_q0 = torch.randn((M, 1), device=DEV, generator=g) / M
    d_q1 = torch.randn((M, 1), device=DEV, generator=g) / M
    ref = pol.to_torch().double().to(DEV)
    q0r, q1r = ref({k: v.double() for k, v in obs.items()})
    ((q0r * d_q0.double()).sum() + (q1r * d_q1.double()).sum()).backward()
    gref = ref.flat_grad().to(DEV).float()[:pol.n_params]
    q0, q1 = pol.forward(obs)
    assert q0.shape == q1.shape == (M, 1)
    sc = max(q0r.abs().max().item(), q1r.abs().max().item())
    assert (q0 - q0r.float()).abs().max().item() <= 2e-6 * sc and (q1 - q1r.float()).abs().max().item() <= 2e-6 * sc
    nfeat = 64 * len(ext)
    assert torch.equal(pol._buffers(M, 0)["feat"][:, nfeat:], obs["action"])          # pass-through columns of the saved feature rows
    b = pol._buffers(M, 0)
    bd = pol._bwd_desc(b, M, d_q0, d_q1, False)[0]
    assert _lib.lib().vf_mlp_backward_data_supported(C.byref(bd)) == 1                # the chain class IS what runs
    res = {}
    for fused in (True, False, True):
        pol.fused_backward = fused
        pol.grad.fill_(0.0)
        pol.forward(obs)
        pol.backward(d_q0, d_q1, None)
        if fused and fused in res:
            assert torch.equal(res[True], pol.grad)
        res[fused] = pol.grad.clone()
    scale = gref.abs().max().item()
    assert (res[True] - res[False]).abs().max().item() <= 5e-6 * scale
    for fused in (True, False):                      # vs fp64: ReLU-mask flips at large M (test_sac_actor_chain_vs_torch_and_block_tile_kernel)
        err = (res[fused] - gref).abs().max().item()
        assert err <= (1e-3 if M >= 16384 else 2e-6) * scale, (fused, err, scale)


@pytest.mark.parametrize("shape", ["reference", "other"])
def test_policy_only_forward_and_split_backward(shape):
    """need_value=False: same action mean (the register-chained kernel skips the value trunk, other layer tables fall
    back to the full forward); backward_data + weight_grad_slots over reserved slots == per-slot backward, summed"""
    from visfly_amd.ppo import MlpPolicy
    ext = {"state": [128, 64]} if shape == "reference" else {"state": [100, 64]}      # (100: no built-in and no generated chain class)
    pol = MlpPolicy({"state": 13}, ext, [64, 64], [64, 64], DEV, seed=21)
    M, n = 777, 3
    g = torch.Generator(device=DEV).manual_seed(5)
    obs = [{"state": torch.randn((M, 13), device=DEV, generator=g)} for _ in range(n)]
    m_full, v_full = pol.forward(obs[0])
    m_full = m_full.clone()
    m_pi, v_pi = pol.forward(obs[0], need_value=False)
    assert v_pi is None and torch.equal(m_pi, m_full)
    d_means = torch.randn((n, M, 4), device=DEV, generator=g) / M
    # reference: per-slot fused backward, accumulated
    pol.grad.zero_()
    want_in = []
    for s in range(n):
        pol.forward(obs[s], slot=s, need_value=False)
        want_in.append(pol.backward(d_means[s], None, None, accumulate=True, need_input_grad=True, slot=s)["state"].clone())
    want = pol.grad.clone()
    if not pol.backward_data_supported(M):
        import os
        assert shape == "other" or os.environ.get("VISFLY_AMD_MLP_CHAIN") == "0"     # A/B switch of the chain kernels
        return
    assert shape == "reference"
    pol.reserve_slots(M, n)
    pol.grad.zero_()
    for s in range(n):
        pol.forward(obs[s], slot=s, need_value=False)
    for s in reversed(range(n)):
        got_in = pol.backward_data(d_means[s], slot=s)["state"]
        assert torch.equal(got_in, want_in[s])
    pol.weight_grad_slots(M, n, d_means, accumulate=True)
    scale = want.abs().max().item()
    assert (pol.grad - want).abs().max().item() <= 2e-6 * scale


@pytest.mark.parametrize("form", ["split", "one-wave"])
@pytest.mark.parametrize("net", ["nav", "hover"])
@pytest.mark.parametrize("B", [25600, 1000, 33, 40000])
def test_fused_ppo_update_equals_separate_launches(net, B, form, monkeypatch):
    """vf_ppo_update (forward + loss + reverse chain in one launch, masks from the live forward registers) + vf_mlp_weight_grad
    vs forward / vf_ppo_loss / backward: same statistics, same gradient (up to fp32 summation order) -- in both forms of the fused
    kernel: two half-network waves per row tile (k_ppo_update_split, the default) and one wave per tile (k_ppo_update_chain)"""
    monkeypatch.setenv("VISFLY_AMD_CHAIN_SPLIT", "1" if form == "split" else "0")     # read by the library per call
    from visfly_amd.ppo import MlpPolicy
    _lib, lib = L()
    dims = {"state": 13, "target": 3} if net == "nav" else {"state": 13}
    pol = MlpPolicy(dims, {k: [128, 64] for k in dims}, [64, 64], [64, 64], DEV, seed=5, log_std_init=-0.3)
    g = torch.Generator(device=DEV).manual_seed(B)
    obs = {k: torch.randn((B, d), device=DEV, generator=g) for k, d in dims.items()}
    mean, value = pol.forward(obs)
    actions = torch.tanh(mean + 0.7 * torch.randn((B, 4), device=DEV, generator=g)).contiguous()
    old_lp = sb3_squashed_log_prob(mean, pol.log_std, actions) + 0.3 * torch.randn(B, device=DEV, generator=g)
    adv, ret = torch.randn(B, device=DEV, generator=g), torch.randn(B, device=DEV, generator=g)
    scratch = torch.zeros(16 * max(1024, (B + 31) // 32), device=DEV)     # one row of 16 partial statistics per 32-row tile (r05: any B)
    res = {}
    for fused in (True, False):
        stats = torch.zeros(16, device=DEV)
        pol.grad.fill_(3.0)
        cfg = _lib.PpoLossCfg(0.2, 0.01, 0.5, 1.0 / B, pol.grad.data_ptr() + 4 * pol.log_std_off, None)
        if fused:
            # a first call on OTHER observation tensors: the cached layer tables must follow the caller's tensors
            other = {k: torch.randn_like(v) for k, v in obs.items()}
            assert pol.ppo_update(other, actions, old_lp, adv, ret, cfg, stats, scratch)
            del other
            assert pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch)
        else:
            m, v = pol.forward(obs)
            d_mean, d_value = torch.empty((B, 4), device=DEV), torch.empty(B, device=DEV)
            _lib.check(lib.vf_ppo_loss(m.data_ptr(), v.data_ptr(), pol.log_std.data_ptr(), actions.data_ptr(), old_lp.data_ptr(),
                                       adv.data_ptr(), ret.data_ptr(), d_mean.data_ptr(), d_value.data_ptr(), stats.data_ptr(), B,
                                       C.byref(cfg), scratch.data_ptr(), st()))
            pol.backward(d_mean, d_value, None)
        res[fused] = (pol.grad.clone(), stats.clone())
    (g1, s1), (g0, s0) = res[True], res[False]
    assert torch.allclose(s1[:9], s0[:9], rtol=2e-5, atol=1e-6 * max(1.0, s0[:9].abs().max().item())), (s1, s0)
    scale = g0.abs().max().item()
    assert (g1 - g0).abs().max().item() <= 5e-6 * scale, ((g1 - g0).abs().max().item(), scale)
    assert torch.allclose(g1[pol.log_std_off:], g0[pol.log_std_off:], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("form", ["split", "one-wave"])
@pytest.mark.parametrize("net,vclip", [("nav", False), ("hover", True)])
@pytest.mark.parametrize("B", [25600, 1000, 33])
def test_fused_ppo_update_on_an_indexed_minibatch(net, vclip, B, form, monkeypatch):
    """r05 (ABI 9): vf_ppo_update reads its minibatch through row indices into the whole rollout buffer (vf_ppo_loss_cfg.row_index: no
    shuffled copy of the buffer per epoch) and leaves the observation rows in minibatch order for the weight gradients -- the same
    statistics and gradient, bit for bit, as the call on the gathered rows; with value clipping (old_value indexed too)"""
    monkeypatch.setenv("VISFLY_AMD_CHAIN_SPLIT", "1" if form == "split" else "0")
    from visfly_amd.ppo import MlpPolicy
    _lib, lib = L()
    dims = {"state": 13, "target": 3} if net == "nav" else {"state": 13}
    pol = MlpPolicy(dims, {k: [128, 64] for k in dims}, [64, 64], [64, 64], DEV, seed=5, log_std_init=-0.3)
    T = 3 * B + 17                                   # rows of the "rollout buffer"
    g = torch.Generator(device=DEV).manual_seed(B)
    obs = {k: torch.randn((T, d), device=DEV, generator=g) for k, d in dims.items()}
    actions = torch.tanh(torch.randn((T, 4), device=DEV, generator=g)).contiguous()
    old_lp, ret, old_v = (torch.randn(T, device=DEV, generator=g) for _ in range(3))
    rows = torch.randperm(T, device=DEV, generator=g)[:B].contiguous()
    adv = torch.randn(B, device=DEV, generator=g)                      # minibatch order (normalised per minibatch)
    scratch = torch.zeros(16 * max(1024, (B + 31) // 32), device=DEV)
    res = {}
    for indexed in (True, False):
        stats = torch.zeros(16, device=DEV)
        pol.grad.fill_(3.0)
        cfg = _lib.PpoLossCfg(0.2, 0.01, 0.5, 1.0 / B, pol.grad.data_ptr() + 4 * pol.log_std_off, None)
        if indexed:
            if vclip:
                cfg.old_value, cfg.clip_range_vf = old_v.data_ptr(), 0.3
            assert pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch, row_index=rows)
            copies = {k: pol._buffers(B, 0)["obs:" + k].clone() for k in dims}
        else:
            ov = old_v[rows].contiguous()
            if vclip:
                cfg.old_value, cfg.clip_range_vf = ov.data_ptr(), 0.3
            o = {k: v[rows].contiguous() for k, v in obs.items()}
            assert pol.ppo_update(o, actions[rows].contiguous(), old_lp[rows].contiguous(), adv, ret[rows].contiguous(), cfg, stats, scratch)
            for k in dims:
                assert torch.equal(copies[k], o[k]), "the observation rows the indexed launch left for the weight gradients"
        res[indexed] = (pol.grad.clone(), stats.clone())
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][0], res[False][0])


def test_ppo_training_with_indexed_minibatches_equals_the_shuffled_copy():
    """PPO.train with the minibatches read through the permutation slice (the default on the chain kernels) ends at the same
    parameters, bit for bit, as with the per-epoch shuffled copy of the rollout buffer (index_minibatches = False); trailing partial
    minibatch, value clipping"""
    from visfly_amd.envs import NavigationEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    flats = []
    for flag in (True, False):
        env = NavigationEnv(num_agent_per_scene=1024, seed=1, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=64, tensor_output=True)
        ppo = PPO(env, n_steps=16, batch_size=6000, n_epochs=3, learning_rate=3e-4, seed=3, clip_range_vf=0.3, policy_kwargs=dict(activation_fn="relu"))
        ppo.index_minibatches = flag
        ppo.learn(16 * 1024 * 3)
        torch.cuda.synchronize()
        flats.append(ppo.policy.flat.clone())
        env.close()
    assert torch.equal(flats[0], flats[1])


@pytest.mark.parametrize("keys,B", [(("state", "target"), 25600), (("state",), 4096), (("state", "target"), 1000), (("state", "target"), 70)])
def test_fused_optimiser_tail_equals_separate_launches(keys, B):
    """vf_mlp_weight_grad_adam (weight gradients + fold + gradient norm + clip + Adam + packed-weight refresh in ONE launch, the waves
    meeting at device counters) leaves, bit for bit, what vf_mlp_weight_grad_sumsq + vf_adam_step leave: gradient, loss statistics,
    their epoch accumulator, the per-block squared-norm partials, parameters, both Adam moments and the packed MFMA weight images --
    over several consecutive steps on the same sync words (the kernel leaves its counters at zero)"""
    from visfly_amd.ppo import MlpPolicy
    _lib, lib = L()
    dims = {"state": 13, "target": 3}
    g = torch.Generator(device=DEV).manual_seed(B)
    obs = {k: torch.randn((B, dims[k]), device=DEV, generator=g) for k in keys}
    actions = torch.tanh(torch.randn((B, 4), device=DEV, generator=g)).contiguous()
    old_lp, adv, ret = (torch.randn(B, device=DEV, generator=g) for _ in range(3))
    out = {}
    for fused in (True, False):
        pol = MlpPolicy({k: dims[k] for k in keys}, {k: [128, 64] for k in keys}, [64, 64], [64, 64], DEV, seed=5)
        pol.lazy_pack = True
        n = pol.n_params
        gbuf = torch.zeros(n + 16, device=DEV)
        pol.grad = gbuf[:n]
        stats, acc = gbuf[n:], torch.zeros(16, device=DEV)
        scratch = torch.zeros(16 * 1024 + 4096, device=DEV)
        m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        sync = torch.zeros(_lib.WGRAD_SYNC_WORDS, dtype=torch.int32, device=DEV)
        trace = []
        for step in range(1, 4):
            cfg = _lib.PpoLossCfg(0.2, 0.01, 0.5, 1.0 / B, pol.grad.data_ptr() + 4 * pol.log_std_off, acc.data_ptr())
            pmap, packed = pol.pack_map()
            acfg = _lib.AdamCfg(1e-3, 0.9, 0.999, 1e-8, 1e-5, 0.05, step, 0, pmap.data_ptr(), packed.data_ptr(), None, 0, pol.log_std_off)
            tail = _lib.WgradTail(pol.flat.data_ptr(), m.data_ptr(), v.data_ptr(), n, acfg, sync.data_ptr()) if fused else None
            res = pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch, want_sumsq=True, tail=tail)
            if fused:
                assert res == "adam", pol.tail_reason
            else:
                sq, nb = res
                acfg.sumsq_partials, acfg.n_sumsq_partials = sq.data_ptr(), nb
                _lib.check(lib.vf_adam_step(pol.flat.data_ptr(), pol.grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, None, C.byref(acfg), st()))
            pol.mark_updated(packed_current=True)
            torch.cuda.synchronize()
            trace.append([t.clone() for t in (pol.grad, stats, acc, pol._sq_part, pol.flat, m, v, pol._packed)])
        slots = sync.view(-1, 16)[:, 0].tolist()                      # one word per 64-byte line (csrc/vf_mlp_wgrad.hip: tail_slot)
        if fused:       # generation = launches; no abort; every arrival count back at zero; release words / group flags at the generation
            counts = slots[2:11] + [slots[19 + 18 * l + k] for l in range(16) for k in (0, 2, 3, 4, 5, 6, 7, 8, 9)]
            assert slots[0] == 3 and slots[1] == 0 and not any(counts) and set(slots) <= {0, 3}, slots[:40]
        else:
            assert not any(slots)
        out[fused] = trace
    for a, b in zip(out[True], out[False]):
        for name, x, y in zip(("grad", "stats", "stats_accum", "sq_part", "param", "exp_avg", "exp_avg_sq", "packed"), a, b):
            assert torch.equal(x, y), name
    assert not torch.equal(out[True][0][4], out[True][2][4]) and float(out[True][0][0].abs().max()) > 0


def test_ppo_training_with_the_fused_tail_equals_the_separate_launches():
    """PPO.learn with the optimiser step's tail inside the weight-gradient launch (opt-in) ends at the same parameters, bit for bit, as
    with fold and Adam as launches of their own; trailing partial minibatch, value clipping, entropy term"""
    from visfly_amd.envs import NavigationEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    flats = []
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
    for flag in (True, False):
        env = NavigationEnv(num_agent_per_scene=1024, seed=1, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=64, tensor_output=True,
                            random_kwargs=spawn)
        ppo = PPO(env, n_steps=16, batch_size=6000, n_epochs=3, learning_rate=3e-4, seed=3, clip_range_vf=0.3, ent_coef=0.01, policy_kwargs=dict(activation_fn="relu"))
        ppo.fused_tail = flag
        ppo.learn(16 * 1024 * 3)
        torch.cuda.synchronize()
        assert (ppo._tail_launches > 0) == flag, ppo.policy.tail_reason
        flats.append((ppo.policy.flat.clone(), ppo.exp_avg.clone(), ppo.exp_avg_sq.clone(), dict(ppo.logs)))
        env.close()
    assert torch.equal(flats[0][0], flats[1][0]) and torch.equal(flats[0][1], flats[1][1]) and torch.equal(flats[0][2], flats[1][2])
    logs = [{k: v for k, v in f[3].items() if k != "time/fps"} for f in flats]
    assert logs[0] == logs[1] and abs(logs[0]["train/explained_variance"]) < 10


def test_predict_is_deterministic_and_bounded():
    """SB3 ``predict(obs, deterministic=True)`` as evaluation harnesses call it (utils/evaluate.py:94): a = tanh(mean)"""
    from visfly_amd.envs import HoverEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    env = HoverEnv(num_agent_per_scene=256, seed=1, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=48, tensor_output=True)
    ppo = PPO(env, n_steps=16, batch_size=2048, n_epochs=1, seed=3, policy_kwargs=dict(activation_fn="relu"))
    a0, _ = ppo.predict(env.reset(), deterministic=True)
    a1, _ = ppo.predict(env.get_observation(), deterministic=True)
    assert torch.equal(a0, a1) and a0.shape == (256, 4) and float(a0.abs().max()) <= 1.0


def test_gather_rows_equals_index_select():
    """vf_gather_rows: all rollout-buffer fields of an epoch's permutation in one launch == torch.index_select per field"""
    _lib, lib = L()
    g = torch.Generator(device=DEV).manual_seed(3)
    rows = 70001
    fields = [torch.randn((rows, w), device=DEV, generator=g) if w > 1 else torch.randn(rows, device=DEV, generator=g) for w in (13, 3, 4, 1, 1, 1)]
    perm = torch.randperm(rows, device=DEV, generator=g)
    outs = [torch.empty_like(f) for f in fields]
    gf = _lib.GatherFields()
    gf.n_fields = len(fields)
    for i, (f, o) in enumerate(zip(fields, outs)):
        gf.width[i], gf.src[i], gf.dst[i] = (f.shape[1] if f.dim() == 2 else 1), f.data_ptr(), o.data_ptr()
    _lib.check(lib.vf_gather_rows(C.byref(gf), perm.data_ptr(), rows, st()))
    for f, o in zip(fields, outs):
        assert torch.equal(o, f.index_select(0, perm))


def test_adam_takes_the_grad_norm_from_the_fold_partials():
    """vf_mlp_weight_grad_sumsq leaves sum(grad^2) as per-block fp64 partials; vf_adam_step (sumsq_partials) adds them
    (+ the log_std tail) itself: same clipped update as with a separate vf_sumsq launch"""
    from visfly_amd.ppo import MlpPolicy
    _lib, lib = L()
    B = 4096
    pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], DEV, seed=5)
    g = torch.Generator(device=DEV).manual_seed(1)
    obs = {"state": torch.randn((B, 13), device=DEV, generator=g), "target": torch.randn((B, 3), device=DEV, generator=g)}
    mean, _ = pol.forward(obs)
    actions = torch.tanh(mean + torch.randn((B, 4), device=DEV, generator=g)).contiguous()
    old_lp, adv, ret = torch.randn(B, device=DEV, generator=g), torch.randn(B, device=DEV, generator=g), torch.randn(B, device=DEV, generator=g)
    stats, scratch = torch.zeros(16, device=DEV), torch.zeros(16 * 1024, device=DEV)
    cfg = _lib.PpoLossCfg(0.2, 0.0, 0.5, 1.0 / B, pol.grad.data_ptr() + 4 * pol.log_std_off, None)
    assert pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch) is True
    stats0, grad0 = stats.clone(), pol.grad.clone()
    stats.zero_()
    sq, nb = pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch, want_sumsq=True)
    # the loss-statistic rows folded by the weight-gradient fold launch: same order, same bits
    assert torch.equal(stats, stats0) and torch.equal(pol.grad, grad0)
    total = float(sq[:nb].sum() + (pol.grad[pol.log_std_off:].double() ** 2).sum())
    assert abs(total - float((pol.grad.double() ** 2).sum())) <= 1e-9 * total
    assert total ** 0.5 > 0.05                                   # clipping at 0.05 is active
    outs = []
    for use_partials in (True, False):
        p, m, v = pol.flat.clone(), torch.zeros_like(pol.flat), torch.zeros_like(pol.flat)
        ss = torch.zeros(1, device=DEV)
        if not use_partials:
            _lib.check(lib.vf_sumsq(pol.grad.data_ptr(), pol.n_params, ss.data_ptr(), scratch.data_ptr(), st()))
        acfg = _lib.AdamCfg(1e-3, 0.9, 0.999, 1e-8, 1e-5, 0.05, 1, 0, None, None, sq.data_ptr() if use_partials else None,
                            nb if use_partials else 0, pol.log_std_off)
        _lib.check(lib.vf_adam_step(p.data_ptr(), pol.grad.data_ptr(), m.data_ptr(), v.data_ptr(), pol.n_params, ss.data_ptr(),
                                    C.byref(acfg), st()))
        outs.append(p)
    assert torch.allclose(outs[0], outs[1], rtol=1e-6, atol=1e-9) and not torch.equal(outs[0], pol.flat)


def test_action_head_fused_into_the_chain_kernels():
    """vf_mlp_forward_act / vf_mlp_backward_data_act == vf_mlp_forward + vf_reparam_fwd / vf_reparam_bwd + vf_mlp_backward_data,
    bit for bit (same arithmetic on the head, in the kernel's registers)"""
    from visfly_amd.ppo import MlpPolicy
    _lib, lib = L()
    M = 1000
    pol = MlpPolicy({"state": 13}, {"state": [128, 64]}, [64, 64], [64, 64], DEV, seed=8, log_std_init=-0.7)
    g = torch.Generator(device=DEV).manual_seed(2)
    obs = {"state": torch.randn((M, 13), device=DEV, generator=g)}
    eps = torch.randn((M, 4), device=DEV, generator=g)
    d_action = torch.randn((M, 4), device=DEV, generator=g) / M
    pol.reserve_slots(M, 2)
    # separate launches on slot 0
    mean, _ = pol.forward(obs, slot=0, need_value=False)
    a0 = torch.empty((M, 4), device=DEV)
    _lib.check(lib.vf_reparam_fwd(mean.data_ptr(), pol.log_std.data_ptr(), eps.data_ptr(), a0.data_ptr(), M, st()))
    dm0, gl0 = torch.empty((M, 4), device=DEV), torch.zeros((M, 4), device=DEV)
    _lib.check(lib.vf_reparam_bwd(d_action.data_ptr(), a0.data_ptr(), pol.log_std.data_ptr(), eps.data_ptr(), dm0.data_ptr(), gl0.data_ptr(), M, st()))
    din0 = pol.backward_data(dm0, slot=0)["state"].clone()
    # fused on slot 1
    a1 = torch.empty((M, 4), device=DEV)
    assert pol.forward_act(obs, eps, a1, slot=1)
    dm1, gl1 = torch.empty((M, 4), device=DEV), torch.zeros((M, 4), device=DEV)
    din1 = pol.backward_data_act(d_action, a1, eps, gl1, dm1, slot=1)["state"]
    assert torch.equal(a0, a1) and torch.equal(dm0, dm1) and torch.equal(gl0, gl1) and torch.equal(din0, din1)
    for name in ("x:state:0", "feat", "pi:0", "pi:1", "g:feat", "g:pi:1"):
        assert torch.equal(pol._buffers(M, 0)[name], pol._buffers(M, 1)[name]), name
    assert torch.equal(pol._buffers(M, 1)["obs:state"], obs["state"])       # the slot's observation copy, written by the kernel


@pytest.mark.parametrize("env_name", ["NavigationEnv", "HoverEnv"])
def test_deferred_bootstrap_equals_per_step_bootstrap(env_name):
    """the TimeLimit bootstrap valued once per rollout over the compact list of truncated rows (vf_rollout_post_collect +
    vf_bootstrap_scatter) leaves exactly the rollout buffer that the per-step second forward + vf_rollout_post leaves"""
    import visfly_amd.envs as E
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    bufs = []
    for defer in (True, False):
        kw = {}
        if env_name == "NavigationEnv":       # its default spawn box is the origin, i.e. inside the floor
            kw["random_kwargs"] = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
        env = getattr(E, env_name)(num_agent_per_scene=3000, seed=5, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=11,
                                   tensor_output=True, **kw)
        ppo = PPO(env, n_steps=48, batch_size=4096, n_epochs=1, seed=2, policy_kwargs=dict(activation_fn="relu"))
        ppo.defer_bootstrap = defer
        ppo.collect_rollouts()
        torch.cuda.synchronize()
        if defer:
            cnt = int(ppo._boot["cursor"].item())
            assert 3000 * 3 <= cnt <= ppo._boot["cap"], cnt            # every agent is truncated ~4 times in 48 steps of 11-step episodes
        bufs.append({k: getattr(ppo.buf, k).clone() for k in ("rewards", "values", "advantages", "returns", "episode_starts", "log_probs")})
        bufs[-1]["_ep_stats"] = ppo._ep_stats.clone()
        env.close()
    ep0, ep1 = bufs[0].pop("_ep_stats"), bufs[1].pop("_ep_stats")          # episodes, sum of returns, sum of lengths, successes
    assert ep0[0] == ep1[0] and ep0[2] == ep1[2] and ep0[3] == ep1[3] and ep0[0] >= 3000 * 3
    assert torch.allclose(ep0[1], ep1[1], rtol=1e-6)                        # per-agent fp32 sums folded once vs fp64 sums per step
    for k in bufs[0]:
        assert torch.equal(bufs[0][k], bufs[1][k]), k
    assert float((bufs[0]["rewards"].abs() > 0).float().mean()) > 0.5


@pytest.mark.parametrize("env_name,N,dyn", [("NavigationEnv", 3000, "euler"), ("HoverEnv", 1000, "euler"), ("NavigationEnv", 16500, "euler"),
                                            ("HoverEnv", 16401, "euler"), ("NavigationEnv", 3000, "rk4_drag"), ("HoverEnv", 16401, "rk4"),
                                            ("HoverEnv", 1000, "euler_nodelay"), ("NavigationEnv", 16500, "rk4_nodelay"),
                                            ("HoverEnv2", 3000, "euler"), ("NavigationEnv2", 3000, "euler"), ("NavigationEnv2", 16500, "rk4"),
                                            ("RacingEnv", 3000, "euler"), ("RacingEnv2", 3000, "euler"), ("RacingEnv2", 16401, "rk4")])
def test_persistent_rollout_equals_the_per_step_loop(env_name, N, dyn):
    _persistent_rollout_vs_loop(env_name, N, dyn)


@pytest.mark.parametrize("env_name,N,dyn,net", [("NavigationEnv", 3000, "euler", "verdict"), ("HoverEnv", 16401, "rk4_nodelay", "one_layer_extractor"),
                                                ("NavigationEnv2", 3000, "euler", "one_layer_extractor"),       # (Nav2: the target is inside the one "state" row)
                                                ("RacingEnv", 3000, "euler", "one_layer_extractor"),
                                                ("RacingEnv2", 3000, "euler", "one_layer_extractor")])          # (r06: 16 gate-relative columns: kernel-side kind VF_ENV_RACING2)
def test_persistent_rollout_of_a_generated_class_equals_the_per_step_loop(env_name, N, dyn, net):
    """r05: a network shape without a built-in chain class gets its roll-out launch as one more plugin, compiled on first use for the env
    kind / action type / integrator / motor-lag setting (visfly_amd/_jit.py: ensure_rollout; csrc/vf_ppo_rollout_kernel.hpp) -- the same
    bit-for-bit comparison with the launch-by-launch loop as for the built-in classes, and no fallback warning"""
    import warnings
    from visfly_amd import _jit, _lib
    _, ext, pi, vf = _jit.PREBUILD[net]
    pk = dict(features_extractor_class="StateTargetExtractor" if "target" in ext else "StateExtractor", activation_fn="ReLU",
              features_extractor_kwargs=dict(net_arch={k: dict(layer=v) for k, v in ext.items()}), net_arch=dict(pi=pi, vf=vf))
    n0 = _lib.lib().vf_chain_plugin_launches()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _persistent_rollout_vs_loop(env_name, N, dyn, policy_kwargs=pk)
    assert _lib.lib().vf_chain_plugin_launches() > n0


def _persistent_rollout_vs_loop(env_name, N, dyn, policy_kwargs=None):
    """collect_rollouts as ONE launch (vf_ppo_rollout: 16 / 32 agents per wave for all n_steps, the same rows-per-wave chain
    vf_mlp_forward picks for N rows) leaves the rollout buffer, the TimeLimit list, the episode statistics, the episode
    outputs and the slab of the launch-by-launch loop, bit for bit -- over two consecutive rollouts with a training pass
    between them (Philox counters, delay-ring phase and spawn counters carry over).  r04: also with the RK4 integrator and per-agent
    drag randomisation re-drawn at every re-spawn (BASELINE configs[2]'s dynamics, utils/maths.py:353-386, dynamics.py:244-267)"""
    import visfly_amd.envs as E
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    dkw = dict(ENV_DYN)
    if dyn.startswith("rk4"):
        dkw["integrator"] = "rk4"
    if dyn.endswith("drag"):
        dkw["drag_random"] = 0.5
    if dyn.endswith("nodelay"):            # r05: no motor lag (envs/base/dynamics.py:534-554) has its persistent instances too
        dkw["ctrl_delay"] = False
    res = []
    for fused in (True, False):
        kw = {}
        if env_name == "NavigationEnv":
            kw["random_kwargs"] = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
        # r05: the *2 variants (relative-position observation rows, NavigationEnv2's reward) run on the persistent launch too
        env = getattr(E, env_name)(num_agent_per_scene=N, seed=5, dynamics_kwargs=dict(dkw), device=DEV, max_episode_steps=7,
                                   tensor_output=True, **kw)
        ppo = PPO(env, n_steps=20, batch_size=N * 20 // (4 if N < 16000 else 20), n_epochs=1, seed=2,
                  policy_kwargs=policy_kwargs or dict(activation_fn="relu"))
        ppo.fused_rollout = fused
        out = {}
        for rnd in range(2):
            ppo.collect_rollouts()
            torch.cuda.synchronize()
            assert ppo.fused_rollout is fused                 # the kernel was not refused
            for k in ("rewards", "values", "advantages", "returns", "episode_starts", "log_probs", "actions"):
                out[f"{rnd}:{k}"] = getattr(ppo.buf, k).clone()
            for k in ppo.obs_keys:
                out[f"{rnd}:obs:{k}"] = ppo.buf.obs[k].clone()
                out[f"{rnd}:last:{k}"] = ppo._last_obs[k].clone()
            cnt = int(ppo._boot["cursor"].item())
            order = torch.argsort(ppo._boot["idx"][:cnt])
            out[f"{rnd}:boot_idx"] = ppo._boot["idx"][:cnt][order].clone()
            out[f"{rnd}:boot_rows"] = ppo._boot["rows0"][:cnt][order].clone()
            assert cnt >= N * 2
            out[f"{rnd}:stat"] = ppo._boot["stat"].clone()
            sl = env.state_slab                               # padded to whole workgroups; pad agents are nobody's state
            out[f"{rnd}:slab"] = sl.transpose(-3, -4).reshape(sl.shape[-3], -1, 4)[:, :N].clone()
            out[f"{rnd}:starts"] = ppo._last_starts.clone()
            out[f"{rnd}:ep"] = torch.stack([env._ep_return, env._ep_length.float(), env._ep_flags.float()]).clone()
            if env._OBS_W == 13:      # (RacingEnv2, r06: the launch's terminal rows are its own 16-column buffer; the truncated ones are compared as boot_rows)
                out[f"{rnd}:terminal"] = env._terminal_obs.clone()
            if rnd == 0:
                ppo.train()
        out["params"] = ppo.policy.flat.clone()
        res.append(out)
        env.close()
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
    assert float((res[0]["1:rewards"].abs() > 0).float().mean()) > 0.5


@pytest.mark.parametrize("n_steps", [1, 2])
def test_persistent_rollout_shortest_horizons(n_steps):
    """n_steps = 1 / 2: the observation row of the last step goes to the env's own buffer, not to a buffer row"""
    from visfly_amd.envs import HoverEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    res = []
    for fused in (True, False):
        env = HoverEnv(num_agent_per_scene=777, seed=3, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=3, tensor_output=True)
        ppo = PPO(env, n_steps=n_steps, batch_size=777 * n_steps, n_epochs=1, seed=2, policy_kwargs=dict(activation_fn="relu"))
        ppo.fused_rollout = fused
        for _ in range(5):
            ppo.collect_rollouts()
        torch.cuda.synchronize()
        assert ppo.fused_rollout is fused
        res.append({k: getattr(ppo.buf, k).clone() for k in ("rewards", "values", "advantages", "returns", "episode_starts", "log_probs", "actions")})
        res[-1]["obs"], res[-1]["last"] = ppo.buf.obs["state"].clone(), ppo._last_obs["state"].clone()
        env.close()
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


def test_persistent_rollout_declines_what_it_has_no_kernel_for():
    """what is left without a persistent instance after r05: per-agent wind rows (a host-side wind function, dynamics.py:270-317).
    vf_ppo_rollout answers VF_EUNSUPPORTED / collect_rollouts does not ask, the loop steps launch by launch and keeps working."""
    from visfly_amd.envs import HoverEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    env = HoverEnv(num_agent_per_scene=512, seed=3, dynamics_kwargs=dict(ENV_DYN, wind_settings=["0.3 - 0.05*x", "0.02*x*x", "0.5*y + 0.1", "0*x + 0.125", "-0.01*x", "0.25*y - 0.05"]), device=DEV,
                   max_episode_steps=5, tensor_output=True)
    ppo = PPO(env, n_steps=8, batch_size=2048, n_epochs=1, seed=2, policy_kwargs=dict(activation_fn="relu"))
    used = []
    inner = env.collect_policy
    env.collect_policy = lambda *a, **k: used.append(inner(*a, **k)) or used[-1]
    ppo.collect_rollouts()
    ppo.collect_rollouts()
    torch.cuda.synchronize()
    assert used and all(u is False for u in used), used
    assert torch.isfinite(ppo.buf.advantages).all() and float(ppo._ep_stats[0]) >= 512
    env.close()


def test_ppo_on_a_host_observation_env_values_its_own_terminal_rows():
    """RacingEnv2 hands the policy 16 gate-relative columns assembled on the host: the TimeLimit bootstrap must value THOSE rows
    (env._terminal_state_rows()), not the kernel's raw 13-column terminal state (the width was hard-coded)"""
    from visfly_amd.envs import RacingEnv2
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    env = RacingEnv2(num_agent_per_scene=512, seed=4, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=9, tensor_output=True)
    ppo = PPO(env, n_steps=24, batch_size=2048, n_epochs=1, seed=1, policy_kwargs=dict(activation_fn="relu"))
    assert ppo.policy.obs_dims["state"] == 16
    ppo.fused_rollout = False                 # (the launch-by-launch loop; the persistent roll-out forms its own 16-column terminal rows)
    ppo.collect_rollouts()
    torch.cuda.synchronize()
    rows = env._terminal_state_rows()
    assert rows.shape == (512, 16)
    assert float(ppo._ep_stats[0]) >= 512 * 2                     # every agent is truncated at least twice in 24 steps
    assert torch.isfinite(ppo.buf.rewards).all() and torch.isfinite(ppo.buf.advantages).all()
    ppo.train()
    assert torch.isfinite(ppo.policy.flat).all()
    env.close()


@pytest.mark.parametrize("keys", [("state",), ("state", "target")])
def test_chain16_and_chain32_agree_to_a_stated_bound(keys):
    """the 16-rows-per-wave forward (v_mfma_f32_16x16x4_f32, picked for M <= 16 384: roll-outs, BPTT) and the 32-row chain
    (v_mfma_f32_32x32x2_f32: larger M, and what vf_ppo_update recomputes a minibatch with) accumulate the same products in a different
    order.  Same rows through both: heads within 2e-6 relative of the output scale (a handful of fp32 ulps; measured ~4e-7) --
    so the log-prob recomputed by the first PPO minibatch differs from the rollout's by rounding only (ratio = 1 +- 1e-6)"""
    from visfly_amd.ppo import MlpPolicy
    dims = {"state": 13, "target": 3}
    pol = MlpPolicy({k: dims[k] for k in keys}, {k: [128, 64] for k in keys}, [64, 64], [64, 64], DEV, seed=6)
    g = torch.Generator(device=DEV).manual_seed(1)
    M16, M32 = 16384, 16384 + 4096
    obs32 = {k: torch.randn((M32, dims[k]), device=DEV, generator=g) for k in keys}
    obs16 = {k: v[:M16].contiguous() for k, v in obs32.items()}
    m16, v16 = [x.clone() for x in pol.forward(obs16, save_activations=False)]
    m32, v32 = [x.clone() for x in pol.forward(obs32, save_activations=False)]
    for a, b in ((m16, m32[:M16]), (v16.view(-1), v32.view(-1)[:M16])):
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        assert err <= 2e-6 * scale, (err, scale)
    ref = pol.to_torch().to(DEV).double()
    with torch.no_grad():
        m0, _ = ref({k: v.double() for k, v in obs16.items()})
    assert float((m16.double() - m0).abs().max()) <= 2e-6 * float(m0.abs().max())        # and both sit that close to fp64


@pytest.mark.parametrize("keys", [("state",), ("state", "target")])
def test_reverse_chain16_and_chain32_agree_to_a_stated_bound(keys):
    """the 16-rows-per-wave reverse chain (policy trunk + observation gradient, picked for M <= 16 384: BPTT shards) and the
    32-row one form the same sums in a different order: same rows through both -> observation gradients within 2e-6 of the
    output scale of each other and of torch's fp64 autograd (rows whose ReLU masks differ between the two forwards -- a
    pre-activation within rounding of zero -- excepted: fewer than 1 in 10 000)"""
    from visfly_amd.ppo import MlpPolicy
    dims = {"state": 13, "target": 3}
    pol = MlpPolicy({k: dims[k] for k in keys}, {k: [128, 64] for k in keys}, [64, 64], [64, 64], DEV, seed=8)
    g = torch.Generator(device=DEV).manual_seed(3)
    M16, M32 = 8192, 16384 + 2048
    obs32 = {k: torch.randn((M32, dims[k]), device=DEV, generator=g) for k in keys}
    obs16 = {k: v[:M16].contiguous() for k, v in obs32.items()}
    dm32 = torch.randn((M32, 4), device=DEV, generator=g)
    dm16 = dm32[:M16].contiguous()
    pol.reserve_slots(M32, 1)
    pol.reserve_slots(M16, 1)
    pol.forward(obs32, slot=0, need_value=False)
    if not pol.backward_data_supported(M32):
        pytest.skip("register-chained reverse pass switched off")
    g32 = {k: v.clone() for k, v in pol.backward_data(dm32, slot=0).items()}
    pol.forward(obs16, slot=0, need_value=False)
    g16 = {k: v.clone() for k, v in pol.backward_data(dm16, slot=0).items()}
    ref = pol.to_torch().to(DEV).double()
    x = {k: v.double().requires_grad_(True) for k, v in obs16.items()}
    mean, _ = ref(x)
    (mean * dm16.double()).sum().backward()
    for k in keys:
        want = x[k].grad
        scale = float(want.abs().max())
        for got in (g16[k], g32[k][:M16]):
            bad = ((got.double() - want).abs().amax(dim=1) > 2e-6 * scale).float().mean()
            assert float(bad) < 1e-4, (k, float(bad))
        bad = ((g16[k] - g32[k][:M16]).abs().amax(dim=1) > 2e-6 * scale).float().mean()
        assert float(bad) < 1e-4, (k, float(bad))

