"""Register-chained kernels for ANY layer list of the reference's policies (utils/policies/extractors.py:376-449 `create_mlp` of any depth per
observation key; `net_arch=dict(pi=[..], vf=[..])` of any depth): shapes libvisfly_amd.so holds no instance of are compiled on first use
(visfly_amd/_jit.py -> csrc/vf_mlp_chain_gen.hpp -> a plugin registered with vf_chain_plugin_load).  CPU: the shape rules, the generated
source, that the plugin builds, loads and is refused when it is not one.  GPU: forward / reverse chain / fused PPO step of generated
classes against torch autograd on the same weights and against the block-tile kernels, as test_chain_backward_vs_torch_and_block_tile_kernel
does for the built-in classes -- and that it IS the plugin that ran (vf_chain_plugin_launches)."""
import ctypes as C
import os
import warnings

import pytest
import torch

DEV = "cuda:0"
from visfly_amd._jit import PREBUILD as SHAPES      # name -> (observation widths, extractor layers, pi, vf); __graft_entry__.build() compiles them


def test_shape_rules_and_generated_source():
    from visfly_amd import _jit
    dims, ext, pi, vf = SHAPES["verdict"]
    sh = _jit.shape_of(dims, ext, pi, vf)
    assert sh == ((16, 8), ((4, 2), (4, 2)), (4, 4), (1,)) and not _jit.is_builtin(sh)
    assert _jit.name_of(sh) == "in16[128,64] in8[128,64] pi[128,128] vf[32]"
    # the YAML-default shapes are the library's own classes: nothing to compile
    assert _jit.is_builtin(_jit.shape_of(dims, ext, [64, 64], [64, 64]))
    assert _jit.is_builtin(_jit.shape_of({"state": 13}, {"state": [128, 64]}, [64, 64], [64, 64]))
    # what stays on the block-tile kernels: widths off the 32 grid or above 128, empty trunks, > 4 layers, SAC heads, pass-through inputs
    assert _jit.shape_of(dims, ext, [48], [32]) is None and _jit.shape_of(dims, ext, [160], [32]) is None
    assert _jit.shape_of(dims, ext, [], [32]) is None and _jit.shape_of(dims, ext, [32] * 5, [32]) is None
    assert _jit.shape_of(dims, ext, pi, vf, head_dims=(1, 1)) is None and _jit.shape_of(dims, ext, pi, vf, passthrough=("action",)) is None
    # the twin critic (r06): heads (1, 1) + ONE pass-through input of <= 4 columns behind ONE observation branch; the default widths built in
    cd, ce = {"state": 13, "action": 4}, {"state": [64, 64, 32]}
    csh = _jit.shape_of(cd, ce, [32], [96, 32], head_dims=(1, 1), passthrough=("action",))
    assert csh == ((16,), ((2, 2, 1),), (1,), (3, 1), (1, 1)) and "heads 1/1" in _jit.name_of(csh) and "static constexpr int PASS = 1;" in _jit.source(csh)
    assert "static constexpr int PASS = 0;" in _jit.source(sh) and _jit.path_of(csh) != _jit.path_of(csh[:4])
    assert _jit.is_builtin(_jit.shape_of(cd, {"state": [128, 64]}, [64, 64], [64, 64], head_dims=(1, 1), passthrough=("action",)))
    assert _jit.shape_of({**dims, "action": 4}, ext, pi, vf, head_dims=(1, 1), passthrough=("action",)) is None       # two branches: 132 columns
    assert _jit.shape_of({"state": 13, "action": 6}, ce, [32], [32], head_dims=(1, 1), passthrough=("action",)) is None
    # the SAC-style Actor (two 4-wide heads): a class of its own, the default widths built in
    assert _jit.shape_of(dims, ext, pi, vf, head_dims=(4, 4)) == sh + ((4, 4),) and "heads 4/4" in _jit.name_of(sh + ((4, 4),))
    assert "static constexpr int HM = 4, HV = 4;" in _jit.source(sh + ((4, 4),)) and "static constexpr int HM = 4, HV = 1;" in _jit.source(sh)
    assert _jit.is_builtin(_jit.shape_of(dims, ext, [64, 64], [64, 64], head_dims=(4, 4)))
    assert _jit.shape_of({"state": 40}, {"state": [64]}, [32], [32]) is None
    src = _jit.source(sh)
    assert "static constexpr int EW[2][4] = {{4, 2, 0, 0}, {4, 2, 0, 0}};" in src and "static constexpr int PW[4] = {4, 4, 0, 0};" in src
    assert "VF_CHAIN_PLUGIN_DEFINE(Net, NetPi" in src
    assert _jit.path_of(sh) == _jit.path_of(_jit.shape_of(dims, ext, pi, vf)) and _jit.path_of(sh) != _jit.path_of(_jit.shape_of(dims, ext, pi, [64]))


def test_plugin_builds_loads_and_is_checked(tmp_path):
    """hipcc cross-compiles the plugin without a GPU (cached: __graft_entry__.build() made it); the registry takes it once, refuses a
    shared object that is not a plugin"""
    from visfly_amd import _jit, _lib
    lib = _lib.lib()
    sh = _jit.shape_of(*SHAPES["verdict"])
    path = _jit.build(sh)
    assert os.path.exists(path) and os.path.dirname(path) == _jit.JIT_DIR
    n0 = lib.vf_chain_plugin_count()
    names0 = [lib.vf_chain_plugin_name(i) for i in range(n0)]
    _lib.check(lib.vf_chain_plugin_load(path.encode()))
    _lib.check(lib.vf_chain_plugin_load(path.encode()))
    n1 = lib.vf_chain_plugin_count()
    assert n1 == n0 + (0 if _jit.name_of(sh).encode() in names0 else 1)
    assert _jit.name_of(sh).encode() in [lib.vf_chain_plugin_name(i) for i in range(n1)]
    assert lib.vf_chain_plugin_name(n1) is None
    assert lib.vf_chain_plugin_load(str(tmp_path / "missing.so").encode()) != 0 and b"vf_chain_plugin_load" in lib.vf_last_error()
    assert lib.vf_chain_plugin_load(_lib.LIB.encode()) != 0 and b"not a chain plugin" in lib.vf_last_error()
    assert lib.vf_chain_plugin_count() == n1


def test_bptt_and_critic_plugins_build_and_load():
    """r06: the two further plugin kinds cross-compile without a GPU and register: a twin-critic class (PASS = 1, heads 1 / 1) and a BPTT
    plugin (one instance each of k_bptt_rollout / k_bptt_reverse for a generated actor class under one env configuration); distinct cache
    files per configuration"""
    from visfly_amd import _jit, _lib
    lib = _lib.lib()
    csh = _jit.shape_of(*_jit.PREBUILD_CRITIC["critic_hover"])
    ash = _jit.shape_of(*_jit.PREBUILD_SAC["sac_hover"])
    cfg = ("bptt", 0, 1, 0, True)
    src = _jit.bptt_source(ash, cfg[1:])
    assert "VF_CHAIN_PLUGIN_BPTT_DEFINE(Net, NetPi, 0, 1, 0, true," in src and '#include "vf_bptt_reverse_kernel.hpp"' in src
    assert _jit.path_of(ash, cfg) != _jit.path_of(ash, ("bptt", 1, 1, 0, True)) != _jit.path_of(ash) and "_bptt0101_" in _jit.path_of(ash, cfg)
    for path in (_jit.build(csh), _jit.build(ash, rollout=cfg)):
        assert os.path.exists(path)
        _lib.check(lib.vf_chain_plugin_load(path.encode()))
    names = [lib.vf_chain_plugin_name(i) for i in range(lib.vf_chain_plugin_count())]
    assert _jit.name_of(csh).encode() in names and any(b"BPTT horizon kind 0 act 1 int 0 delay 1" in n for n in names)
    # the BPTT plugin is for ACTOR classes: not for the critic's shape, not for a built-in one
    assert _jit.ensure_bptt(csh, cfg[1:]) is False and _jit.ensure_bptt(_jit.shape_of({"state": 13}, {"state": [128, 64]}, [64, 64], [64, 64], (4, 4)), cfg[1:]) is False


def make(name, **kw):
    from visfly_amd.ppo import MlpPolicy
    dims, ext, pi, vf = SHAPES[name]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pol = MlpPolicy(dims, ext, pi, vf, DEV, seed=9, **kw)
    assert pol.chain_jit, "the shape has a generated chain class"
    return pol, dims


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 33, 777, 25600])
@pytest.mark.parametrize("name", list(SHAPES))
def test_generated_forward_vs_torch_and_block_tile_kernel(name, M):
    from visfly_amd import _lib
    lib = _lib.lib()
    pol, dims = make(name)
    g = torch.Generator(device=DEV).manual_seed(M)
    obs = {k: torch.randn((M, d), device=DEV, generator=g) for k, d in dims.items()}
    ref = pol.to_torch().double().to(DEV)
    m0, v0 = ref({k: v.double() for k, v in obs.items()})
    n0 = lib.vf_chain_plugin_launches()
    mean, value = pol.forward(obs)
    mean, value = mean.clone(), value.clone()
    assert lib.vf_chain_plugin_launches() == n0 + 1, "the plugin served the forward"
    mean_pi, none = pol.forward(obs, save_activations=False, need_value=False)
    assert none is None and lib.vf_chain_plugin_launches() == n0 + 2 and torch.equal(mean_pi, mean), "policy-only class: same mean, bit for bit"
    sc = max(m0.abs().max().item(), v0.abs().max().item(), 1e-3)
    assert (mean.double() - m0).abs().max().item() <= 2e-6 * sc and (value.view(-1).double() - v0.view(-1)).abs().max().item() <= 2e-6 * sc
    pol.fused = False                               # layer by layer on the block-tile kernels
    m1, v1 = pol.forward(obs)
    assert lib.vf_chain_plugin_launches() == n0 + 2
    assert (m1 - mean).abs().max().item() <= 4e-6 * sc and (v1.view(-1) - value.view(-1)).abs().max().item() <= 4e-6 * sc
    # every saved activation (what the weight gradients read) equals the layer-by-layer path's
    b1 = {k: v.clone() for k, v in pol._buffers(M, 0).items() if isinstance(v, torch.Tensor) and k.split(":")[0] in ("x", "pi", "vf", "feat")}
    pol.fused = True
    pol.forward(obs)
    b2 = pol._buffers(M, 0)
    assert b1, "hidden activations are kept"
    for k, v in b1.items():
        assert (b2[k] - v).abs().max().item() <= 4e-6 * max(v.abs().max().item(), 1e-3), k


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 33, 777, 25600])
@pytest.mark.parametrize("mode", ["ppo", "bptt"])
@pytest.mark.parametrize("name", list(SHAPES))
def test_generated_backward_vs_torch_and_block_tile_kernel(name, mode, M):
    """the reverse chain of a generated class, the two variants the trainers use (both trunks without / policy trunk with the observation
    gradient) + the row-slab weight gradients, against torch autograd and the layer-by-layer kernels; deterministic; accumulate mode"""
    from visfly_amd import _lib
    lib = _lib.lib()
    pol, dims = make(name)
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
        n0 = lib.vf_chain_plugin_launches()
        d_in = pol.backward(d_mean, d_value, None, need_input_grad=ig)
        assert (lib.vf_chain_plugin_launches() > n0) == fused, "the plugin's reverse chain ran (and only when asked)"
        if fused and fused in res:
            assert torch.equal(res[True][0], pol.grad)
        res[fused] = (pol.grad.clone(), {k: v.clone() for k, v in d_in.items()})
    scale = gref.abs().max().item()
    for fused in (True, False):
        gk = res[fused][0].clone()
        gk[pol.log_std_off:] = 0
        assert (gk - gref).abs().max().item() <= 5e-6 * scale, (fused, (gk - gref).abs().max().item(), scale)
        for k, v in res[fused][1].items():
            assert torch.allclose(v, xs[k].grad, rtol=1e-4, atol=1e-6 * xs[k].grad.abs().max().item())
    pol.fused_backward = True
    pol.forward(obs)
    pol.backward(d_mean, d_value, None, accumulate=True, need_input_grad=ig)
    assert torch.allclose(pol.grad, 2 * res[True][0], rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [25600, 1000, 33])
@pytest.mark.parametrize("name", list(SHAPES))
def test_generated_fused_ppo_update_equals_separate_launches(name, B):
    """vf_ppo_update on a generated class (k_ppo_update_chain<ChainNetG<..>>: forward + loss + reverse chain in one launch) vs forward /
    vf_ppo_loss / backward"""
    from visfly_amd import _lib
    from test_ppo_gpu import sb3_squashed_log_prob
    lib = _lib.lib()
    pol, dims = make(name, log_std_init=-0.3)
    g = torch.Generator(device=DEV).manual_seed(B)
    obs = {k: torch.randn((B, d), device=DEV, generator=g) for k, d in dims.items()}
    mean, value = pol.forward(obs)
    actions = torch.tanh(mean + 0.7 * torch.randn((B, 4), device=DEV, generator=g)).contiguous()
    old_lp = sb3_squashed_log_prob(mean, pol.log_std, actions) + 0.3 * torch.randn(B, device=DEV, generator=g)
    adv, ret = torch.randn(B, device=DEV, generator=g), torch.randn(B, device=DEV, generator=g)
    scratch = torch.zeros(16 * max(1024, (B + 31) // 32), device=DEV)
    st = _lib.current_stream(torch.device(DEV))
    res = {}
    for fused in (True, False):
        stats = torch.zeros(16, device=DEV)
        pol.grad.fill_(3.0)
        cfg = _lib.PpoLossCfg(0.2, 0.01, 0.5, 1.0 / B, pol.grad.data_ptr() + 4 * pol.log_std_off, None)
        if fused:
            n0 = lib.vf_chain_plugin_launches()
            assert pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, stats, scratch)
            assert lib.vf_chain_plugin_launches() > n0
        else:
            m, v = pol.forward(obs)
            d_mean, d_value = torch.empty((B, 4), device=DEV), torch.empty(B, device=DEV)
            _lib.check(lib.vf_ppo_loss(m.data_ptr(), v.data_ptr(), pol.log_std.data_ptr(), actions.data_ptr(), old_lp.data_ptr(),
                                       adv.data_ptr(), ret.data_ptr(), d_mean.data_ptr(), d_value.data_ptr(), stats.data_ptr(), B,
                                       C.byref(cfg), scratch.data_ptr(), st))
            pol.backward(d_mean, d_value, None)
        res[fused] = (pol.grad.clone(), stats.clone())
    (g1, s1), (g0, s0) = res[True], res[False]
    assert torch.allclose(s1[:9], s0[:9], rtol=2e-5, atol=1e-6 * max(1.0, s0[:9].abs().max().item())), (s1, s0)
    scale = g0.abs().max().item()
    assert (g1 - g0).abs().max().item() <= 5e-6 * scale, ((g1 - g0).abs().max().item(), scale)
    assert torch.allclose(g1[pol.log_std_off:], g0[pol.log_std_off:], rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
def test_ppo_trains_a_non_default_net_arch_on_chain_kernels(monkeypatch):
    """`PPO(net_arch=dict(pi=[128, 128], vf=[32]))` trains without the block-tile warning, on the plugin's kernels;
    the same run with the compilation switched off (VISFLY_AMD_JIT=0: block-tile kernels, with the warning) ends at the same parameters
    up to fp32 summation order"""
    from visfly_amd import _lib
    from visfly_amd.envs import NavigationEnv
    from visfly_amd.ppo import PPO
    from _golden import ENV_DYN
    lib = _lib.lib()
    kw = dict(n_steps=16, batch_size=4096, n_epochs=2, learning_rate=3e-4, seed=3,
              policy_kwargs=dict(features_extractor_class="StateTargetExtractor",
                                 features_extractor_kwargs=dict(net_arch=dict(state=dict(layer=[128, 64]), target=dict(layer=[128, 64]))),
                                 net_arch=dict(pi=[128, 128], vf=[32]), activation_fn="ReLU"))
    flats = []
    for jit in (True, False):
        monkeypatch.setenv("VISFLY_AMD_JIT", "1" if jit else "0")
        lib.vf_chain_plugin_set_enabled(1 if jit else 0)       # (the registry is per process: the first run's plugin would serve the second)
        env = NavigationEnv(num_agent_per_scene=1024, seed=1, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=64, tensor_output=True)
        n0 = lib.vf_chain_plugin_launches()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            ppo = PPO(env, **kw)
            assert ppo.policy.spec["pi"] == [128, 128] and ppo.policy.spec["vf"] == [32] and ppo.policy.chain_jit == jit
            ppo.learn(16 * 1024 * 2)
        torch.cuda.synchronize()
        tile = [x for x in w if "block-tile" in str(x.message)]
        if jit:
            assert not tile, [str(x.message) for x in tile]
            # 2 iterations x (the roll-out as ONE launch of the class's roll-out plugin + 2 epochs x 4 fused minibatch steps)
            assert lib.vf_chain_plugin_launches() - n0 >= 2 * (1 + 8)
            assert not [x for x in w if "vf_ppo_rollout" in str(x.message)], "no launch-by-launch fallback of the roll-out either"
        else:
            assert tile and lib.vf_chain_plugin_launches() == n0
        flats.append(ppo.policy.flat.clone())
        env.close()
    lib.vf_chain_plugin_set_enabled(1)
    d = (flats[0] - flats[1]).abs().max().item()
    assert d <= 2e-5, d


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ragged", "builtin_nav"])
def test_saved_activations_of_cold_launches(name):
    """r05 regression: the copies of the hidden activations that the chain kernels trickle out as 128-bit buffer stores.  With the column
    offset in the store's soffset SGPR the compiler let the next epilogue overwrite the data registers in the slot after the store, and
    on a COLD launch (freshly mapped buffers: the store's issue stalls) one float of 8 rows went out as the new register value -- once in
    ~3 launches of this shape (csrc/vf_mlp_chain.hpp: chain_buffer_store).  Fresh allocations every repetition."""
    from visfly_amd.ppo import MlpPolicy
    dims, ext, pi, vf = SHAPES[name] if name in SHAPES else ({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64])
    M = 25600
    for rep in range(12):
        torch.cuda.empty_cache()
        pol = MlpPolicy(dims, ext, pi, vf, DEV, seed=9)
        g = torch.Generator(device=DEV).manual_seed(M + rep)
        obs = {k: torch.randn((M, d), device=DEV, generator=g) for k, d in dims.items()}
        pol.forward(obs)
        b0 = {k: v.clone() for k, v in pol._buffers(M, 0).items() if isinstance(v, torch.Tensor) and k.split(":")[0] in ("x", "pi", "vf", "feat")}
        pol.fused = False
        pol.forward(obs)
        b1 = pol._buffers(M, 0)
        for k, v in b0.items():
            assert (v - b1[k]).abs().max().item() <= 4e-6 * max(b1[k].abs().max().item(), 1e-3), (rep, k)
        del pol, b0, b1


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 33, 777, 16384])
@pytest.mark.parametrize("ig", [True, False])
@pytest.mark.parametrize("name", ["sac_nav", "sac_hover", "sac_nav_bptt"])
def test_generated_sac_actor_vs_torch_and_block_tile_kernel(name, ig, M):
    """the reference's SAC-style Actor (utils/policies/td_policies.py:146-252: latent_pi -> mu, log_latent_pi -> log_std, two 4-wide heads; the
    actor of its BPTT and SHAC loops) on a NON-default shape: generated class, forward + reverse chain of both trunks with / without the
    observation gradient against an fp64 torch network on the same weights and against the block-tile kernels
    (test_sac_actor_chain_vs_torch_and_block_tile_kernel's checks for the built-in classes)"""
    from visfly_amd import _jit, _lib
    from visfly_amd.ppo import MlpPolicy
    lib = _lib.lib()
    dims, ext, pi, vf, heads = _jit.PREBUILD_SAC[name]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pol = MlpPolicy(dims, ext, pi, vf, DEV, seed=9, ortho_init=False, head_dims=heads, log_std_param=False)
    assert pol.chain_jit
    g = torch.Generator(device=DEV).manual_seed(M + 5)
    obs = {k: torch.randn((M, d), device=DEV, generator=g) for k, d in dims.items()}
    d_mu = torch.randn((M, 4), device=DEV, generator=g) / M
    d_ls = torch.randn((M, 4), device=DEV, generator=g) / M
    ref = pol.to_torch().double().to(DEV)
    xs = {k: v.double().requires_grad_(ig) for k, v in obs.items()}
    m0, v0 = ref(xs)
    ((m0 * d_mu.double()).sum() + (v0 * d_ls.double()).sum()).backward()
    gref = ref.flat_grad().to(DEV).float()
    n0 = lib.vf_chain_plugin_launches()
    mu, ls = pol.forward(obs)
    assert lib.vf_chain_plugin_launches() == n0 + 1 and mu.shape == ls.shape == (M, 4)
    sc = max(m0.abs().max().item(), v0.abs().max().item())
    assert (mu - m0.float()).abs().max().item() <= 2e-6 * sc and (ls - v0.float()).abs().max().item() <= 2e-6 * sc
    res = {}
    for fused in (True, False, True):
        pol.fused_backward = fused
        pol.grad.fill_(0.0)
        pol.forward(obs)
        n1 = lib.vf_chain_plugin_launches()
        d_in = pol.backward(d_mu, d_ls, None, need_input_grad=ig)
        assert (lib.vf_chain_plugin_launches() > n1) == fused
        if fused and fused in res:
            assert torch.equal(res[True][0], pol.grad)
        res[fused] = (pol.grad.clone(), {k: v.clone() for k, v in d_in.items()})
    scale = gref.abs().max().item()
    assert (res[True][0] - res[False][0]).abs().max().item() <= 5e-6 * scale
    for fused in (True, False):
        err = (res[fused][0] - gref).abs().max().item()
        assert err <= (1e-3 if M >= 16384 else 2e-6) * scale, (fused, err, scale)      # (a ReLU unit within rounding of 0 may flip vs fp64)
    for k in res[True][1]:
        assert torch.allclose(res[True][1][k], res[False][1][k], rtol=1e-4, atol=1e-6 * res[False][1][k].abs().max().item())
        want = xs[k].grad.float()
        bad = ((res[True][1][k] - want).abs() > 1e-4 * want.abs() + 1e-5 * want.abs().max()).any(dim=1)
        assert int(bad.sum()) <= (8 if M >= 16384 else 0), (k, int(bad.sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 33, 777, 16384, 40000])
@pytest.mark.parametrize("name", ["critic_hover", "critic_wide"])
def test_generated_twin_critic_vs_torch_and_block_tile_kernel(name, M):
    """r06: the reference's twin ContinuousCritic (utils/policies/td_policies.py:82-143: own extractor, th.cat([features, actions]) -> qf0 / qf1
    -> Q) on a NON-default shape: generated class with the pass-through action tile and two 1-wide heads (ChainNetG<Spec>, PASS = 1) -- forward
    and reverse chain against an fp64 torch network on the same weights and against the block-tile kernels it ran on until r05
    (test_twin_critic_chain_vs_torch_and_block_tile_kernel's checks for the built-in class); the saved feature rows carry the action columns"""
    from visfly_amd import _jit, _lib
    from visfly_amd.ppo import MlpPolicy
    lib = _lib.lib()
    dims, ext, pi, vf, heads, pas = _jit.PREBUILD_CRITIC[name]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pol = MlpPolicy(dims, ext, pi, vf, DEV, seed=11, ortho_init=False, head_dims=heads, passthrough=pas, log_std_param=False)
    assert pol.chain_jit
    g = torch.Generator(device=DEV).manual_seed(M + 7)
    obs = {k: torch.randn((M, d), device=DEV, generator=g) for k, d in dims.items()}
    obs["action"] = torch.tanh(obs["action"])
    d_q0 = torch.randn((M, 1), device=DEV, generator=g) / M
    d_q1 = torch.randn((M, 1), device=DEV, generator=g) / M
    ref = pol.to_torch().double().to(DEV)
    q0r, q1r = ref({k: v.double() for k, v in obs.items()})
    ((q0r * d_q0.double()).sum() + (q1r * d_q1.double()).sum()).backward()
    gref = ref.flat_grad().to(DEV).float()[:pol.n_params]
    n0 = lib.vf_chain_plugin_launches()
    q0, q1 = pol.forward(obs)
    assert lib.vf_chain_plugin_launches() == n0 + 1 and q0.shape == q1.shape == (M, 1)
    sc = max(q0r.abs().max().item(), q1r.abs().max().item())
    assert (q0 - q0r.float()).abs().max().item() <= 2e-6 * sc and (q1 - q1r.float()).abs().max().item() <= 2e-6 * sc
    nfeat = ext["state"][-1]
    assert torch.equal(pol._buffers(M, 0)["feat"][:, nfeat:], obs["action"])          # pass-through columns of the saved feature rows
    b = pol._buffers(M, 0)
    bd = pol._bwd_desc(b, M, d_q0, d_q1, False)[0]
    assert lib.vf_mlp_backward_data_supported(C.byref(bd)) == 1                        # the generated class IS what runs
    res = {}
    for fused in (True, False, True):
        pol.fused_backward = fused
        pol.grad.fill_(0.0)
        pol.forward(obs)
        n1 = lib.vf_chain_plugin_launches()
        pol.backward(d_q0, d_q1, None)
        assert (lib.vf_chain_plugin_launches() > n1) == fused
        if fused and fused in res:
            assert torch.equal(res[True], pol.grad)
        res[fused] = pol.grad.clone()
    scale = gref.abs().max().item()
    assert (res[True] - res[False]).abs().max().item() <= 5e-6 * scale
    for fused in (True, False):                      # vs fp64: ReLU-mask flips at large M (test_sac_actor_chain_vs_torch_and_block_tile_kernel)
        err = (res[fused] - gref).abs().max().item()
        assert err <= (1e-3 if M >= 16384 else 2e-6) * scale, (fused, err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1000, 4096 + 17, 40000])
@pytest.mark.parametrize("name", ["critic_hover", "critic_wide"])
def test_generated_fused_critic_step_equals_the_three_launch_step(name, M):
    """vf_twin_q_update on a generated twin-critic class (forward + twin-Q loss + reverse chain in one launch of the plugin's
    k_twin_q_update_chain) leaves the loss and the flat gradient of the forward / vf_twin_q_loss / backward path (same chain arithmetic;
    the masks come from registers instead of the saved activations) -- test_fused_critic_step_equals_the_three_launch_step for the built-in class"""
    from visfly_amd import _jit, _lib
    from visfly_amd.ppo import MlpPolicy, _ptr
    L = _lib.lib()
    st = _lib.current_stream(torch.device(DEV))
    dims, ext, pi, vf, heads, pas = _jit.PREBUILD_CRITIC[name]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        c = MlpPolicy(dims, ext, pi, vf, DEV, seed=3, ortho_init=False, head_dims=heads, passthrough=pas, log_std_param=False)
        g = torch.Generator(device=DEV).manual_seed(M)
        obs = {"state": torch.randn((M, 13), device=DEV, generator=g), "action": torch.tanh(torch.randn((M, 4), device=DEV, generator=g))}
        target = torch.randn(M, device=DEV, generator=g)
        out = {}
        for fused in (False, True):
            c.grad.fill_(0.0)
            loss = torch.empty(1, device=DEV)
            if fused:
                n0 = L.vf_chain_plugin_launches()
                assert c.twin_q_update(obs, target, loss, M), "the generated critic class must run on vf_twin_q_update"
                assert L.vf_chain_plugin_launches() == n0 + 1 and c._fused_twin_q is not False
            else:
                q0, q1 = c.forward(obs, save_activations=True)
                dq0, dq1 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
                scr = torch.empty(int(L.vf_twin_q_loss_scratch_doubles(M)), dtype=torch.float64, device=DEV)
                _lib.check(L.vf_twin_q_loss(_ptr(q0.view(-1)), _ptr(q1.view(-1)), _ptr(target), _ptr(dq0), _ptr(dq1), _ptr(loss), scr.data_ptr(), M, M, st))
                c.backward(dq0.view(M, 1), dq1.view(M, 1), None)
            out[fused] = (float(loss), c.grad[:c.n_params].clone())
    (l0, g0), (l1, g1) = out[False], out[True]
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0)), (l0, l1)
    scale = g0.abs().max().item()
    assert scale > 0 and (g1 - g0).abs().max().item() <= 2e-6 * scale, ((g1 - g0).abs().max().item(), scale)


@pytest.mark.gpu
def test_shac_with_a_non_default_net_arch_runs_its_critic_on_chain_kernels():
    """SHAC(net_arch=dict(pi=[32], qf=[32])) over a [64, 64, 32] extractor: actor AND twin critic have generated chain classes -- the critic
    updates are the plugin's fused step (no block-tile fallback warning from vf_twin_q_update), the horizons the actor class's persistent launches"""
    from visfly_amd import _lib
    from visfly_amd.shac import SHAC
    from visfly_amd.envs import HoverEnv2
    from _golden import ENV_DYN
    lib = _lib.lib()
    pk = dict(features_extractor_class="StateExtractor", features_extractor_kwargs={"net_arch": {"state": {"layer": [64, 64, 32]}}},
              net_arch=dict(pi=[32], qf=[32]), activation_fn="relu", share_features_extractor=False)
    env = HoverEnv2(num_agent_per_scene=1024, seed=5, dynamics_kwargs=dict(ENV_DYN), device=DEV, tensor_output=True, max_episode_steps=64,
                    requires_grad=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        algo = SHAC(env, policy="MultiInputPolicy", policy_kwargs=pk, horizon=8, learning_rate=1e-3, seed=1)
        assert algo.critic.chain_jit and algo.policy.chain_jit
        env.reset()
        n0 = lib.vf_chain_plugin_launches()
        algo.learn(2 * 8 * 1024)
        torch.cuda.synchronize()
    assert lib.vf_chain_plugin_launches() - n0 >= 2 * algo.gradient_steps
    assert algo.critic._fused_twin_q is not False
    bad = [str(x.message) for x in w if "vf_twin_q_update" in str(x.message) or "block-tile" in str(x.message) or "falling back" in str(x.message)]
    assert not bad, bad           # (r06: the horizon too -- the actor's persistent launches come from its BPTT plugin)
    env.close()


@pytest.mark.gpu
def test_bptt_with_the_reference_actor_on_a_non_default_shape_runs_on_chain_kernels():
    """BPTT(policy="MultiInputPolicy") -- the reference's own actor -- with a non-default `net_arch`: forward / reverse chains come from
    the generated class (r06: inside the class's own persistent launches -- no fallback warning), and one update's gradient equals the
    block-tile kernels' (plugins switched off: launch by launch) to fp32 summation order"""
    from visfly_amd import _lib
    from visfly_amd.bptt import BPTT
    from visfly_amd.envs import HoverEnv
    from _golden import ENV_DYN
    lib = _lib.lib()
    pk = dict(features_extractor_class="StateExtractor", features_extractor_kwargs={"net_arch": {"state": {"layer": [64, 64, 32]}}},
              net_arch=dict(pi=[32], qf=[96, 32]), activation_fn="relu", share_features_extractor=False)
    grads = []
    for on in (1, 0):
        lib.vf_chain_plugin_set_enabled(on)
        env = HoverEnv(num_agent_per_scene=2048, seed=5, dynamics_kwargs=dict(ENV_DYN), device=DEV, tensor_output=True, max_episode_steps=64)
        n0 = lib.vf_chain_plugin_launches()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            algo = BPTT(env, policy="MultiInputPolicy", policy_kwargs=pk, horizon=8, learning_rate=1e-3, seed=1)
            algo._grad_reverse_sweep()
        torch.cuda.synchronize()
        if on:
            assert lib.vf_chain_plugin_launches() - n0 == 2, "one persistent roll-out + one persistent reverse sweep from the BPTT plugin"
            assert not [str(x.message) for x in w if "falling back" in str(x.message)], [str(x.message) for x in w]
        else:
            assert lib.vf_chain_plugin_launches() == n0
        grads.append(algo.policy.grad.clone())
        env.close()
    lib.vf_chain_plugin_set_enabled(1)
    scale = grads[1].abs().max().item()
    assert scale > 0 and (grads[0] - grads[1]).abs().max().item() <= 2e-5 * scale


def _bptt_env(kind, N):
    import visfly_amd.envs as E
    from _golden import ENV_DYN
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
    if kind == "nav":
        return E.NavigationEnv(num_agent_per_scene=N, seed=5, dynamics_kwargs=dict(ENV_DYN), device=DEV, max_episode_steps=7, requires_grad=True,
                               tensor_output=True, random_kwargs=spawn)
    if kind == "racing2":          # 16 gate-relative columns, formed and differentiated inside the launches (kernel-side kind VF_ENV_RACING2)
        from _golden import RACING_DYN
        return E.RacingEnv2(num_agent_per_scene=N, seed=5, dynamics_kwargs=dict(RACING_DYN), device=DEV, max_episode_steps=7, requires_grad=True,
                            tensor_output=True)
    dkw = dict(ENV_DYN, action_type="thrust") if kind == "hover_thrust" else dict(ENV_DYN)
    return E.HoverEnv(num_agent_per_scene=N, seed=5, dynamics_kwargs=dkw, device=DEV, max_episode_steps=7, requires_grad=True, tensor_output=True)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1000, 4096])
@pytest.mark.parametrize("kind", ["hover", "nav", "racing2"])
def test_persistent_launches_of_a_generated_reference_actor_equal_the_loop(kind, N):
    """r06: BPTT(policy="MultiInputPolicy", net_arch=non-default) on the persistent launches of its GENERATED actor
    class (the BPTT plugin: k_bptt_rollout / k_bptt_reverse instances for ChainNetG<Spec>, 16 agents per wave) against the launch-by-launch
    sweep on the same class (plugin forward / reverse chains with the same rows-per-wave choice): actions, both heads, every saved
    activation and masked gradient of every slot, tape, done flags, adjoint slab, loss, flat gradient and the parameters after two updates
    are bit-identical (test_persistent_launches_with_the_reference_actor_equal_the_loop for the built-in classes)"""
    from visfly_amd import _jit, _lib
    from visfly_amd.bptt import BPTT
    lib = _lib.lib()
    H = 10
    name = "sac_nav_bptt" if kind == "nav" else "sac_hover"
    dims, ext, pi, vf, heads = _jit.PREBUILD_SAC[name]
    pk = dict(features_extractor_class="StateTargetExtractor" if kind == "nav" else "StateExtractor",
              features_extractor_kwargs={"net_arch": {k: {"layer": list(v)} for k, v in ext.items()}},
              net_arch=dict(pi=list(pi), qf=[64, 64]), activation_fn="relu", share_features_extractor=False)
    res = []
    for fused in (True, False):
        env = _bptt_env(kind, N)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            algo = BPTT(env, policy="MultiInputPolicy", policy_kwargs=pk, horizon=H, learning_rate=1e-3, seed=9)
            assert algo.reference_actor and tuple(algo.policy.head_dims) == (4, 4) and algo.policy.chain_jit
            assert algo.policy.chain_shape == _jit.shape_of(dims, ext, pi, vf, heads)
            algo.fused_rollout = algo.fused_reverse = fused
            used, rev_used = [], []
            orig, orig_rev = env.rollout_policy, env.reverse_policy
            env.rollout_policy = lambda *a, **k: used.append(orig(*a, **k)) or used[-1]
            env.reverse_policy = lambda *a, **k: rev_used.append(orig_rev(*a, **k)) or rev_used[-1]
            out = {}
            blk_names = None
            for it in range(2):
                n0 = lib.vf_chain_plugin_launches()
                loss = algo._grad_reverse_sweep()
                torch.cuda.synchronize()
                assert lib.vf_chain_plugin_launches() - n0 == (2 if fused else 2 * H), "every policy pass came from the plugins"
                out[f"loss{it}"] = loss.clone()
                out[f"grad{it}"] = algo.policy.grad.clone()
                out[f"action{it}"] = algo._last_rollout["action"].clone()
                out[f"done{it}"] = algo._last_rollout["done"].clone().bool()
                live = lambda x: x.transpose(-3, -4).reshape(*x.shape[:-4], x.shape[-3], -1, 4)[..., :N, :].clone()
                out[f"tape{it}"] = live(env._tape[:H])
                out[f"slab{it}"] = live(env._slab)
                out[f"adj{it}"] = live(env._adj)
                out[f"obs{it}"] = env.get_observation()["state"].clone()
                blk = algo.policy._slot_blocks[N][1]
                blk_names = [k for k, v in blk.items() if isinstance(v, torch.Tensor) and k.split(":")[0] in ("obs", "x", "feat", "pi", "vf", "mean", "value", "g")]
                for k in blk_names:
                    out[f"{k}{it}"] = blk[k][:H].clone()
                algo._apply(loss)
            out["flat"] = algo.policy.flat.clone()
        assert used == ([True, True] if fused else []) and rev_used == used, (used, rev_used, [str(x.message) for x in w])
        assert not [str(x.message) for x in w if "falling back" in str(x.message)], [str(x.message) for x in w]
        assert any(k.startswith("g:") for k in blk_names) and any(k.startswith("pi:") for k in blk_names)
        res.append(out)
        env.close()
    assert bool(res[0]["done0"].any()), "no episode ended inside the horizon"
    assert float(res[0]["grad0"].abs().max()) > 0 and float(res[0]["value0"].abs().max()) > 0
    for k in res[0]:
        a, b = res[0][k], res[1][k]
        same = torch.equal(a, b) if a.dtype == torch.bool else torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert same, f"{kind}: {k} differs (max abs {float((a.float() - b.float()).abs().max()):.3e})"


@pytest.mark.gpu
def test_persistent_launches_of_a_generated_mlp_policy_actor_equal_the_loop():
    """... and BPTT's own MlpPolicy actor (state-independent log_std) on a non-default shape: the horizon steps the generated POLICY-ONLY
    class with the action head (16 rows per wave), persistent launches from the BPTT plugin == the launch-by-launch loop, bit for bit"""
    from visfly_amd import _jit, _lib
    from visfly_amd.bptt import BPTT
    lib = _lib.lib()
    dims, ext, pi, vf = SHAPES["one_layer_extractor"]
    N, H = 2000, 8
    res = []
    for fused in (True, False):
        env = _bptt_env("hover_thrust", N)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            algo = BPTT(env, horizon=H, learning_rate=1e-3, seed=3,
                        policy_kwargs=dict(features_extractor_kwargs={"net_arch": {"state": {"layer": list(ext["state"])}}},
                                           net_arch=dict(pi=list(pi), vf=list(vf)), activation_fn="relu"))
            assert not algo.reference_actor and algo.policy.chain_jit and algo.policy.chain_shape == _jit.shape_of(dims, ext, pi, vf)
            algo.fused_rollout = algo.fused_reverse = fused
            n0 = lib.vf_chain_plugin_launches()
            algo.learn(3 * H * N)
            torch.cuda.synchronize()
            assert lib.vf_chain_plugin_launches() - n0 == 3 * (2 if fused else 2 * H)
        assert not [str(x.message) for x in w if "falling back" in str(x.message)], [str(x.message) for x in w]
        res.append((algo.policy.flat.clone(), algo.policy.grad.clone()))
        env.close()
    assert float(res[0][1].abs().max()) > 0
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.gpu
def test_cold_jit_compile_on_this_box(tmp_path, monkeypatch):
    """"compiled on first use" where users hit it: a shape nobody pre-built, an EMPTY cache directory, this box's own hipcc (four parts in
    parallel + link), the plugin registered and its forward checked against torch -- the seconds go to gpurun_out/ (profiles/r06_jit_cold.txt)"""
    import shutil
    import time
    from visfly_amd import _jit, _lib
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this box: a non-default net_arch then runs on the block-tile kernels (MlpPolicy warns once)")
    lib = _lib.lib()
    dims, ext, pi, vf = {"state": 13}, {"state": [32]}, [32], [64]            # the smallest class there is: one 32-wide layer per part
    sh = _jit.shape_of(dims, ext, pi, vf)
    assert sh is not None and not _jit.is_builtin(sh) and sh not in [_jit.shape_of(*v) for v in _jit.PREBUILD.values()]
    monkeypatch.setattr(_jit, "JIT_DIR", str(tmp_path / "jit"))
    monkeypatch.delitem(_jit._loaded, sh, raising=False)
    assert not os.path.exists(_jit.path_of(sh)) and _jit.path_of(sh).startswith(str(tmp_path))
    t0 = time.time()
    assert _jit.ensure(sh)
    secs = time.time() - t0
    assert os.path.exists(_jit.path_of(sh)) and _jit.name_of(sh).encode() in [lib.vf_chain_plugin_name(i) for i in range(lib.vf_chain_plugin_count())]
    from visfly_amd.ppo import MlpPolicy
    pol = MlpPolicy(dims, ext, pi, vf, DEV, seed=4)
    assert pol.chain_jit
    M = 1000
    obs = {"state": torch.randn((M, 13), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))}
    ref = pol.to_torch().double().to(DEV)
    m0, v0 = ref({k: v.double() for k, v in obs.items()})
    n0 = lib.vf_chain_plugin_launches()
    mean, value = pol.forward(obs)
    assert lib.vf_chain_plugin_launches() == n0 + 1, "the freshly compiled plugin served the forward"
    sc = max(m0.abs().max().item(), v0.abs().max().item(), 1e-3)
    assert (mean.double() - m0).abs().max().item() <= 2e-6 * sc and (value.view(-1).double() - v0.view(-1)).abs().max().item() <= 2e-6 * sc
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "r06_jit_cold.txt"), "a") as f:
            f.write(f"cold JIT of `{_jit.name_of(sh)}` on the GPU box: {secs:.1f} s (hipcc x 4 parts + link, {os.cpu_count()} cores)\n")
    assert secs < 600
