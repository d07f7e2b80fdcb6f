"""the reference's YAML policy_kwargs and its SB3-style checkpoint layout (visfly_amd/checkpoint.py).

The torch module below is written from the reference's module *structure* (utils/policies/policies.py:18-49,
utils/policies/extractors.py:464-486, SB3 ActorCriticPolicy attribute names); its state_dict() is what a
checkpoint of the reference holds for a StateTargetExtractor policy.
"""
import types

import pytest
import torch
import torch.nn as nn
import yaml

from visfly_amd import checkpoint
from visfly_amd.ppo import MlpPolicy

ALG_YAML = """
algorithm:
  policy: CustomMultiInputPolicy
  policy_kwargs:
    ortho_init: False
    features_extractor_class: StateTargetExtractor
    features_extractor_kwargs:
      net_arch:
        state:
          layer: [128, 64]
        target:
          layer: [96, 32]
    net_arch:
      pi: [64, 48]
      vf: [64, 64]
    activation_fn: ReLU
    optimizer_kwargs:
      weight_decay: 1.0e-5
  gamma: 0.99
  n_steps: 256
env:
  num_agent_per_scene: 48
  max_episode_steps: 256
  dynamics_kwargs: {dt: 0.03, ctrl_dt: 0.03, action_type: bodyrate}
eval_env:
  num_agent_per_scene: 1
"""


def _mlp(dims):
    mods = []
    for a, b in zip(dims[:-1], dims[1:]):
        mods += [nn.Linear(a, b), nn.ReLU()]
    return nn.Sequential(*mods)


class _Extractor(nn.Module):
    def __init__(self):
        super().__init__()
        self.state_extractor = _mlp([13, 128, 64])
        self.target_extractor = _mlp([3, 96, 32])

    def forward(self, obs):
        return torch.cat([self.state_extractor(obs["state"]), self.target_extractor(obs["target"])], dim=-1)


class _MlpExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        self.policy_net = _mlp([96, 64, 48])
        self.value_net = _mlp([96, 64, 64])


class _ReferenceShapedPolicy(nn.Module):
    def __init__(self):
        super().__init__()
        self.log_std = nn.Parameter(torch.full((4,), -0.5))
        self.features_extractor = _Extractor()
        self.pi_features_extractor = self.features_extractor     # SB3: shared extractor registered under three names
        self.vf_features_extractor = self.features_extractor
        self.mlp_extractor = _MlpExtractor()
        self.action_net = nn.Linear(48, 4)
        self.value_net = nn.Linear(64, 1)

    def forward(self, obs):
        f = self.features_extractor(obs)
        return self.action_net(self.mlp_extractor.policy_net(f)), self.value_net(self.mlp_extractor.value_net(f))


def _policy(tmp_cfg=None):
    cfg = yaml.safe_load(ALG_YAML)
    pk = checkpoint.policy_kwargs_from_reference(cfg["algorithm"]["policy_kwargs"], ["state", "target"])
    return pk, MlpPolicy({"state": 13, "target": 3}, pk["extractor"], pk["pi"], pk["vf"], "cpu", seed=3)


def test_reference_yaml_blocks(tmp_path):
    p = tmp_path / "cfg.yaml"
    p.write_text(ALG_YAML)
    cfg = checkpoint.load_yaml_config(str(p))
    assert cfg["eval_env"]["num_agent_per_scene"] == 1 and cfg["eval_env"]["max_episode_steps"] == 256   # deep merge
    assert cfg["env"]["num_agent_per_scene"] == 48
    pk = checkpoint.policy_kwargs_from_reference(cfg["algorithm"]["policy_kwargs"], ["state", "target"])
    assert pk == dict(extractor={"state": [128, 64], "target": [96, 32]}, pi=[64, 48], vf=[64, 64], weight_decay=1e-5,
                      ortho_init=False, activation=1, extractor_activation=1)      # the YAML sets ortho_init: false (default: the reference's True) and relu
    no_ortho_key = {k: v for k, v in cfg["algorithm"]["policy_kwargs"].items() if k != "ortho_init"}
    assert checkpoint.policy_kwargs_from_reference(no_ortho_key, ["state", "target"])["ortho_init"] is True
    # r06: activation_fn absent = the reference policy's own default, Tanh trunks over ReLU extractor MLPs (policies.py:108, extractors.py:666)
    no_act = checkpoint.policy_kwargs_from_reference({k: v for k, v in cfg["algorithm"]["policy_kwargs"].items() if k != "activation_fn"}, ["state", "target"])
    assert (no_act["activation"], no_act["extractor_activation"]) == (2, 1)
    assert checkpoint.policy_kwargs_from_reference(None, ["state"])["activation"] == 2
    with pytest.raises(NotImplementedError):
        checkpoint.policy_kwargs_from_reference(dict(cfg["algorithm"]["policy_kwargs"], squash_output=False), ["state", "target"])
    native = dict(extractor={"state": [32]}, pi=[16], vf=[16])
    assert checkpoint.policy_kwargs_from_reference(native, ["state"]) == native
    bad = dict(cfg["algorithm"]["policy_kwargs"], features_extractor_class="StateTargetImageExtractor")
    with pytest.raises(NotImplementedError):
        checkpoint.policy_kwargs_from_reference(bad, ["state", "target"])
    with_act = checkpoint.policy_kwargs_from_reference(dict(cfg["algorithm"]["policy_kwargs"], activation_fn="leaky_relu",
                                                            features_extractor_kwargs=dict(cfg["algorithm"]["policy_kwargs"]["features_extractor_kwargs"], activation_fn="elu")),
                                                       ["state", "target"])
    assert (with_act["activation"], with_act["extractor_activation"]) == (4, 3)
    with pytest.raises(NotImplementedError):
        checkpoint.policy_kwargs_from_reference(dict(cfg["algorithm"]["policy_kwargs"], activation_fn="gelu"), ["state", "target"])
    with pytest.raises(ValueError):
        checkpoint.policy_kwargs_from_reference(cfg["algorithm"]["policy_kwargs"], ["state"])


def test_orthogonal_init_matches_sb3_gains():
    """reference default ortho_init=True (policies.py:109): SB3 ActorCriticPolicy._build -- orthogonal weights with gain
    sqrt(2) (extractor, trunks), 0.01 (action_net), 1 (value_net), zero biases"""
    pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], "cpu", seed=1)
    for ly in pol.layers:
        w = pol.weight(ly)
        gain = {"mean": 0.01, "value": 1.0}.get(ly.dst, 2 ** 0.5)
        rows, cols = w.shape
        gram = (w @ w.T) if rows <= cols else (w.T @ w)
        assert torch.allclose(gram, gain ** 2 * torch.eye(min(rows, cols)), atol=1e-4), ly.dst
        assert float(pol.bias(ly).abs().max()) == 0.0
    kaiming = MlpPolicy({"state": 13}, {"state": [32]}, [16], [16], "cpu", seed=1, ortho_init=False)
    assert float(kaiming.bias(kaiming.layers[0]).abs().max()) > 0.0


def test_state_dict_keys_match_the_references_own_modules():
    """tests/golden/ckpt_keys.json (oracle/gen_golden.py::gen_ckpt_keys): names and shapes read off the reference's own
    StateTargetExtractor and create_mlp modules for the YAML above -- every one of them is in the state_dict this package
    writes, with the same shape (the remaining keys are SB3's action_net / value_net / log_std and the pi_/vf_ aliases)"""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_keys.json")))
    _, pol = _policy()
    got = {k: list(v.shape) for k, v in checkpoint.policy_state_dict(pol).items()}
    assert gold.pop("features_dim") == pol.widths["feat"]
    assert len(gold) == 16
    for k, shape in gold.items():
        assert got.get(k) == shape, (k, shape, got.get(k))
    rest = set(got) - set(gold)
    assert {k for k in rest if not k.startswith(("pi_features_extractor.", "vf_features_extractor."))} == {
        "action_net.weight", "action_net.bias", "value_net.weight", "value_net.bias", "log_std"}


def test_state_dict_uses_the_reference_parameter_names():
    _, pol = _policy()
    ref = _ReferenceShapedPolicy()
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    got = {k: tuple(v.shape) for k, v in checkpoint.policy_state_dict(pol).items()}
    assert got == want
    extra = checkpoint.load_policy_state_dict(pol, ref.state_dict())
    assert extra == []
    obs = {"state": torch.randn(37, 13), "target": torch.randn(37, 3)}
    m0, v0 = ref(obs)
    m1, v1 = pol.to_torch()(obs)
    assert torch.equal(m0, m1) and torch.equal(v0, v1)
    assert torch.equal(pol.log_std.cpu(), ref.log_std.detach())
    sd = dict(ref.state_dict())
    sd["features_extractor.depth_extractor.0.weight"] = torch.zeros(4, 4)
    with pytest.raises(KeyError):
        checkpoint.load_policy_state_dict(pol, sd)
    sd = {k: v for k, v in ref.state_dict().items() if not k.startswith(("features_extractor.", "vf_features_extractor."))}
    checkpoint.load_policy_state_dict(pol, sd)        # any one of the three aliases is enough
    sd = dict(ref.state_dict())
    sd["action_net.weight"] = torch.zeros(4, 64)
    with pytest.raises(ValueError):
        checkpoint.load_policy_state_dict(pol, sd)


def test_archive_round_trip_and_adam_layout(tmp_path):
    _, pol = _policy()
    ref = _ReferenceShapedPolicy()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    m, v = ref({"state": torch.randn(8, 13), "target": torch.randn(8, 3)})
    (m.sum() + v.sum() + ref.log_std.sum()).backward()
    opt.step()
    # torch numbers the parameters in registration order: the layout save() writes must be that order
    shapes = [tuple(p.shape) for p in ref.parameters()]
    assert [s for _, s in checkpoint._param_order(pol)] == shapes

    n = pol.n_params
    tr = types.SimpleNamespace(policy=pol, exp_avg=torch.randn(n), exp_avg_sq=torch.rand(n), _opt_step=7, lr=5e-5,
                               betas=(0.9, 0.999), adam_eps=1e-8, weight_decay=1e-5, num_timesteps=12345, gamma=0.99)
    path = checkpoint.save(tr, str(tmp_path / "PPO_std_1"))
    assert path.endswith(".zip")
    sd, osd, data = checkpoint.read_archive(path)
    assert set(sd) == set(ref.state_dict()) and data["num_timesteps"] == 12345
    torch.optim.Adam(ref.parameters()).load_state_dict(osd)       # a torch Adam over the reference-shaped module takes it

    _, pol2 = _policy()
    pol2.flat.zero_()
    tr2 = types.SimpleNamespace(policy=pol2, exp_avg=torch.zeros(n), exp_avg_sq=torch.zeros(n), _opt_step=0, num_timesteps=0)
    checkpoint.load_into(tr2, path)
    assert torch.equal(pol2.flat, pol.flat) and torch.equal(tr2.exp_avg, tr.exp_avg) and torch.equal(tr2.exp_avg_sq, tr.exp_avg_sq)
    assert tr2._opt_step == 7 and tr2.num_timesteps == 12345

    # an archive as the reference writes it: torch-saved policy.pth + optimiser, pickled (non-JSON) `data`
    import io
    import zipfile
    rp = str(tmp_path / "reference_like.zip")
    with zipfile.ZipFile(rp, "w") as z:
        for name, obj in (("policy.pth", ref.state_dict()), ("policy.optimizer.pth", opt.state_dict())):
            b = io.BytesIO()
            torch.save(obj, b)
            z.writestr(name, b.getvalue())
        z.writestr("data", b"\x80\x04 not json")
    checkpoint.load_into(tr2, rp)
    obs = {"state": torch.randn(5, 13), "target": torch.randn(5, 3)}
    assert torch.equal(pol2.to_torch()(obs)[0], ref(obs)[0])
    assert tr2._opt_step == 1
    k = checkpoint._param_order(pol2)[1]
    assert torch.equal(tr2.exp_avg[k[0]:k[0] + 13 * 128].view(128, 13), opt.state_dict()["state"][1]["exp_avg"])
