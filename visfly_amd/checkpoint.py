"""Data formats either side of the trainers: the reference's YAML blocks and its SB3-style checkpoints.

* ``load_yaml_config`` / ``policy_kwargs_from_reference``: the ``algorithm:`` / ``env:`` blocks of the reference's
  experiment files (exps/examples/alg_cfgs/*/PPO.yaml, env_cfgs/*.yaml; utils/common.py:232-237) are accepted as they
  are; the SB3-style ``policy_kwargs`` (features_extractor_class / features_extractor_kwargs.net_arch.<key>.layer /
  net_arch.pi|vf / activation_fn / optimizer_kwargs) are translated to ``MlpPolicy``'s arguments.  Anything the MFMA
  policy does not implement (image extractors, batch / layer norm, non-ReLU activations, recurrent extractors)
  raises instead of being dropped.
* ``policy_state_dict`` / ``load_policy_state_dict``: the flat parameter buffer <-> the parameter names the
  reference's ``CustomMultiInputActorCriticPolicy`` has in its ``state_dict()`` (utils/policies/policies.py:55-190,
  utils/policies/extractors.py:464-486; SB3 2.2.1 ``ActorCriticPolicy``: features_extractor.<key>_extractor.<i>,
  mlp_extractor.policy_net|value_net.<i>, action_net, value_net, log_std; with a shared extractor SB3 registers the
  same module as pi_features_extractor / vf_features_extractor too, so those aliases are written and accepted).
* ``save`` / ``load``: a zip archive laid out like SB3's ``BaseAlgorithm.save`` (utils/algorithms/PPO.py:418-572 defers
  to it): ``policy.pth`` (torch state_dict, reference names), ``policy.optimizer.pth`` (torch.optim.Adam state_dict
  layout), ``data`` (JSON of the hyper-parameters; plain values, no pickled objects), ``_stable_baselines3_version``.
  ``load`` reads archives written by the reference as well: only ``policy.pth`` and, when present, the optimiser
  moments are taken from them; the pickled ``data`` entries of SB3 are ignored.
"""
import copy
import io
import json
import zipfile
from typing import Dict, List, Optional

import torch as th

_EXTRACTORS = {"StateExtractor": ("state",), "StateTargetExtractor": ("state", "target")}
_HEAD_NAMES = {"mean": "action_net", "value": "value_net"}
ARCHIVE_VERSION = "2.2.1"       # environment.yml:275 pins stable-baselines3==2.2.1


def deep_merge(origin: dict, target: dict) -> dict:
    """utils/common.py:210-229: recursive dict merge, `target` wins"""
    out = copy.deepcopy(origin)
    for k, v in target.items():
        out[k] = deep_merge(out[k], v) if isinstance(out.get(k), dict) and isinstance(v, dict) else copy.deepcopy(v)
    return out


def load_yaml_config(path: str) -> dict:
    """utils/common.py:232-237: eval_env inherits from env"""
    import yaml
    with open(path, "r") as f:
        cfg = yaml.safe_load(f)
    if "env" in cfg:
        cfg["eval_env"] = deep_merge(cfg["env"], cfg.get("eval_env") or {})
    return cfg


def policy_kwargs_from_reference(pk: Optional[dict], obs_keys: List[str]) -> dict:
    """SB3-style policy_kwargs of the reference's YAMLs -> MlpPolicy arguments (`extractor`, `pi`, `vf`,
    `log_std_init`) plus `weight_decay` from optimizer_kwargs.  Already-native dicts pass through."""
    pk = dict(pk or {})
    if "extractor" in pk:       # already MlpPolicy arguments
        return pk
    out: Dict[str, object] = {}
    cls = pk.get("features_extractor_class", "StateTargetExtractor" if "target" in obs_keys else "StateExtractor")
    cls = cls if isinstance(cls, str) else cls.__name__
    if cls not in _EXTRACTORS:
        raise NotImplementedError(f"features_extractor_class {cls}: only the vector extractors "
                                  f"{sorted(_EXTRACTORS)} are on the visual=False path")
    missing = [k for k in _EXTRACTORS[cls] if k not in obs_keys]
    if missing:
        raise ValueError(f"{cls} needs observation keys {missing}")
    arch = (pk.get("features_extractor_kwargs") or {}).get("net_arch") or {}
    ext = {}
    for k in _EXTRACTORS[cls]:
        a = arch.get(k) or {}
        if a.get("bn") or a.get("ln"):
            raise NotImplementedError("batch / layer norm in the extractor MLPs is not implemented")
        ext[k] = list(a.get("layer", []))
    if arch:        # (no features_extractor_kwargs.net_arch: the trainers' default, the YAMLs' [128, 64] per key)
        out["extractor"] = ext
    # trunks: the policy's activation_fn -- the reference's default is Tanh (policies.py:108; its YAMLs set relu); extractor MLPs:
    # features_extractor_kwargs.activation_fn -- default ReLU (extractors.py:560,583,666).  relu | tanh | elu | leaky_relu
    # (policies.py:64-69) as strings or torch classes
    from .ppo import activation_kind
    out["activation"] = activation_kind(pk.get("activation_fn", "tanh"))
    out["extractor_activation"] = activation_kind((pk.get("features_extractor_kwargs") or {}).get("activation_fn", "relu"))
    if pk.get("squash_output", True) is False:
        raise NotImplementedError("squash_output=False: the action head is the tanh-squashed Gaussian (policies.py:114,177-181)")
    if pk.get("use_sde"):
        raise NotImplementedError("use_sde: state-dependent exploration is not implemented")
    out["ortho_init"] = bool(pk.get("ortho_init", True))           # policies.py:109, SB3 ActorCriticPolicy._build
    na = pk.get("net_arch") or {}
    if isinstance(na, list):        # SB3 shorthand: shared sizes for both trunks
        na = dict(pi=na, vf=na)
    for flag in ("pi_bn", "pi_ln", "vf_bn", "vf_ln", "squash_output"):
        if na.get(flag):
            raise NotImplementedError(f"net_arch.{flag} is not implemented")
    out["pi"], out["vf"] = list(na.get("pi", [64, 64])), list(na.get("vf", [64, 64]))
    if "log_std_init" in pk:
        out["log_std_init"] = float(pk["log_std_init"])
    if pk.get("share_features_extractor", True) is False:
        raise NotImplementedError("separate pi / vf feature extractors are not implemented")
    wd = (pk.get("optimizer_kwargs") or {}).get("weight_decay")
    if wd is not None:
        out["weight_decay"] = float(wd)
    return out


def _names(policy):
    """[(layer, reference module prefix)] in schedule order"""
    out, idx = [], {}
    for ly in policy.layers:
        if ly.dst in _HEAD_NAMES:
            out.append((ly, [_HEAD_NAMES[ly.dst]]))
            continue
        if ly.first or ly.src.startswith("x:") or ly.dst == "feat":
            key = ly.src.split(":")[1] if ly.src.startswith(("obs:", "x:")) else None
            i = idx.get(("ext", key), 0)
            idx[("ext", key)] = i + 1
            out.append((ly, [f"{p}.{key}_extractor.{2 * i}" for p in ("features_extractor", "pi_features_extractor",
                                                                        "vf_features_extractor")]))
            continue
        trunk = ly.dst.split(":")[0]
        i = idx.get(trunk, 0)
        idx[trunk] = i + 1
        out.append((ly, [f"mlp_extractor.{'policy_net' if trunk == 'pi' else 'value_net'}.{2 * i}"]))
    return out


def policy_state_dict(policy) -> Dict[str, th.Tensor]:
    sd = {"log_std": policy.log_std.detach().cpu().clone()}
    for ly, prefixes in _names(policy):
        w, b = policy.weight(ly).detach().cpu().clone(), policy.bias(ly).detach().cpu().clone()
        for p in prefixes:
            sd[p + ".weight"], sd[p + ".bias"] = w, b
    return sd


def load_policy_state_dict(policy, sd: Dict[str, th.Tensor], strict: bool = True):
    used = set()
    with th.no_grad():
        for ly, prefixes in _names(policy):
            p = next((p for p in prefixes if p + ".weight" in sd), None)
            if p is None:
                raise KeyError(f"state_dict has none of {[q + '.weight' for q in prefixes]}")
            w, b = sd[p + ".weight"], sd[p + ".bias"]
            if tuple(w.shape) != (ly.No, ly.K) or tuple(b.shape) != (ly.No,):
                raise ValueError(f"{p}: shape {tuple(w.shape)} does not match the policy's layer ({ly.No}, {ly.K})")
            policy.weight(ly).copy_(w.to(policy.device, th.float32))
            policy.bias(ly).copy_(b.to(policy.device, th.float32))
            used.update(q + s for q in prefixes for s in (".weight", ".bias"))
        if "log_std" in sd:
            policy.log_std.copy_(sd["log_std"].to(policy.device, th.float32).reshape(-1))
            used.add("log_std")
        elif strict:
            raise KeyError("state_dict has no log_std")
    extra = sorted(set(sd) - used)
    if strict and extra:
        raise KeyError(f"unexpected keys in state_dict (layers the MFMA policy does not have): {extra[:8]}")
    policy.mark_updated()
    return extra


def _param_order(policy):
    """torch.optim.Adam numbers parameters in module registration order (SB3: features_extractor, mlp_extractor,
    action_net, value_net; log_std is registered first in ActorCriticPolicy._build): -> [(offset, shape)]"""
    order = [(policy.log_std_off, (policy.n_params - policy.log_std_off,))]
    named = _names(policy)
    rank = lambda pre: (0 if pre.startswith("features_extractor") else 1 if pre.startswith("mlp_extractor.policy") else
                        2 if pre.startswith("mlp_extractor.value") else 3 if pre == "action_net" else 4)
    for ly, prefixes in sorted(named, key=lambda x: rank(x[1][0])):
        order += [(ly.w_off, (ly.No, ly.K)), (ly.b_off, (ly.No,))]
    return order


def _hyper(trainer) -> dict:
    keys = ("n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "clip_range", "ent_coef", "vf_coef", "max_grad_norm",
            "lr", "weight_decay", "adam_eps", "betas", "normalize_advantage", "target_kl", "seed", "H", "num_timesteps", "_opt_step",
            "clip_range_vf")
    out = {k: getattr(trainer, k) for k in keys if hasattr(trainer, k)}
    # schedules (callables of progress_remaining) are archived as their current value: an archive holds data, not code
    out = {k: (trainer._now(v) if callable(v) else v) for k, v in out.items()}
    out["learning_rate"] = out.pop("lr", None)
    out["algorithm"], out["policy_spec"] = type(trainer).__name__, trainer.policy.spec
    out["obs_dims"] = trainer.policy.obs_dims
    return out


def save(trainer, path: str):
    """trainer: PPO / BPTT / SHAC of this package (policy + Adam moments + hyper-parameters)"""
    pol = trainer.policy
    path = path if str(path).endswith(".zip") else f"{path}.zip"
    opt_state = {}
    for i, (off, shape) in enumerate(_param_order(pol)):
        n = 1
        for s in shape:
            n *= s
        opt_state[i] = dict(step=th.tensor(float(trainer._opt_step)),
                            exp_avg=trainer.exp_avg[off:off + n].detach().cpu().reshape(shape).clone(),
                            exp_avg_sq=trainer.exp_avg_sq[off:off + n].detach().cpu().reshape(shape).clone())
    group = dict(lr=trainer.lr, betas=tuple(trainer.betas), eps=trainer.adam_eps, weight_decay=trainer.weight_decay,
                 amsgrad=False, params=list(range(len(opt_state))))
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("data", json.dumps(_hyper(trainer), indent=1, default=str))
        for name, obj in (("policy.pth", policy_state_dict(pol)),
                          ("policy.optimizer.pth", dict(state=opt_state, param_groups=[group]))):
            buf = io.BytesIO()
            th.save(obj, buf)
            z.writestr(name, buf.getvalue())
        z.writestr("_stable_baselines3_version", ARCHIVE_VERSION)
    return path


def read_archive(path: str):
    """-> (policy state_dict, optimiser state_dict or None, data dict or None)"""
    path = path if str(path).endswith(".zip") else f"{path}.zip"
    with zipfile.ZipFile(path) as z:
        names = set(z.namelist())
        if "policy.pth" not in names:
            raise FileNotFoundError(f"{path}: no policy.pth in the archive")
        sd = th.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
        opt = None
        if "policy.optimizer.pth" in names:
            opt = th.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu", weights_only=True)
        data = None
        if "data" in names:
            try:
                data = json.loads(z.read("data").decode())
            except (ValueError, UnicodeDecodeError):
                data = None
    return sd, opt, data


_CTOR_KEYS = ("n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "clip_range", "clip_range_vf", "ent_coef", "vf_coef",
              "max_grad_norm", "learning_rate", "weight_decay", "adam_eps", "betas", "normalize_advantage", "target_kl", "seed",
              "horizon")


def ctor_kwargs_from_archive(path: str, overrides: Optional[dict] = None) -> dict:
    """constructor kwargs for ``cls.load`` (SB3 BaseAlgorithm.load restores them from the archive's `data`, PPO.py:432-572):
    the saved hyper-parameters and network shape, overridden by the caller's kwargs.  Archives written by the reference carry
    a pickled `data` this package does not unpickle -- then only the caller's kwargs apply."""
    _sd, _opt, data = read_archive(path)
    kw = {}
    if isinstance(data, dict):
        for k in _CTOR_KEYS:
            v = data.get("H") if k == "horizon" else data.get(k)
            if v is not None and not isinstance(v, dict):
                kw[k] = tuple(v) if k == "betas" else v
        spec = data.get("policy_spec")
        if isinstance(spec, dict) and "extractor" in spec:
            kw["policy_kwargs"] = dict(extractor=spec["extractor"], pi=spec["pi"], vf=spec["vf"], activation=spec.get("activation", "relu"),
                                       extractor_activation=spec.get("extractor_activation", "relu"))
    kw.update(overrides or {})
    return kw


def load_into(trainer, path: str, load_optimizer: bool = True):
    """in-place load (SB3 ``set_parameters``): policy parameters and, if the archive has them and the layout matches,
    the Adam moments and step count"""
    sd, opt, data = read_archive(path)
    pol = trainer.policy
    load_policy_state_dict(pol, sd, strict=True)
    if load_optimizer and opt and opt.get("state"):
        order = _param_order(pol)
        st = opt["state"]
        if len(st) == len(order) and all(tuple(st[i]["exp_avg"].shape) == order[i][1] for i in range(len(order))):
            for i, (off, shape) in enumerate(order):
                n = st[i]["exp_avg"].numel()
                trainer.exp_avg[off:off + n] = st[i]["exp_avg"].reshape(-1).to(pol.device, th.float32)
                trainer.exp_avg_sq[off:off + n] = st[i]["exp_avg_sq"].reshape(-1).to(pol.device, th.float32)
            trainer._opt_step = int(float(st[0]["step"]))
        else:
            import warnings
            warnings.warn(f"{path}: the optimiser state does not match this policy's parameter layout "
                          f"({len(st)} vs {len(order)} tensors) -- Adam moments NOT restored", stacklevel=2)
    if isinstance(data, dict) and "num_timesteps" in data and not isinstance(data["num_timesteps"], dict):
        trainer.num_timesteps = int(data["num_timesteps"])
    return trainer
