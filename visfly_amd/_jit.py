"""Register-chained MLP kernels for network shapes libvisfly_amd.so holds no instance of: compiled on first use.

The chain kernels keep a wave's activations in MFMA accumulator registers, so the layer shapes are compile-time
(csrc/vf_mlp_chain.hpp).  The library carries the YAML-default shapes of the reference (one or two [128, 64] extractor branches,
[64, 64] trunks); for any other `features_extractor_kwargs.net_arch` / `net_arch=dict(pi=.., vf=..)` (utils/policies/extractors.py:376-449)
this module writes a translation unit that names the shape (csrc/vf_mlp_chain_gen.hpp: ChainNetG<Spec>), compiles it with hipcc for gfx950
into ``csrc/jit/libvf_chain_<hash>.so`` (four parts in parallel, ~40 s once; cached by the hash of the shape, the chain headers and the
flags) and registers it with the library (vf_chain_plugin_load).  ``__graft_entry__.build()`` pre-builds the shapes the tests use, so
the GPU box finds them in the snapshot.

Shapes that qualify: 1-2 observation branches of <= 32 columns, 1-4 layers per branch / trunk, widths in multiples of 32 up to 128,
the actor-critic heads (4, 1), the SAC-style Actor's (4, 4), or the twin critic's (1, 1) with its pass-through action input behind ONE
observation branch.  Everything else keeps running on the block-tile kernels (MlpPolicy warns once).
``VISFLY_AMD_JIT=0`` switches the compilation off.

Two more plugin kinds hold ONE instance each of a persistent launch for a generated class under one env kind / action type / integrator /
motor-lag setting, built when a trainer first needs them: the roll-out plugin (PPO's collect_rollouts: ``ensure_rollout``,
csrc/vf_ppo_rollout_kernel.hpp) and, r06, the BPTT plugin (both halves of a BPTT / SHAC horizon for an actor class: ``ensure_bptt``,
csrc/vf_bptt_rollout_kernel.hpp + csrc/vf_bptt_reverse_kernel.hpp).
"""
import hashlib
import os
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

from ._build import CSRC, HIPCC_FLAGS, INCLUDE

JIT_DIR = os.path.join(CSRC, "jit")
MAX_DEPTH = 4
# the shapes libvisfly_amd.so itself instantiates (NetHover / NetNav and their policy-only classes)
_BUILTIN = {((16,), ((4, 2),), (2, 2), (2, 2)), ((16, 8), ((4, 2), (4, 2)), (2, 2), (2, 2)),
            ((16,), ((4, 2),), (2, 2), (2, 2), (4, 4)), ((16, 8), ((4, 2), (4, 2)), (2, 2), (2, 2), (4, 4)),      # (.., (4, 4)): the SAC-style Actor
            ((16,), ((4, 2),), (2, 2), (2, 2), (1, 1))}                                                            # (.., (1, 1)): its twin critic (NetCriticHover)
_HEADERS = ("vf_mlp_chain.hpp", "vf_mlp_chain_bwd.hpp", "vf_mlp_chain_kernels.hpp", "vf_mlp_chain_gen.hpp", "vf_chain_plugin.hpp",
            "vf_common.hpp", "vf_ppo_device.hpp")
_ROLLOUT_HEADERS = ("vf_ppo_rollout_kernel.hpp", "vf_env_epilogue.hpp", "vf_env_device.hpp", "vf_dyn_device.hpp", "vf_xmath.hpp",
                    "vf_quad.hpp", "vf_handles.hpp")
_BPTT_HEADERS = _ROLLOUT_HEADERS[1:] + ("vf_bptt_rollout_kernel.hpp", "vf_bptt_reverse_kernel.hpp", "vf_env_bwd_body.hpp", "vf_env_bwd_quad.hpp",
                                        "vf_dyn_quad.hpp")
_loaded = {}
# shapes compiled ahead of time by __graft_entry__.build() (name -> observation widths, extractor layers, pi, vf): the ones the parity
# tests run (tests/test_chain_jit_gpu.py), so that a GPU box without a warm cache finds them in the snapshot
PREBUILD = {
    "verdict": ({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [128, 128], [32]),
    "one_layer_extractor": ({"state": 13}, {"state": [64]}, [64], [64, 64, 32]),
    "ragged": ({"state": 13, "target": 3}, {"state": [96, 64, 32], "target": [32]}, [32, 32, 32], [128]),
    # 30 forward tiles: the fused PPO step keeps the ReLU masks as bits (csrc/vf_mlp_chain_gen.hpp: kGenLiveTiles)
    "wide": ({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [128, 128], [128, 128]),
}
# the SAC-style Actor (heads (4, 4); td_policies.Actor: `pi` = latent_pi, `vf` = log_latent_pi) on a non-default shape: (.., head_dims)
# the reference policy's DEFAULT activations (Tanh trunks, policies.py:108; ReLU extractor MLPs, extractors.py:666) on the YAMLs' shapes:
# (.., head_dims, (trunk, extractor) activations)
PREBUILD_ACT = {
    "tanh_nav": ({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], (4, 1), (2, 1)),
    "tanh_hover": ({"state": 13}, {"state": [128, 64]}, [64, 64], [64, 64], (4, 1), (2, 1)),
    "elu_leaky_nav": ({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], (4, 1), (3, 4)),
}
# the twin critic (heads (1, 1) + pass-through action; td_policies.ContinuousCritic) on non-default shapes: (.., head_dims, passthrough)
PREBUILD_CRITIC = {
    "critic_hover": ({"state": 13, "action": 4}, {"state": [64, 64, 32]}, [32], [32], (1, 1), ("action",)),      # SHAC(net_arch pi=[32]) over features [64, 64, 32]
    "critic_wide": ({"state": 13, "action": 4}, {"state": [128, 96]}, [128, 64], [128, 64], (1, 1), ("action",)),
}
PREBUILD_SAC = {
    "sac_nav": ({"state": 13, "target": 3}, {"state": [128, 64], "target": [64]}, [128, 64], [64], (4, 4)),
    "sac_hover": ({"state": 13}, {"state": [64, 64, 32]}, [32], [32], (4, 4)),      # (the Actor BPTT builds for pi=[32]: log_latent_pi mirrors latent_pi)
    "sac_nav_bptt": ({"state": 13, "target": 3}, {"state": [64, 64], "target": [32]}, [64], [64], (4, 4)),      # ... over StateTargetExtractor, pi=[64]
}
# ... and their roll-out plugins for the configurations tests/test_ppo_gpu.py runs: (shape name, (VF_ENV_*, VF_ACT_*, VF_INT_*, ctrl_delay))
# ... and the BPTT plugins (both persistent launches of a horizon) tests/test_chain_jit_gpu.py runs: (shape name in PREBUILD_SAC / PREBUILD,
# (kernel-side env kind, VF_ACT_*, VF_INT_*, ctrl_delay))
PREBUILD_BPTT = [("sac_hover", (0, 1, 0, True)), ("sac_nav_bptt", (1, 1, 0, True)), ("one_layer_extractor", (0, 0, 0, True)),
                 ("sac_hover", (3, 0, 0, True))]        # RacingEnv2: kernel-side kind VF_ENV_RACING2, thrust
PREBUILD_ROLLOUT = [("verdict", (1, 1, 0, True)), ("one_layer_extractor", (0, 1, 1, False)), ("one_layer_extractor", (1, 1, 0, True)),
                    ("one_layer_extractor", (2, 1, 0, True)), ("one_layer_extractor", (3, 1, 0, True))]      # RacingEnv / RacingEnv2 (kernel-side kind 3)


def shape_of(obs_dims, extractor, pi, vf, head_dims=(4, 1), passthrough=(), acts=(1, 1)):
    """(KIN, extractor widths in tiles, pi tiles, vf tiles[, (4, 4)][, ("act", trunks, extractor)]) of a network the generated chain classes
    cover, else None.  Heads (4, 1): the actor-critic of the PPO policies; (4, 4): the SAC-style Actor of BPTT / SHAC (mu / log_std heads).
    acts: VF_ACTIVATION_* of the trunks / the extractor MLPs -- (1, 1) = ReLU networks (no element: the keys of r05's shapes)"""
    critic = tuple(head_dims) == (1, 1)      # td_policies.ContinuousCritic: th.cat([features, actions]) -> qf0 / qf1 (one pass-through input of <= 4 columns)
    if critic != bool(passthrough) or tuple(head_dims) not in ((4, 1), (4, 4), (1, 1)) or not 1 <= len(extractor) <= 2:
        return None
    if critic and (len(passthrough) != 1 or len(extractor) != 1 or not 1 <= int(obs_dims[list(passthrough)[0]]) <= 4):
        return None
    kin, ew = [], []
    for k, hidden in extractor.items():
        d = int(obs_dims[k])
        if not 1 <= d <= 32 or not 1 <= len(hidden) <= MAX_DEPTH:
            return None
        kin.append((d + 7) & ~7)
        ew.append(tuple(int(h) for h in hidden))
    trunks = (tuple(int(h) for h in pi), tuple(int(h) for h in vf))
    if not all(1 <= len(t) <= MAX_DEPTH for t in trunks):
        return None
    widths = [h for e in ew for h in e] + [h for t in trunks for h in t]
    if any(h % 32 or not 32 <= h <= 128 for h in widths):
        return None
    tiles = lambda t: tuple(h // 32 for h in t)
    sh = tuple(kin), tuple(tiles(e) for e in ew), tiles(trunks[0]), tiles(trunks[1])
    sh = sh if tuple(head_dims) == (4, 1) else sh + (tuple(head_dims),)
    return sh if tuple(acts) == (1, 1) else sh + (("act", int(acts[0]), int(acts[1])),)


def is_builtin(shape):
    return shape in _BUILTIN          # (ReLU networks only: a shape with an ("act", ..) element is never one of these)


def _heads(shape):
    return (4, 4) if (4, 4) in shape[4:] else (1, 1) if (1, 1) in shape[4:] else (4, 1)


def _acts(shape):
    """(trunk activation, extractor activation) as VF_ACTIVATION_*"""
    for e in shape[4:]:
        if e and e[0] == "act":
            return e[1], e[2]
    return 1, 1


_ACT_NAME = {1: "relu", 2: "tanh", 3: "elu", 4: "leaky_relu"}


def name_of(shape):
    kin, ew, pw, vw = shape[:4]
    f = lambda t: "[" + ",".join(str(32 * x) for x in t) + "]"
    a = _acts(shape)
    return (" ".join(f"in{k}{f(e)}" for k, e in zip(kin, ew)) + f" pi{f(pw)} vf{f(vw)}" + {(4, 4): " heads 4/4", (1, 1): " (+) action, heads 1/1", (4, 1): ""}[_heads(shape)] +
            ("" if a == (1, 1) else f" act {_ACT_NAME[a[0]]}/{_ACT_NAME[a[1]]}"))


def source(shape):
    """the generated translation unit: a Spec (csrc/vf_mlp_chain_gen.hpp) and the plugin's entry points"""
    kin, ew, pw, vw = shape[:4]
    pad = lambda t, n=MAX_DEPTH: ", ".join(str(x) for x in (tuple(t) + (0,) * n)[:n])
    nb = len(kin)
    return f"""// generated by visfly_amd/_jit.py -- chain plugin for: {name_of(shape)}
#define VF_CHAIN_PLUGIN 1
#include "vf_mlp_chain_gen.hpp"
#include "vf_chain_plugin.hpp"
namespace {{
struct Spec {{
    static constexpr int NB = {nb};
    static constexpr int KIN[2] = {{{pad(kin, 2)}}};
    static constexpr int DE[2] = {{{pad([len(e) for e in ew], 2)}}};
    static constexpr int EW[2][{MAX_DEPTH}] = {{{{{pad(ew[0])}}}, {{{pad(ew[1] if nb > 1 else ())}}}}};
    static constexpr int DP = {len(pw)}, DV = {len(vw)};
    static constexpr int PW[{MAX_DEPTH}] = {{{pad(pw)}}};
    static constexpr int VW[{MAX_DEPTH}] = {{{pad(vw)}}};
    static constexpr bool VF = true;
    static constexpr int HM = {_heads(shape)[0]}, HV = {_heads(shape)[1]};
    static constexpr int PASS = {1 if _heads(shape) == (1, 1) else 0};
    static constexpr int ACT = {_acts(shape)[0]}, EACT = {_acts(shape)[1]};       // VF_ACTIVATION_* of the trunks / the extractor MLPs
}};
struct SpecPi : Spec {{
    static constexpr bool VF = false;
}};
}}  // namespace
using Net = vf::ChainNetG<Spec>;
using NetPi = vf::ChainNetG<SpecPi>;
VF_CHAIN_PLUGIN_DEFINE(Net, NetPi, "{name_of(shape)}")
"""


def rollout_source(shape, cfg):
    """the roll-out plugin of `shape` under cfg = (env kind, action type, integrator, ctrl_delay) -- VF_ENV_* / VF_ACT_* / VF_INT_* values:
    ONE instance of the persistent roll-out launch (csrc/vf_ppo_rollout_kernel.hpp)"""
    kind, act, integ, delay = cfg
    return source(shape) + f"""#include "vf_ppo_rollout_kernel.hpp"
VF_CHAIN_PLUGIN_ROLLOUT_DEFINE(Net, {int(kind)}, {int(act)}, {int(integ)}, {"true" if delay else "false"}, "{name_of(shape)} roll-out kind {int(kind)} act {int(act)} int {int(integ)} delay {int(bool(delay))}")
"""


def bptt_source(shape, cfg):
    """the BPTT plugin of `shape` (an actor class: the SAC-style Actor, or an actor-critic whose policy-only class the horizon steps) under
    cfg = (KERNEL-side env kind, action type, integrator, ctrl_delay): ONE instance each of the two persistent launches of a horizon
    (csrc/vf_bptt_rollout_kernel.hpp, csrc/vf_bptt_reverse_kernel.hpp), compiled as parts 5 / 6 (+ 7: the table)"""
    kind, act, integ, delay = cfg
    return source(shape) + f"""#if VF_CHAIN_PLUGIN_PART != 6
#include "vf_bptt_rollout_kernel.hpp"
#endif
#if VF_CHAIN_PLUGIN_PART != 5
#include "vf_bptt_reverse_kernel.hpp"
#endif
VF_CHAIN_PLUGIN_BPTT_DEFINE(Net, NetPi, {int(kind)}, {int(act)}, {int(integ)}, {"true" if delay else "false"}, "{name_of(shape)} BPTT horizon kind {int(kind)} act {int(act)} int {int(integ)} delay {int(bool(delay))}")
"""


def _flags():
    extra = os.environ.get("VISFLY_AMD_JIT_FLAGS", "").split()       # e.g. -DVF_GEN_LIVE_TILES=0 (A/B builds; part of the cache key)
    return [f for f in HIPCC_FLAGS if f not in ("-shared", "-Wall")] + ["-ftemplate-depth=4096", "-Wno-unused-const-variable"] + extra + [
        "-I", INCLUDE, "-I", CSRC]


def _is_bptt(rollout):
    return rollout is not None and rollout[0] == "bptt"


def _source_of(shape, rollout):
    """rollout: None (the chain plugin), (kind, act, integ, delay) (PPO's roll-out plugin) or ("bptt", kind, act, integ, delay)"""
    return source(shape) if rollout is None else bptt_source(shape, rollout[1:]) if _is_bptt(rollout) else rollout_source(shape, rollout)


def _key(shape, rollout=None):
    h = hashlib.sha256()
    h.update(_source_of(shape, rollout).encode())
    h.update(" ".join(_flags()[:-4]).encode())
    for n in _HEADERS + (() if rollout is None else _BPTT_HEADERS if _is_bptt(rollout) else _ROLLOUT_HEADERS) + (os.path.join(INCLUDE, "visfly_amd.h"),):
        with open(n if os.path.isabs(n) else os.path.join(CSRC, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _slug(shape, rollout=None):
    kin, ew, pw, vw = shape[:4]
    t = lambda x: "".join(str(v) for v in x)
    s = "e" + "_".join(f"{k}x{t(e)}" for k, e in zip(kin, ew)) + f"_p{t(pw)}_v{t(vw)}" + {(4, 4): "_h44", (1, 1): "_h11", (4, 1): ""}[_heads(shape)]
    if _acts(shape) != (1, 1):
        s += "_a%d%d" % _acts(shape)
    if _is_bptt(rollout):
        return s + "_bptt" + "".join(str(int(x)) for x in rollout[1:])
    return s if rollout is None else s + "_roll" + "".join(str(int(x)) for x in rollout)


def path_of(shape, rollout=None):
    """<cache>/libvf_chain_<shape>[_roll<cfg>]_<hash of source + headers + flags>.so"""
    return os.path.join(JIT_DIR, f"libvf_chain_{_slug(shape, rollout)}_{_key(shape, rollout)}.so")


def build(shape, verbose=False, rollout=None):
    """compile the plugin of `shape` (rollout = (kind, act, integ, delay): its roll-out plugin for that configuration) unless the cache
    holds it -> path of the shared object"""
    out = path_of(shape, rollout)
    if os.path.exists(out):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the chain kernels of a non-default network shape are compiled on first use")
    os.makedirs(JIT_DIR, exist_ok=True)
    # one compiler run per shape and machine: the ranks of a job construct the same policy at the same time -- the first takes the
    # lock and compiles, the others wait on it and find the file
    import fcntl
    lock = open(os.path.join(JIT_DIR, f".{_slug(shape, rollout)}.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    tmpdir = None
    try:
        if os.path.exists(out):
            return out
        tmpdir = tempfile.mkdtemp(prefix="visfly_amd_jit_")
        src = os.path.join(tmpdir, "plugin.hip")
        with open(src, "w") as f:
            f.write(_source_of(shape, rollout))

        def part(p):
            obj = os.path.join(tmpdir, f"part{p}.o")
            cmd = [hipcc] + _flags() + [f"-DVF_CHAIN_PLUGIN_PART={p}", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on the chain plugin for {name_of(shape)}:\n{r.stderr[-4000:]}")
            return obj

        with ThreadPoolExecutor(max_workers=4) as pool:
            objs = list(pool.map(part, range(4) if rollout is None else [5, 6, 7] if _is_bptt(rollout) else [4]))
        tmp = f"{out}.{os.getpid()}.tmp"       # private name, then rename: concurrent builders (ranks) never see a torn file
        try:
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"linking the chain plugin for {name_of(shape)} failed:\n{r.stderr[-4000:]}")
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        for old in [] if os.environ.get("VISFLY_AMD_JIT_FLAGS") else os.listdir(JIT_DIR):          # builds of this shape against older headers
            stem = old[:-20] if len(old) > 20 else old          # libvf_chain_<slug>_<16 hex>.so
            if stem == f"libvf_chain_{_slug(shape, rollout)}" and old.endswith(".so") and old != os.path.basename(out):
                os.remove(os.path.join(JIT_DIR, old))
    finally:
        if tmpdir:
            shutil.rmtree(tmpdir, ignore_errors=True)
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()
    return out


def ensure(shape):
    """build (or find) and register the plugin of `shape` -> True when the library now has chain kernels for it"""
    if shape is None or os.environ.get("VISFLY_AMD_JIT", "1") == "0":
        return False
    if is_builtin(shape):
        return True
    if shape not in _loaded:
        from . import _lib
        _loaded[shape] = None                # (a failed build is not repeated by every policy of this shape: ~40 s of hipcc each time)
        path = build(shape)
        _lib.check(_lib.lib().vf_chain_plugin_load(path.encode()))
        _loaded[shape] = path
    return _loaded[shape] is not None


def ensure_rollout(shape, cfg):
    """the same for the roll-out plugin of `shape` under cfg = (env kind, action type, integrator, ctrl_delay)"""
    if shape is None or is_builtin(shape) or os.environ.get("VISFLY_AMD_JIT", "1") == "0":
        return False
    key = (shape, tuple(int(x) for x in cfg))
    if key not in _loaded:
        from . import _lib
        _loaded[key] = None                  # (a failed build is not retried on every roll-out: the caller warns once and falls back)
        path = build(shape, rollout=key[1])
        _lib.check(_lib.lib().vf_chain_plugin_load(path.encode()))
        _loaded[key] = path
    return _loaded[key] is not None


def ensure_bptt(shape, cfg):
    """the same for the BPTT plugin (the two persistent launches of a horizon) of an actor class `shape` under cfg = (KERNEL-side env kind,
    action type, integrator, ctrl_delay); ~2 min of hipcc on first use"""
    if shape is None or is_builtin(shape) or _heads(shape) == (1, 1) or os.environ.get("VISFLY_AMD_JIT", "1") == "0":
        return False
    key = (shape, ("bptt",) + tuple(int(x) for x in cfg))
    if key not in _loaded:
        from . import _lib
        _loaded[key] = None
        path = build(shape, rollout=key[1])
        _lib.check(_lib.lib().vf_chain_plugin_load(path.encode()))
        _loaded[key] = path
    return _loaded[key] is not None


def prebuild(verbose=False):
    """compile the PREBUILD shapes (in parallel) -> paths"""
    act = [shape_of(*v[:5], acts=v[5]) for v in PREBUILD_ACT.values()]
    jobs = ([(shape_of(*v), None) for v in list(PREBUILD.values()) + list(PREBUILD_SAC.values()) + list(PREBUILD_CRITIC.values())] +
            [(sh, None) for sh in act] +
            [(shape_of(*PREBUILD[n]), cfg) for n, cfg in PREBUILD_ROLLOUT] + [(act[0], (1, 1, 0, True))] +       # + the Tanh policy's roll-out on NavigationEnv
            [(shape_of(*{**PREBUILD, **PREBUILD_SAC}[n]), ("bptt",) + cfg) for n, cfg in PREBUILD_BPTT])
    # every chain job runs four hipcc parts of fully unrolled kernels: bound the number in flight by the cores of the build box
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), (os.cpu_count() or 4) // 4))) as pool:
        paths = list(pool.map(lambda j: build(j[0], verbose, rollout=j[1]), jobs))
    return paths
