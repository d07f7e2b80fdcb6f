"""On-device PPO for the state-vector drone envs.

Restates the reference's PPO path (utils/algorithms/PPO.py:116-337 on top of SB3 2.2.1
``OnPolicyAlgorithm.collect_rollouts`` / ``RolloutBuffer``; policy = ``CustomMultiInputActorCriticPolicy``
with ``StateExtractor`` / ``StateTargetExtractor`` MLPs, utils/policies/policies.py:195-254,
utils/policies/extractors.py:376-449,578-592,662-678) with every arithmetic piece running as a HIP
kernel behind the C-ABI: policy/value MLP forward + backward on the fp32 MFMA, squashed-Gaussian
head sampling, GAE scan, advantage normalisation, clipped-surrogate loss, grad-norm clip + Adam.
The rollout buffer lives on the device ([T][N] SoA); nothing crosses PCIe inside ``learn``.

Multi-GPU: one process per GPU, agents sharded by rank; the only exchange is ONE all-reduce of the
flat fp32 gradient buffer per optimiser step (plus the two fp64 advantage sums so that the
normalisation is over the global minibatch) through ``torch.distributed`` (RCCL on ROCm).
"""
import ctypes as C
import os
import time
from typing import Dict, List, Optional

import numpy as np
import torch as th

from . import _jit, _lib, checkpoint, parallel


def _ptr(t, off=0):
    return None if t is None else t.data_ptr() + 4 * off


# create_mlp's activation_fn (utils/policies/extractors.py:376-449) under the aliases of CustomMultiInputActorCriticPolicy
# (utils/policies/policies.py:64-69) -> VF_ACTIVATION_* of include/visfly_amd.h
ACTIVATIONS = {"relu": 1, "tanh": 2, "elu": 3, "leaky_relu": 4}
_TORCH_ACT = {1: "ReLU", 2: "Tanh", 3: "ELU", 4: "LeakyReLU"}


def activation_kind(a) -> int:
    """'relu' / 'Tanh' / nn.ELU / nn.LeakyReLU() ... -> VF_ACTIVATION_*"""
    if isinstance(a, int):
        if a in _TORCH_ACT:
            return a
        raise ValueError(f"activation kind {a}")
    name = a if isinstance(a, str) else getattr(a, "__name__", type(a).__name__)
    key = name.lower().replace("leakyrelu", "leaky_relu")
    if key not in ACTIVATIONS:
        raise NotImplementedError(f"activation_fn {name}: the MLP kernels implement {sorted(ACTIVATIONS)} (policies.py:64-69)")
    return ACTIVATIONS[key]


class _Linear:
    """one nn.Linear (+ activation) of the schedule: reads src[:, sc:sc+K], writes dst[:, dc:dc+No]; ``relu``: the activation kind
    (VF_ACTIVATION_*: 0 none, 1 ReLU, 2 Tanh, 3 ELU, 4 LeakyReLU -- the name is the ABI's)"""

    def __init__(self, K, No, relu, src, sc, dst, dc, w_off, b_off, first):
        self.K, self.No, self.relu = K, No, relu
        self.src, self.sc, self.dst, self.dc = src, sc, dst, dc
        self.w_off, self.b_off, self.first = w_off, b_off, first
        self.frozen = False       # identity pass-through (MlpPolicy ``passthrough``): no gradient, not in the backward tables


class MlpPolicy:
    """Actor-critic MLP over dict observations of flat vectors.

    ``extractor``: {obs key: hidden layer sizes} -- one ReLU MLP per key, outputs concatenated
    (StateExtractor / StateTargetExtractor); ``pi`` / ``vf``: hidden sizes of the policy / value
    trunks (MlpExtractor2); heads: action_net (-> 4) and value_net (-> 1); state-independent
    ``log_std`` (4), tanh-squashed Gaussian (policies.py:114,177-181).  All parameters live in one
    flat fp32 device buffer (weights [No][K] row-major like nn.Linear, then biases, ... , log_std).
    """

    def __init__(self, obs_dims: Dict[str, int], extractor: Dict[str, List[int]], pi: List[int], vf: List[int],
                 device, action_dim: int = 4, log_std_init: float = 0.0, seed: int = 0, ortho_init: bool = True,
                 head_dims=None, passthrough=(), log_std_param: bool = True, activation="relu", extractor_activation="relu"):
        """``head_dims`` (default (action_dim, 1)): widths of the two heads "mean" / "value" -- the SHAC actor of the reference
        has two 4-wide heads (mu and a state-dependent log_std, utils/policies/td_policies.py:230-243), its twin critic two
        1-wide ones.  ``passthrough``: input keys whose columns are appended to the features unchanged, after the extractor
        outputs -- ``th.cat([features, actions], dim=-1)`` of ContinuousCritic.forward (:137); realised as a frozen identity
        layer (W = I, b = 0: exact in fp32) whose parameters sit BEHIND the trainable ones in ``flat`` and never receive a
        gradient.  ``log_std_param`` False: no state-independent log_std parameter.
        ``activation`` / ``extractor_activation``: activation of the trunks' hidden layers (the policy's ``activation_fn``,
        policies.py:108: the reference's default there is Tanh) and of the extractor MLPs (``features_extractor_kwargs.activation_fn``,
        extractors.py:560,583: default ReLU): relu | tanh | elu | leaky_relu.  The register-chained classes built into the library are
        ReLU networks; for another activation the class is generated and compiled on first use like a non-default net_arch
        (visfly_amd/_jit.py: the activation is part of the class), else the block-tile kernels run it."""
        assert action_dim == 4, "the head kernels are written for the 4-d drone action"
        self.device = th.device(device)
        self.passthrough = [k for k in passthrough]
        self.obs_keys = list(extractor.keys()) + self.passthrough
        self.obs_dims = {k: int(obs_dims[k]) for k in self.obs_keys}
        self.head_dims = tuple(head_dims) if head_dims is not None else (action_dim, 1)
        self.act, self.ext_act = activation_kind(activation), activation_kind(extractor_activation)
        self.spec = dict(extractor={k: list(v) for k, v in extractor.items()}, pi=list(pi), vf=list(vf))
        if (self.act, self.ext_act) != (1, 1):        # (archives of ReLU networks keep the r05 spec)
            self.spec.update(activation=_TORCH_ACT[self.act].lower().replace("leakyrelu", "leaky_relu"),
                             extractor_activation=_TORCH_ACT[self.ext_act].lower().replace("leakyrelu", "leaky_relu"))
        # ---- activation buffers (name -> width) and the layer schedule ----
        self.widths: Dict[str, int] = {}
        self.layers: List[_Linear] = []
        off = 0

        def add(K, No, relu, src, sc, dst, dc, first=False):
            nonlocal off
            self.layers.append(_Linear(K, No, relu, src, sc, dst, dc, off, off + K * No, first))
            off += K * No + No

        feat_w = sum((v[-1] if v else self.obs_dims[k]) for k, v in extractor.items()) + sum(self.obs_dims[k] for k in self.passthrough)
        self.widths["feat"] = feat_w
        col = 0
        for k, hidden in extractor.items():
            src, sc, K = "obs:" + k, 0, self.obs_dims[k]
            if not hidden:
                raise NotImplementedError("identity extractor branches are not supported")
            for li, h in enumerate(hidden):
                last = li == len(hidden) - 1
                dst = "feat" if last else f"x:{k}:{li}"
                dc = col if last else 0
                if not last:
                    self.widths[dst] = h
                add(K, h, self.ext_act, src, sc, dst, dc, first=(li == 0))
                src, sc, K = dst, dc, h
            col += hidden[-1]
        pass_cols = []
        for k in self.passthrough:
            pass_cols.append((k, col))
            col += self.obs_dims[k]
        for trunk, hidden, head_dim, head in (("pi", pi, self.head_dims[0], "mean"), ("vf", vf, self.head_dims[1], "value")):
            src, sc, K = "feat", 0, feat_w
            for li, h in enumerate(hidden):
                dst = f"{trunk}:{li}"
                self.widths[dst] = h
                add(K, h, self.act, src, sc, dst, 0)
                src, sc, K = dst, 0, h
            self.widths[head] = head_dim
            add(K, head_dim, 0, src, sc, head, 0)
        self.log_std_off = off
        self.n_params = off + (action_dim if log_std_param else 0)      # trainable parameters (what Adam / the all-reduce see)
        # frozen identity layers of the pass-through inputs: executed with the extractors, parameters behind the trainable ones
        off = self.n_params
        n_ext = sum(len(v) for v in extractor.values())
        for j, (k, c) in enumerate(pass_cols):
            d = self.obs_dims[k]
            ly = _Linear(d, d, 0, "obs:" + k, 0, "feat", c, off, off + d * d, True)
            ly.frozen = True
            self.layers.insert(n_ext + j, ly)
            off += d * d + d
        self.n_total = off
        for ly in self.layers:
            if ly.K > 128 or ly.No > 128:
                raise ValueError(f"layer widths up to 128 are supported by the MFMA linear kernels (layer {ly.src} -> {ly.dst} is {ly.K} -> {ly.No}"
                                 + (f": the extractor outputs {feat_w - sum(self.obs_dims[k] for k in self.passthrough)} features (+) "
                                    f"{sum(self.obs_dims[k] for k in self.passthrough)} pass-through columns -- narrow the extractors' last layers"
                                    if self.passthrough and ly.K == feat_w else "") + ")")
        # ---- parameters, reproducible from `seed`.  ortho_init (the reference's default, policies.py:109): SB3's
        # ActorCriticPolicy._build -- orthogonal weights with gain sqrt(2) for the extractor and trunk layers, 0.01 for
        # action_net, 1 for value_net, zero biases; otherwise nn.Linear's default (kaiming-uniform) ----
        g = th.Generator().manual_seed(seed)
        flat = th.zeros(self.n_total)
        self.ortho_init = bool(ortho_init)
        for ly in self.layers:
            if ly.frozen:
                flat[ly.w_off:ly.w_off + ly.K * ly.No] = th.eye(ly.K).reshape(-1)
                continue
            if ortho_init:
                gain = {"mean": 0.01, "value": 1.0}.get(ly.dst, float(np.sqrt(2.0)))
                w = th.empty(ly.No, ly.K)
                th.nn.init.orthogonal_(w, gain=gain, generator=g)
                flat[ly.w_off:ly.w_off + ly.K * ly.No] = w.reshape(-1)
            else:
                bound = 1.0 / np.sqrt(ly.K)
                flat[ly.w_off:ly.w_off + ly.K * ly.No] = (th.rand(ly.K * ly.No, generator=g) * 2 - 1) * bound
                flat[ly.b_off:ly.b_off + ly.No] = (th.rand(ly.No, generator=g) * 2 - 1) * bound
        flat[self.log_std_off:self.n_params] = log_std_init
        self.flat = flat.to(self.device)
        self.grad = th.zeros(self.n_params, device=self.device)
        self._bufs: Dict[tuple, Dict[str, th.Tensor]] = {}
        self._gbufs: Dict[int, Dict[str, th.Tensor]] = {}
        self._scratch = None
        self._plan = self._plan_fused()
        self._descs = {}
        # chain kernels of a shape the library holds no instance of: compiled on first use (visfly_amd/_jit.py)
        # (heads (4, 1) with the log_std parameter: the PPO policies' actor-critic; (4, 4) without: the SAC-style Actor of BPTT / SHAC;
        # (1, 1) without, behind a pass-through action input: its twin critic)
        self.chain_shape = (_jit.shape_of(self.obs_dims, extractor, pi, vf, self.head_dims, self.passthrough, acts=(self.act, self.ext_act))
                            if bool(log_std_param) == (self.head_dims == (4, 1)) else None)
        self.chain_jit = False
        if self._plan is None:          # more activation buffers than a vf_mlp_desc names (VF_MLP_MAX_BUFS): layer-by-layer launches
            self.chain_shape = None
        if self.chain_shape is not None and not _jit.is_builtin(self.chain_shape) and self.device.type == "cuda":
            try:
                self.chain_jit = _jit.ensure(self.chain_shape)
            except Exception as e:      # no hipcc on this machine, a shape the compiler rejects, ...: the block-tile kernels run it
                import warnings
                warnings.warn(f"visfly_amd: no chain kernels for {_jit.name_of(self.chain_shape)} ({e})")
        self.fused = True
        self.fused_backward = True
        self._packed, self._pack_desc, self._stamp, self._packed_stamp, self.lazy_pack = None, None, 0, -1, False
        self._pack_map = None
        self._pi_only_ok = True
        self._fused_ppo = None             # None: untried, False: vf_ppo_update does not support this network
        self._tail_ok, self.tail_reason = None, ""      # False: vf_mlp_weight_grad_adam declined (reason kept)
        self._fused_twin_q = None          # the same for vf_twin_q_update (a twin critic's fused update step)
        self._steps_ok, self._steps_out = None, {}      # vf_mlp_forward_steps (forward_steps)
        self._act_fused = None             # likewise for vf_mlp_forward_act
        self._sq_part = None
        self._slot_blocks: Dict[int, tuple] = {}

    def _warn_fallback(self, what):
        """the register-chained kernels exist for the reference-default shapes (instantiated in the library) and for every shape
        visfly_amd/_jit.py can generate an instance of (1-2 observation branches, 1-4 layers per branch / trunk, widths in multiples
        of 32 up to 128: compiled on first use); any other net_arch runs on the general block-tile / per-layer kernels -- correct,
        but at roughly half the MFMA throughput.  Say so once instead of silently."""
        if not getattr(self, "_warned_fallback", False):
            self._warned_fallback = True
            import warnings
            warnings.warn(f"visfly_amd: {what} has no register-chained instance for this network "
                          f"(extractor {self.spec['extractor']}, pi {self.spec['pi']}, vf {self.spec['vf']}); "
                          "falling back to the block-tile MFMA kernels (about half the throughput)", stacklevel=3)

    def _plan_fused(self):
        """LDS layout for the one-launch forward (vf_mlp_forward): every activation gets a [64][w|1] region
        (odd row stride); regions are recycled once their last reader has run (first fit over a coalescing
        free list).  The weights do not occupy LDS (packed copy in global memory, read as the MFMA B operand).
        None if the activations do not fit the 160 KiB LDS -> layer-by-layer launches."""
        rows = 64
        odd = lambda w: ((w + 15) & ~15) + 1                      # odd stride covering the width padded to 16
        names = ["obs:" + k for k in self.obs_keys]
        if len(names) > 4 or len(self.layers) > _lib.MLP_MAX_LAYERS:
            return None
        last_read = {}
        for li, ly in enumerate(self.layers):
            last_read[ly.src] = li
        ids, off, stride, size = {}, {}, {}, {}
        free, top = [], 0

        def release(o, n):
            free.append((o, n))
            free.sort()
            merged = []
            for fo, fs in free:
                if merged and merged[-1][0] + merged[-1][1] == fo:
                    merged[-1] = (merged[-1][0], merged[-1][1] + fs)
                else:
                    merged.append((fo, fs))
            free[:] = merged

        def alloc(name, w):
            nonlocal top
            st = odd(w)
            need = rows * st
            for fi, (fo, fs) in enumerate(free):
                if fs >= need:
                    free.pop(fi)
                    if fs > need:
                        release(fo + need, fs - need)
                    off[name], stride[name], size[name] = fo, st, need
                    return
            off[name], stride[name], size[name] = top, st, need
            top += need

        for bi, n in enumerate(names):
            ids[n] = bi
            alloc(n, self.obs_dims[n[4:]])
        nxt = 4
        for li, ly in enumerate(self.layers):
            if ly.dst not in ("mean", "value") and ly.dst not in ids:
                ids[ly.dst] = nxt
                nxt += 1
                alloc(ly.dst, self.widths[ly.dst])
            for n in list(off):            # release regions nobody reads any more (never the one being written)
                if last_read.get(n, -1) <= li and n in size and n != ly.dst and n not in ("mean", "value"):
                    release(off[n], size.pop(n))
        if nxt > _lib.MLP_MAX_BUFS:
            return None
        if top * 4 > 160 * 1024:
            return None
        wt_off, wb_off, wr_off, wq_off, o = [], [], [], [], 0
        for ly in self.layers:               # packed weights: forward [round16(K)][round32(No)], data gradient [round16(No)][round32(K)],
            wt_off.append(o)                 # register-chain image: ceil(No / 32) * G blocks of 256 floats (include/visfly_amd.h)
            o += ((ly.K + 15) & ~15) * ((ly.No + 31) & ~31)
            wb_off.append(o)
            o += ((ly.No + 15) & ~15) * ((ly.K + 31) & ~31)
            wr_off.append(o)
            o += ((ly.No + 31) >> 5) * self._chain_groups(ly) * 256
            wq_off.append(o)                 # reverse-chain image: ceil(K / 32) * ceil(No / 8) blocks
            o += ((ly.K + 31) >> 5) * ((ly.No + 7) >> 3) * 256
        return dict(ids=ids, off=off, stride=stride, total=top, wt_off=wt_off, wb_off=wb_off, wr_off=wr_off, wq_off=wq_off,
                    packed_floats=o)

    @staticmethod
    def _chain_groups(ly):
        """reduction groups (of 8) of a layer's register-chain image: observation layers keep the natural k order"""
        return (ly.K + 7) >> 3 if ly.first else ((ly.K + 31) >> 5) * 4

    def _fused_desc(self, b, save: bool):
        p = self._plan
        d = _lib.MlpDesc()
        d.n_layers, d.n_inputs = len(self.layers), len(self.obs_keys)
        d.identity_mask = sum(1 << i for i, ly in enumerate(self.layers) if ly.frozen)
        for i, k in enumerate(self.obs_keys):
            d.in_dim[i] = self.obs_dims[k]
        for n, bid in p["ids"].items():
            d.lds_off[bid], d.lds_stride[bid] = p["off"][n], p["stride"][n]
        d.lds_floats = p["total"]
        for li, ly in enumerate(self.layers):
            L = d.layer[li]
            L.K, L.No, L.relu = ly.K, ly.No, int(ly.relu)
            L.src, L.src_col = p["ids"][ly.src], ly.sc
            L.dst = {"mean": _lib.MLP_OUT0, "value": _lib.MLP_OUT1}.get(ly.dst, p["ids"].get(ly.dst, 0))
            L.dst_col, L.w_off, L.b_off, L.wt_off, L.wb_off = ly.dc, ly.w_off, ly.b_off, p["wt_off"][li], p["wb_off"][li]
            L.wr_off, L.wq_off = p["wr_off"][li], p["wq_off"][li]
            if save and b is not None and ly.dst not in ("mean", "value"):
                L.save, L.save_ld = b[ly.dst].data_ptr(), b[ly.dst].shape[1]
        return d

    def mark_updated(self, packed_current: bool = False):
        """call after writing ``self.flat`` (optimiser step, load): the packed forward weights are refreshed
        by the next forward.  With ``lazy_pack`` False (default) every forward repacks -- safe for callers that
        write ``flat`` directly; the trainers set ``lazy_pack`` and call this after each step.
        ``packed_current``: the writer already refreshed the packed images (vf_adam_step with a pack map)."""
        self._stamp += 1
        if packed_current and self._packed is not None:
            self._packed_stamp = self._stamp

    def pack_map(self):
        """-> (int32 [n_params, 4] device tensor, packed buffer): for every parameter the float offsets of its copies in
        the packed forward / data-gradient weight images (-1: biases, log_std), for vf_adam_cfg.pack_map"""
        if self._plan is None:
            return None, None
        if self._pack_map is None:
            import numpy as np
            m = np.full((self.n_total, 4), -1, np.int32)
            for li, ly in enumerate(self.layers):
                n, k = np.meshgrid(np.arange(ly.No), np.arange(ly.K), indexing="ij")
                flat = ly.w_off + n * ly.K + k
                m[flat, 0] = self._plan["wt_off"][li] + k * ((ly.No + 31) & ~31) + n
                m[flat, 1] = self._plan["wb_off"][li] + n * ((ly.K + 31) & ~31) + k
                G, a, i = self._chain_groups(ly), n >> 5, n & 31
                if ly.first:          # k = 8 g + 2 j + h
                    g, j, h = k >> 3, (k & 7) >> 1, k & 1
                else:                 # k = 32 (g / 4) + 8 (g % 4) + 4 h + j
                    g, h, j = (k >> 5) * 4 + ((k & 31) >> 3), (k & 7) >> 2, k & 3
                m[flat, 2] = self._plan["wr_off"][li] + (((a * G + g) * 64 + h * 32 + i) << 2) + j
                # reverse chain: block (k / 32, n / 8), lane (n % 8) / 4 * 32 + k % 32, word n % 4
                GQ = (ly.No + 7) >> 3
                m[flat, 3] = self._plan["wq_off"][li] + ((((k >> 5) * GQ + (n >> 3)) * 64 + ((n & 7) >> 2) * 32 + (k & 31)) << 2) + (n & 3)
            self._pack_map = th.from_numpy(m).to(self.device)
            self._stamp += 1          # force one full pack (zero pads) before the incremental refreshes
            self._pack()
        return self._pack_map, self._packed

    def _pack(self):
        if self._packed is None:
            self._packed = th.empty(self._plan["packed_floats"], dtype=th.float32, device=self.device)
            self._pack_desc = self._fused_desc(None, False)
        if self.lazy_pack and self._packed_stamp == self._stamp:
            return
        _lib.check(_lib.lib().vf_mlp_pack_weights(C.byref(self._pack_desc), _ptr(self.flat), _ptr(self._packed), self._stream()))
        self._packed_stamp = self._stamp

    # -------------------------------------------------------------------------------------------
    def weight(self, ly):
        return self.flat[ly.w_off:ly.w_off + ly.K * ly.No].view(ly.No, ly.K)

    def bias(self, ly):
        return self.flat[ly.b_off:ly.b_off + ly.No]

    @property
    def log_std(self):
        return self.flat[self.log_std_off:self.n_params]

    def _buffers(self, M, slot=0):
        """activation buffers of one forward pass (kept for backward).  ``slot`` selects one of several resident
        sets for the same M -- a BPTT horizon keeps every step's activations (HBM is plentiful: 60 MB per step at
        16 384 rows) instead of recomputing the forward in the reverse sweep; gradient buffers are shared."""
        b = self._bufs.get((M, slot))
        if b is None:
            if len({m for m, _ in self._bufs}) > 4 and M not in {m for m, _ in self._bufs}:
                self._bufs.clear()
                self._gbufs.clear()
                self._descs.clear()
                self._slot_blocks.clear()
            b = {name: th.empty((M, w), dtype=th.float32, device=self.device) for name, w in self.widths.items()}
            g = self._gbufs.get(M)
            if g is None:
                g = self._gbufs[M] = {"g:" + name: th.empty((M, w), dtype=th.float32, device=self.device)
                                      for name, w in self.widths.items() if name not in ("mean", "value")}
            b.update(g)
            self._bufs[(M, slot)] = b
        return b

    def reserve_slots(self, M, n):
        """activation / gradient buffers of slots 0..n-1 stored back to back per buffer ([n][M][w]), observations
        copied in: the rows of all slots then form ONE (n M, w) matrix per buffer, which is what lets
        ``weight_grad_slots`` reduce the weight gradient of a whole BPTT horizon in one launch"""
        if self._slot_blocks.get(M, (0,))[0] >= n:
            return
        for key in [k for k in self._bufs if k[0] == M]:
            del self._bufs[key]
        self._descs = {k: v for k, v in self._descs.items() if M not in k[:2]}
        f = dict(dtype=th.float32, device=self.device)
        blk = {name: th.empty((n, M, w), **f) for name, w in self.widths.items()}
        blk.update({"g:" + name: th.empty((n, M, w), **f) for name, w in self.widths.items() if name not in ("mean", "value")})
        blk.update({"obs:" + k: th.empty((n, M, d), **f) for k, d in self.obs_dims.items()})
        self._slot_blocks[M] = (n, blk)
        for s in range(n):
            b = {name: t[s] for name, t in blk.items()}
            b["_contig"] = True
            self._bufs[(M, s)] = b

    def _stream(self):
        return _lib.current_stream(self.device)

    def forward(self, obs: Dict[str, th.Tensor], save_activations: bool = True, slot: int = 0, need_value: bool = True,
                out_value: Optional[th.Tensor] = None):
        """-> mean (M,4), value (M,1) (``need_value=False``: value is None and, where the kernel supports it, the value
        trunk is not run; ``out_value``: the value head is written there -- a (M,) row of the rollout buffer -- instead
        of the internal buffer).  One launch for the whole network when the LDS plan fits
        (``self.fused``); ``save_activations`` keeps every layer output in HBM for ``backward``
        (inference passes False and moves only observations in and heads out)."""
        M = obs[self.obs_keys[0]].shape[0]
        b = self._buffers(M, slot)
        L, st = _lib.lib(), self._stream()
        for k in self.obs_keys:
            t = obs[k]
            assert t.is_cuda and t.dtype == th.float32 and t.is_contiguous() and t.shape == (M, self.obs_dims[k])
            if b.get("_contig"):
                b["obs:" + k].copy_(t)           # reserved slots own their observation rows
            else:
                b["obs:" + k] = t
        self._last_M, self._last_slot = M, slot
        if self._plan is not None and self.fused:
            key = (M, slot, bool(save_activations))
            d = self._descs.get(key)
            if d is None:
                d = self._descs[key] = self._fused_desc(b, save_activations)
            ins = [_ptr(b["obs:" + k]) for k in self.obs_keys] + [None] * (4 - len(self.obs_keys))
            self._pack()
            skip_vf = not need_value and self._pi_only_ok
            vout = b["value"] if out_value is None else out_value
            rc = L.vf_mlp_forward(C.byref(d), _ptr(self.flat), _ptr(self._packed), ins[0], ins[1], ins[2], ins[3],
                                  _ptr(b["mean"]), None if skip_vf else _ptr(vout), M, st)
            if rc == _lib.EUNSUPPORTED and skip_vf:      # this layer table runs on the LDS kernel: it computes both heads
                self._pi_only_ok = False
                rc = L.vf_mlp_forward(C.byref(d), _ptr(self.flat), _ptr(self._packed), ins[0], ins[1], ins[2], ins[3],
                                      _ptr(b["mean"]), _ptr(vout), M, st)
            if rc:
                _lib.check(rc)
            return b["mean"], (vout if need_value else None)
        for ly in self.layers:
            X, Y = b[ly.src], b[ly.dst]
            rc = L.vf_linear_fwd(_ptr(X, ly.sc), X.shape[1], _ptr(self.flat, ly.w_off), _ptr(self.flat, ly.b_off),
                                 _ptr(Y, ly.dc), Y.shape[1], M, ly.K, ly.No, int(ly.relu), st)
            if rc:
                _lib.check(rc)
        if out_value is not None:
            out_value.copy_(b["value"].view(out_value.shape))
            return b["mean"], out_value
        return b["mean"], b["value"]

    def backward(self, d_mean: th.Tensor, d_value: Optional[th.Tensor], d_log_std: Optional[th.Tensor],
                 accumulate: bool = False, need_input_grad: bool = False, slot: Optional[int] = None):
        """fills (or, with ``accumulate``, adds into) ``self.grad`` -- flat, same layout as ``self.flat`` --
        from the head gradients.  d_value None skips the value trunk (first-order policy optimisation has
        no critic), d_mean None the policy trunk (a critic network uses the value trunk only);
        need_input_grad also returns {obs key: dLoss/d obs} (BPTT differentiates through obs)."""
        M = self._last_M
        b = self._buffers(M, self._last_slot if slot is None else slot)
        L, st = _lib.lib(), self._stream()
        if self.fused_backward and self._plan is not None:
            return self._backward_fused(b, M, d_mean, d_value, d_log_std, accumulate, need_input_grad)
        need = max(int(L.vf_linear_bwd_scratch_floats(M, ly.K, ly.No)) for ly in self.layers)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = th.empty(need, dtype=th.float32, device=self.device)
        gbuf = {} if d_mean is None else {"mean": d_mean}
        if d_value is not None:
            gbuf["value"] = d_value.view(M, self.head_dims[1])
        wgrad = L.vf_linear_bwd_weight_acc if accumulate else L.vf_linear_bwd_weight
        touched = set()
        d_in = {}
        for ly in reversed(self.layers):
            if ly.frozen:
                continue
            if d_value is None and (ly.dst == "value" or ly.dst.startswith("vf:")):
                continue
            if d_mean is None and (ly.dst == "mean" or ly.dst.startswith("pi:")):
                continue
            dY = gbuf.get(ly.dst, b.get("g:" + ly.dst))
            Y, X = b[ly.dst], b[ly.src]
            ym = _ptr(Y, ly.dc) if ly.relu else None
            rc = wgrad(_ptr(dY, ly.dc), dY.shape[1], ym, Y.shape[1], _ptr(X, ly.sc), X.shape[1],
                       _ptr(self.grad, ly.w_off), _ptr(self.grad, ly.b_off), M, ly.K, ly.No, _ptr(self._scratch), int(ly.relu), st)
            if rc:
                _lib.check(rc)
            if ly.first and not need_input_grad:
                continue
            if ly.first:
                dX = d_in.setdefault(ly.src[4:], th.empty((M, ly.K), dtype=th.float32, device=self.device))
            else:
                dX = b["g:" + ly.src]
            key = (ly.src, ly.sc)
            rc = L.vf_linear_bwd_data(_ptr(dY, ly.dc), dY.shape[1], ym, Y.shape[1], _ptr(self.flat, ly.w_off),
                                      _ptr(dX, ly.sc), dX.shape[1], M, ly.K, ly.No, 1 if key in touched else 0, int(ly.relu), st)
            if rc:
                _lib.check(rc)
            touched.add(key)
        if d_log_std is not None:
            if accumulate:
                self.grad[self.log_std_off:self.n_params] += d_log_std
            else:
                self.grad[self.log_std_off:self.n_params] = d_log_std
        return d_in

    def _bwd_desc(self, b, M, d_mean, d_value, need_input_grad):
        """vf_mlp_bwd_desc of the reverse sweep over buffer set b: layers in reverse order, a trunk without head
        gradient skipped -> (desc, {obs key: dLoss/d obs tensor})"""
        gbuf = {} if d_mean is None else {"mean": d_mean}
        if d_value is not None:
            gbuf["value"] = d_value.view(-1, self.head_dims[1])
        d = _lib.MlpBwdDesc()
        d.n_fold = self.log_std_off
        touched, d_in, n = set(), {}, 0
        for ly in reversed(self.layers):
            if ly.frozen:
                continue
            if d_value is None and (ly.dst == "value" or ly.dst.startswith("vf:")):
                continue
            if d_mean is None and (ly.dst == "mean" or ly.dst.startswith("pi:")):
                continue
            dY = gbuf.get(ly.dst, b.get("g:" + ly.dst))
            Y, X = b[ly.dst], b[ly.src]
            e = d.layer[n]
            n += 1
            e.K, e.No, e.w_off, e.b_off = ly.K, ly.No, ly.w_off, ly.b_off
            li = self.layers.index(ly)
            e.wb_off, e.wq_off = self._plan["wb_off"][li], self._plan["wq_off"][li]
            e.dY, e.ld_dy = _ptr(dY, ly.dc), dY.shape[-1]
            e.Y, e.ld_y, e.act = (_ptr(Y, ly.dc) if ly.relu else None), Y.shape[-1], int(ly.relu)
            e.X, e.ld_x = _ptr(X, ly.sc), X.shape[-1]
            e.need_dx = 0
            if ly.first and not need_input_grad:
                continue
            if ly.first:
                dX = d_in.setdefault(ly.src[4:], th.empty((M, ly.K), dtype=th.float32, device=self.device))
            else:
                dX = b["g:" + ly.src]
            key = (ly.src, ly.sc)
            e.dX, e.ld_dx, e.need_dx = _ptr(dX, ly.sc), dX.shape[-1], (2 if key in touched else 1)
            touched.add(key)
        d.n_layers = n
        return d, d_in

    def forward_act(self, obs, eps, action, slot=0):
        """policy trunk + action head in one launch (vf_mlp_forward_act): ``action`` (M,4) <- tanh(mean + exp(log_std) eps),
        activations saved for the reverse pass of `slot`.  -> False when the network is not a register-chained class."""
        if self._act_fused is False or self._plan is None or not self.fused:
            return False
        M = action.shape[0]
        b = self._buffers(M, slot)
        ins, copies = [], []
        for k in self.obs_keys:
            t = obs[k]
            assert t.is_cuda and t.dtype == th.float32 and t.is_contiguous() and t.shape == (M, self.obs_dims[k])
            ins.append(_ptr(t))
            if b.get("_contig"):
                copies.append(_ptr(b["obs:" + k]))       # reserved slots own their observation rows: written by the kernel
            else:
                b["obs:" + k] = t
                copies.append(None)
        ins += [None] * (2 - len(ins))
        copies += [None] * (2 - len(copies))
        self._last_M, self._last_slot = M, slot
        key = (M, slot, True)
        d = self._descs.get(key)
        if d is None:
            d = self._descs[key] = self._fused_desc(b, True)
        self._pack()
        rc = _lib.lib().vf_mlp_forward_act(C.byref(d), _ptr(self.flat), _ptr(self._packed), ins[0], ins[1], _ptr(self.log_std),
                                           _ptr(eps), _ptr(action), copies[0], copies[1], M, self._stream())
        if rc == _lib.EUNSUPPORTED:
            self._act_fused = False
            return False
        if rc:
            _lib.check(rc)
        return True

    def backward_data_act(self, d_action, action, eps, g_log_std, d_mean, slot):
        """``backward_data`` with the action head's reverse fused in (vf_mlp_backward_data_act): d_mean (M,4) is written for
        ``weight_grad_slots``, g_log_std (M,4) accumulated; -> {obs key: dLoss/d obs}"""
        M = d_action.shape[0]
        b = self._buffers(M, slot)
        cached = self._descs.get(("bwd_data", M, slot)) if b.get("_contig") else None
        if cached is None:
            d, d_in = self._bwd_desc(b, M, d_mean, None, True)
            if b.get("_contig"):
                self._descs[("bwd_data", M, slot)] = (d, d_in)
        else:
            d, d_in = cached
            d.layer[0].dY = _ptr(d_mean)
        self._pack()
        _lib.check(_lib.lib().vf_mlp_backward_data_act(C.byref(d), _ptr(self._packed), _ptr(d_action), _ptr(action), _ptr(self.log_std),
                                                       _ptr(eps), _ptr(g_log_std), M, self._stream()))
        return d_in

    def backward_data_supported(self, M, slot=0, both_heads=False):
        """can ``backward_data`` (policy trunk [+ second trunk: ``both_heads``] + observation gradient) run for this network?"""
        if self._plan is None or not self.fused_backward:
            return False
        b = self._buffers(M, slot)
        dm = th.empty((M, self.head_dims[0]), dtype=th.float32, device=self.device)
        dv = th.empty((M, self.head_dims[1]), dtype=th.float32, device=self.device) if both_heads else None
        d, _ = self._bwd_desc(b, M, dm, dv, True)
        return bool(_lib.lib().vf_mlp_backward_data_supported(C.byref(d)))

    def _head_entries(self, both_heads):
        """indices of the (mean, value) head layers in the reverse layer table ``_bwd_desc`` builds (reversed layer order, a
        trunk without head gradient skipped)"""
        order = [ly.dst for ly in reversed(self.layers)
                 if not ly.frozen and (both_heads or not (ly.dst == "value" or ly.dst.startswith("vf:")))]
        return order.index("mean"), (order.index("value") if both_heads else None)

    def backward_data(self, d_mean, slot, d_value=None):
        """reverse chain of slot `slot` only (policy trunk; with ``d_value`` both trunks -- the reference's Actor, whose second
        head is log_std): masked layer gradients stay in the slot's g: buffers for ``weight_grad_slots``; -> {obs key: dLoss/d obs}
        (for reserved slots the returned tensors are reused by the next call on the same slot)"""
        M = d_mean.shape[0]
        b = self._buffers(M, slot)
        key = ("bwd_data", M, slot, d_value is not None)
        cached = self._descs.get(key) if b.get("_contig") else None    # reserved slots: fixed buffers
        if cached is None:
            d, d_in = self._bwd_desc(b, M, d_mean, d_value, True)
            if b.get("_contig"):
                self._descs[key] = (d, d_in)
        else:
            d, d_in = cached
            im, iv = self._head_entries(d_value is not None)
            d.layer[im].dY = _ptr(d_mean)          # the heads' gradients are the caller's tensors
            if iv is not None:
                d.layer[iv].dY = _ptr(d_value)
        self._pack()
        _lib.check(_lib.lib().vf_mlp_backward_data(C.byref(d), _ptr(self._packed), M, self._stream()))
        return d_in

    def weight_grad_slots(self, M, n, d_mean_all, accumulate=False, d_value_all=None, lo=0):
        """weight / bias gradients of the policy trunk (+ the second trunk when ``d_value_all`` is given) + extractors summed over
        slots lo..lo+n-1 (reserved with ``reserve_slots``; d_mean_all (n, M, 4) [d_value_all (n, M, w)] hold the head gradients the
        ``backward_data`` calls were given for THOSE slots)"""
        nblk, blk = self._slot_blocks[M]
        assert lo + n <= nblk and d_mean_all.shape == (n, M, 4) and d_mean_all.is_contiguous()
        assert d_value_all is None or (d_value_all.shape == (n, M, self.head_dims[1]) and d_value_all.is_contiguous())
        # one launch over 1 M rows runs at 67 TF/s, two over 512 K rows each at 76 (profiles/r06_bptt_wgrad_chunks.txt: not a cache effect --
        # the rows are as fast cold as hot): a horizon above WGRAD_SPLIT_ROWS is reduced in equal runs of whole slots, in slot order
        cap = int(os.environ.get("VISFLY_AMD_WGRAD_SPLIT_ROWS", "524288"))
        if n > 1 and n * M > cap > 0:
            per = max(1, min(n - 1, cap // M))
            per = (n + ((n + per - 1) // per) - 1) // ((n + per - 1) // per)        # equal runs
            for s0 in range(0, n, per):
                k = min(per, n - s0)
                self.weight_grad_slots(M, k, d_mean_all[s0:s0 + k], accumulate or s0 > 0, None if d_value_all is None else d_value_all[s0:s0 + k], lo + s0)
            return
        b = {name: t[lo:lo + n].reshape(-1, t.shape[-1]) for name, t in blk.items()}
        d, _ = self._bwd_desc(b, n * M, d_mean_all.view(-1, 4), None if d_value_all is None else d_value_all.view(n * M, -1), False)
        L = _lib.lib()
        need = int(L.vf_mlp_backward_partial_floats(C.byref(d), n * M))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = th.empty(need, dtype=th.float32, device=self.device)
        _lib.check(L.vf_mlp_weight_grad(C.byref(d), _ptr(self._scratch), _ptr(self.grad), n * M, 1 if accumulate else 0, self._stream()))

    def bucket_split(self):
        """first flat-parameter offset of the trunks: [0, split) = the extractor MLPs' parameters, [split, n_params) = both trunks, the
        heads and log_std -- the two gradient buckets of the two-bucket exchange (PPO.grad_buckets)"""
        return min(ly.w_off for ly in self.layers if not ly.frozen and not (ly.first or ly.src.startswith("x:") or ly.dst == "feat"))

    def ppo_update(self, obs, actions, old_lp, adv, ret, loss_cfg, stats, loss_scratch, want_sumsq=False, row_index=None, tail=None,
                   between=None):
        """forward + PPO loss + reverse chain in one launch, then the weight gradients into ``self.grad`` (vf_ppo_update +
        vf_mlp_weight_grad).  -> False when the network is not one of the register-chained classes (the caller then
        runs forward / vf_ppo_loss / backward).  ``want_sumsq``: -> (fp64 partials tensor, count) of the squared norm of the
        gradient the fold wrote, for vf_adam_cfg.sumsq_partials (no separate grad-norm launch).
        ``tail`` (a ``_lib.WgradTail``: parameters, Adam moments and configuration, sync words): the weight-gradient launch also folds,
        forms the gradient norm, clips and runs Adam (vf_mlp_weight_grad_adam: the optimiser step is TWO launches) -> "adam"; when the
        library declines (VF_EUNSUPPORTED) the call continues as ``want_sumsq`` and the caller runs vf_adam_step.
        ``between`` (callable): the weight gradients are formed in TWO launch pairs -- trunks + heads first, then the extractor MLPs -- and
        ``between()`` runs after the first pair was enqueued (the trainer starts the first bucket's all-reduce there).
        ``row_index`` (int64, M entries; vf_ppo_loss_cfg.row_index): obs / actions / old_lp / ret (and loss_cfg.old_value) are the
        WHOLE rollout buffer and row m of the minibatch is their row row_index[m] -- no shuffled copy; ``adv`` stays in minibatch order."""
        if self._fused_ppo is False or self._plan is None or not (self.fused and self.fused_backward):
            return False
        M = actions.shape[0] if row_index is None else row_index.numel()
        b = self._buffers(M, 0)
        L, st = _lib.lib(), self._stream()
        whole = {}
        for k in self.obs_keys:
            t = obs[k]
            assert t.is_cuda and t.dtype == th.float32 and t.is_contiguous() and t.shape[1:] == (self.obs_dims[k],)
            if row_index is not None:
                # the launch leaves the minibatch's observation rows, in minibatch order, where the weight gradients read them
                whole[k] = t
                if not b.get("_contig"):        # (reserved slots own their observation rows already)
                    own = b.get("_own:" + k)
                    if own is None:
                        own = b["_own:" + k] = th.empty((M, self.obs_dims[k]), dtype=th.float32, device=self.device)
                    b["obs:" + k] = own
                continue
            assert t.shape[0] == M
            if b.get("_contig"):
                b["obs:" + k].copy_(t)
            else:
                b["obs:" + k] = t
        if row_index is not None:
            assert row_index.dtype == th.int64 and row_index.is_contiguous() and adv.numel() == M
            loss_cfg.row_index = row_index.data_ptr()
            loss_cfg.obs_copy0 = _ptr(b["obs:" + self.obs_keys[0]])
            loss_cfg.obs_copy1 = _ptr(b["obs:" + self.obs_keys[1]]) if len(self.obs_keys) > 1 else None
        self._last_M, self._last_slot = M, 0
        key = (M, 0, True)
        d = self._descs.get(key)
        if d is None:
            d = self._descs[key] = self._fused_desc(b, True)
        if "d:mean" not in b:
            b["d:mean"], b["d:value"] = th.empty((M, 4), dtype=th.float32, device=self.device), th.empty(M, dtype=th.float32, device=self.device)
        cached = self._descs.get(("ppo_bwd", M))
        if cached is None:
            bd = self._bwd_desc(b, M, b["d:mean"], b["d:value"], False)[0]
            # entries whose X is an observation: that pointer is the caller's tensor and changes from call to call
            firsts = [(i, ly.src) for i, ly in enumerate(reversed(self.layers)) if ly.first]
            cached = self._descs[("ppo_bwd", M)] = (bd, firsts)
        bd, firsts = cached
        for i, src in firsts:
            bd.layer[i].X = _ptr(b[src])
        ins = [_ptr(whole[k] if row_index is not None else b["obs:" + k]) for k in self.obs_keys] + [None] * (2 - len(self.obs_keys))
        self._pack()
        # want_sumsq: the loss-statistic rows are folded by the weight-gradient fold launch (one launch less)
        want_sumsq = want_sumsq or tail is not None
        rc = L.vf_ppo_update(C.byref(d), C.byref(bd), _ptr(self.flat), _ptr(self._packed), ins[0], ins[1], _ptr(self.log_std),
                             _ptr(actions), _ptr(old_lp), _ptr(adv), _ptr(ret), None if want_sumsq else _ptr(stats), M,
                             C.byref(loss_cfg), _ptr(loss_scratch), st)
        if rc == _lib.EUNSUPPORTED:
            self._fused_ppo = False
            self._warn_fallback("vf_ppo_update")
            return False
        if rc:
            _lib.check(rc)
        need = int(L.vf_mlp_backward_partial_floats(C.byref(bd), M))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = th.empty(need, dtype=th.float32, device=self.device)
        if want_sumsq:
            nb = int(L.vf_mlp_weight_grad_fold_blocks(C.byref(bd)))
            if self._sq_part is None or self._sq_part.numel() < nb:
                self._sq_part = th.empty(nb, dtype=th.float64, device=self.device)
            ls = _lib.StatsFold(_ptr(loss_scratch), (M + 31) // 32, 0, _ptr(stats), loss_cfg.d_log_std_out, loss_cfg.stats_accum)
            if tail is not None and self._tail_ok is not False:
                tail.adam.sumsq_partials = self._sq_part.data_ptr()
                rc = L.vf_mlp_weight_grad_adam(C.byref(bd), _ptr(self._scratch), _ptr(self.grad), M, 0, C.byref(ls), C.byref(tail), st)
                if rc == 0:
                    return "adam"
                if rc != _lib.EUNSUPPORTED:
                    _lib.check(rc)
                self._tail_ok = False          # the library said why (vf_last_error); the separate fold + Adam launches from here on
                self.tail_reason = L.vf_last_error().decode()
            _lib.check(L.vf_mlp_weight_grad_sumsq(C.byref(bd), _ptr(self._scratch), _ptr(self.grad), M, 0, self._sq_part.data_ptr(),
                                                  C.byref(ls), st))
            return self._sq_part, nb
        if between is not None:        # two launch pairs on the plan of the whole table (same bits): trunk / head layers, then the extractors'
            split = self.bucket_split()
            first = sum(1 << i for i in range(bd.n_layers) if bd.layer[i].w_off >= split)
            _lib.check(L.vf_mlp_weight_grad_layers(C.byref(bd), _ptr(self._scratch), _ptr(self.grad), M, 0, first, st))
            between()
            _lib.check(L.vf_mlp_weight_grad_layers(C.byref(bd), _ptr(self._scratch), _ptr(self.grad), M, 0, ((1 << bd.n_layers) - 1) & ~first, st))
            return True
        _lib.check(L.vf_mlp_weight_grad(C.byref(bd), _ptr(self._scratch), _ptr(self.grad), M, 0, st))
        return True

    def forward_steps(self, obs, m_step, n_steps):
        """inference forward of ``n_steps`` consecutive blocks of ``m_step`` rows in one launch, every row as ``forward`` computes it
        in a launch over its block alone (vf_mlp_forward_steps) -> (mean, value) of (n_steps m_step) rows, or None when the library
        has no register-chained class for this network / row count (the caller loops ``forward``).  The returned tensors are the
        policy's own output buffers for this row count: the next call with the same (m_step, n_steps) overwrites them."""
        if self._steps_ok is False or self._plan is None or not self.fused or m_step % 32:
            return None
        M = m_step * n_steps
        for k in self.obs_keys:
            t = obs[k]
            assert t.is_cuda and t.dtype == th.float32 and t.is_contiguous() and t.shape == (M, self.obs_dims[k])
        if self._pack_desc is None:
            self._pack_desc = self._fused_desc(None, False)
        out = self._steps_out.get(M)
        if out is None:
            f = dict(dtype=th.float32, device=self.device)
            out = self._steps_out[M] = (th.empty((M, self.head_dims[0]), **f), th.empty((M, self.head_dims[1]), **f))
        ins = [_ptr(obs[k]) for k in self.obs_keys] + [None] * (3 - len(self.obs_keys))
        self._pack()
        rc = _lib.lib().vf_mlp_forward_steps(C.byref(self._pack_desc), _ptr(self.flat), _ptr(self._packed), ins[0], ins[1], ins[2],
                                             _ptr(out[0]), _ptr(out[1]), int(m_step), int(n_steps), self._stream())
        if rc == _lib.EUNSUPPORTED:
            self._steps_ok = False
            return None
        if rc:
            _lib.check(rc)
        return out

    def twin_q_update(self, obs, target, loss_out, m_global):
        """a twin critic's update step up to the flat gradient (shac.py:267-270): forward + mse_loss(target, min(Q1, Q2)) + reverse
        chain in one launch (vf_twin_q_update), then the weight gradients into ``self.grad``.  ``obs``: the extractor's observations
        + the pass-through "action" rows; ``loss_out`` (1,) device tensor.  -> False when the network is not the register-chained
        critic class (the caller then runs forward / vf_twin_q_loss / backward)."""
        if self._fused_twin_q is False or self._plan is None or not (self.fused and self.fused_backward) or len(self.obs_keys) != 2:
            return False
        M = target.shape[0]
        b = self._buffers(M, 0)
        L, st = _lib.lib(), self._stream()
        for k in self.obs_keys:
            t = obs[k]
            assert t.is_cuda and t.dtype == th.float32 and t.is_contiguous() and t.shape == (M, self.obs_dims[k])
            if b.get("_contig"):
                b["obs:" + k].copy_(t)
            else:
                b["obs:" + k] = t
        self._last_M, self._last_slot = M, 0
        key = (M, 0, True)
        d = self._descs.get(key)
        if d is None:
            d = self._descs[key] = self._fused_desc(b, True)
        if "d:q0" not in b:
            b["d:q0"] = th.empty((M, self.head_dims[0]), dtype=th.float32, device=self.device)
            b["d:q1"] = th.empty((M, self.head_dims[1]), dtype=th.float32, device=self.device)
        cached = self._descs.get(("twin_q_bwd", M))
        if cached is None:
            bd = self._bwd_desc(b, M, b["d:q0"], b["d:q1"], False)[0]
            firsts = [(i, ly.src) for i, ly in enumerate(l for l in reversed(self.layers) if not l.frozen) if ly.first]
            need = int(L.vf_twin_q_update_scratch_doubles(M))
            cached = self._descs[("twin_q_bwd", M)] = (bd, firsts, th.empty(need, dtype=th.float64, device=self.device))
        bd, firsts, scr = cached
        for i, src in firsts:                 # entries whose X is an observation: the caller's tensor, may change from call to call
            bd.layer[i].X = _ptr(b[src])
        self._pack()
        rc = L.vf_twin_q_update(C.byref(d), C.byref(bd), _ptr(self.flat), _ptr(self._packed), _ptr(b["obs:" + self.obs_keys[0]]),
                                _ptr(b["obs:" + self.obs_keys[1]]), _ptr(target), _ptr(loss_out), scr.data_ptr(), M, int(m_global), st)
        if rc == _lib.EUNSUPPORTED:
            self._fused_twin_q = False
            self._warn_fallback("vf_twin_q_update")
            return False
        if rc:
            _lib.check(rc)
        need = int(L.vf_mlp_backward_partial_floats(C.byref(bd), M))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = th.empty(need, dtype=th.float32, device=self.device)
        _lib.check(L.vf_mlp_weight_grad(C.byref(bd), _ptr(self._scratch), _ptr(self.grad), M, 0, st))
        return True

    def _backward_fused(self, b, M, d_mean, d_value, d_log_std, accumulate, need_input_grad):
        """the same sweep as the layer-by-layer path, as ONE launch + one fold (vf_mlp_backward): a block owns
        its 64-row tiles for every layer, so g:* buffers written by one entry are read by the next without
        a grid-wide barrier."""
        L, st = _lib.lib(), self._stream()
        d, d_in = self._bwd_desc(b, M, d_mean, d_value, need_input_grad)
        need = int(L.vf_mlp_backward_partial_floats(C.byref(d), M))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = th.empty(need, dtype=th.float32, device=self.device)
        self._pack()
        _lib.check(L.vf_mlp_backward(C.byref(d), _ptr(self._packed), _ptr(self._scratch), _ptr(self.grad), M,
                                     1 if accumulate else 0, st))
        if d_log_std is not None:
            if accumulate:
                self.grad[self.log_std_off:self.n_params] += d_log_std
            else:
                self.grad[self.log_std_off:self.n_params] = d_log_std
        return d_in

    # -------------------------------------------------------------------------------------------
    def to_torch(self):
        """equivalent torch.nn module (fp32) sharing NO storage -- the plain-PyTorch reference the
        numerics tests compare against"""
        import torch.nn as nn
        pol = self

        class Ref(nn.Module):
            def __init__(s):
                super().__init__()
                s.lin = nn.ModuleList()
                for ly in pol.layers:
                    m = nn.Linear(ly.K, ly.No)
                    m.weight.data.copy_(pol.weight(ly).cpu())
                    m.bias.data.copy_(pol.bias(ly).cpu())
                    s.lin.append(m)
                s.log_std = nn.Parameter(pol.log_std.detach().cpu().clone())

            def forward(s, obs):
                acts = {"obs:" + k: v for k, v in obs.items()}
                M = next(iter(obs.values())).shape[0]
                for ly, m in zip(pol.layers, s.lin):
                    x = acts[ly.src][:, ly.sc:ly.sc + ly.K]
                    y = m(x)
                    y = getattr(nn, _TORCH_ACT[ly.relu])()(y) if ly.relu else y
                    if ly.dst == "feat":
                        if "feat" not in acts:
                            acts["feat"] = th.zeros((M, pol.widths["feat"]), dtype=y.dtype, device=y.device)
                        acts["feat"] = th.cat([acts["feat"][:, :ly.dc], y, acts["feat"][:, ly.dc + ly.No:]], dim=1)
                    else:
                        acts[ly.dst] = y
                return acts["mean"], acts["value"]

            def flat_grad(s):
                g = th.zeros(pol.n_params)
                for ly, m in zip(pol.layers, s.lin):
                    if ly.frozen or m.weight.grad is None:
                        continue
                    g[ly.w_off:ly.w_off + ly.K * ly.No] = m.weight.grad.reshape(-1)
                    g[ly.b_off:ly.b_off + ly.No] = m.bias.grad
                if s.log_std.grad is not None and pol.n_params > pol.log_std_off:
                    g[pol.log_std_off:] = s.log_std.grad
                return g

        return Ref()


class RolloutBuffer:
    """[T][N] device-resident rollout storage (SB3 RolloutBuffer semantics; mirror at
    utils/algorithms/common.py:46-215)"""

    def __init__(self, T, N, obs_dims: Dict[str, int], device):
        f = dict(dtype=th.float32, device=device)
        self.T, self.N = T, N
        self.obs = {k: th.zeros((T, N, d), **f) for k, d in obs_dims.items()}
        self.actions = th.zeros((T, N, 4), **f)
        self.rewards, self.values, self.log_probs = th.zeros((T, N), **f), th.zeros((T, N), **f), th.zeros((T, N), **f)
        self.episode_starts = th.zeros((T, N), **f)
        self.advantages, self.returns = th.zeros((T, N), **f), th.zeros((T, N), **f)


class PPO:
    """PPO.learn / collect_rollouts / train of the reference (utils/algorithms/PPO.py:116-337)."""

    def __init__(self, env, n_steps=256, batch_size=25600, n_epochs=5, gamma=0.99, gae_lambda=0.95, clip_range=0.2,
                 ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, learning_rate=1e-4, weight_decay=1e-5,
                 normalize_advantage=True, policy_kwargs: Optional[dict] = None, seed=0, adam_eps=1e-8,
                 betas=(0.9, 0.999), target_kl=None, policy=None, verbose=0, device=None, clip_range_vf=None):
        # `policy`, `verbose`, `device`: accepted so that the `algorithm:` block of the reference's YAMLs can be passed
        # as **kwargs (exps/examples/alg_cfgs/*/PPO.yaml); the policy is always the MFMA MlpPolicy on the env's device
        if policy not in (None, "CustomMultiInputPolicy", "MultiInputPolicy", "MlpPolicy"):
            raise NotImplementedError(f"policy {policy}: vector-observation actor-critic policies only")
        self.env = env
        env.tensor_output, env.requires_grad = True, False        # PPO.py:80-82 forces the non-grad path
        self.device = env.device
        self.n_envs = env.num_envs
        self.n_steps, self.batch_size, self.n_epochs = n_steps, batch_size, n_epochs
        # learning_rate / clip_range / clip_range_vf: a float or, as SB3's get_schedule_fn accepts them, a callable of
        # progress_remaining (1 at the start of learn(), 0 at its end; PPO.py:150-152,184-189) evaluated once per train() call
        self.gamma, self.gae_lambda, self.clip_range = gamma, gae_lambda, clip_range
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.lr_schedule, self.weight_decay, self.adam_eps, self.betas = learning_rate, weight_decay, adam_eps, betas
        self.normalize_advantage, self.target_kl, self.seed = normalize_advantage, target_kl, seed
        self.clip_range_vf = clip_range_vf                          # PPO.py:237-243; None = no value clipping
        self._current_progress_remaining = 1.0
        # TimeLimit bootstrap valued once per rollout (collect_rollouts) from the terminal rows the step kernel wrote; envs that
        # assemble their observation on the host (RacingEnv2: 16 gate-relative columns) have no such kernel rows -> valued per step
        self.defer_bootstrap, self._boot = (not getattr(env, "_HOST_OBS", False)) or getattr(env, "_OBS_W", 13) != 13, None
        if hasattr(env, "obs_gate_exact"):       # RacingEnv / RacingEnv2: this policy does not read "gate"; its "state" rows use the agent's
            env.obs_gate_exact = False           # current gate (what the persistent roll-out forms), as under BPTT / SHAC
        self.fused_rollout = True           # collect_rollouts as one persistent launch where the library has the kernel (vf_ppo_rollout)
        self._last_obs = None
        if not getattr(env, "_is_initial", False):
            self._last_obs = env.reset()
        obs = env.get_observation()
        self.obs_keys = [k for k in obs.keys() if k in ("state", "target")]
        obs_dims = {k: obs[k].shape[1] for k in self.obs_keys}
        pk = checkpoint.policy_kwargs_from_reference(policy_kwargs, self.obs_keys)
        self.weight_decay = pk.get("weight_decay", self.weight_decay)   # optimizer_kwargs.weight_decay of the YAMLs
        extractor = pk.get("extractor", {k: [128, 64] for k in self.obs_keys})
        self.policy = MlpPolicy(obs_dims, extractor, pk.get("pi", [64, 64]), pk.get("vf", [64, 64]), self.device,
                                log_std_init=pk.get("log_std_init", 0.0), seed=seed, ortho_init=pk.get("ortho_init", True),
                                activation=pk.get("activation", "relu"), extractor_activation=pk.get("extractor_activation", "relu"))
        self.policy.lazy_pack = True        # this trainer calls mark_updated() after every optimiser step
        self.world, self.rank = parallel.world_size(), parallel.rank()
        # every rank must start from the SAME parameters (only gradients are exchanged afterwards) and draw DIFFERENT
        # exploration noise for its agents: rank 0's initial weights go to everybody, the rank goes into the Philox key
        parallel.broadcast_(self.policy.flat)
        self.policy.mark_updated()
        self._noise_key = (int(seed) ^ (self.rank << 32)) & (2 ** 64 - 1)
        self.buf = RolloutBuffer(n_steps, self.n_envs, obs_dims, self.device)
        n = self.policy.n_params
        dev = self.device
        self.exp_avg, self.exp_avg_sq = th.zeros(n, device=dev), th.zeros(n, device=dev)
        # loss-statistic partial rows of a minibatch (16 floats per 32-row tile; vf_ppo_loss / vf_ppo_update) + 4096 floats of reduction scratch
        self._scratch = th.zeros(16 * max(1024, (min(batch_size, n_steps * self.n_envs) + 31) // 32) + 4096, device=dev)
        # flat gradient and the 16 loss statistics in ONE buffer: an optimiser step on several GPUs is exactly one all-reduce
        self._gbuf = th.zeros(n + 16, device=dev)
        self.policy.grad = self._gbuf[:n]
        self._stats = self._gbuf[n:]
        self._ep_stats = th.zeros(4, dtype=th.float64, device=dev)   # episodes, sum return, sum length, successes
        self._sumsq = th.zeros(1, device=dev)
        self._sums = th.zeros(2, dtype=th.float64, device=dev)
        self._opt_step = 0
        self._n_updates = 0            # SB3's counter: epochs (PPO.py:294), what train/n_updates logs
        self._sample_step = 0
        self.num_timesteps = 0
        self._last_starts = th.ones(self.n_envs, device=dev)
        self._shuf = None
        # fold + gradient norm + clip + Adam inside the weight-gradient launch (vf_mlp_weight_grad_adam): single-GPU steps without a
        # target_kl check between backward and optimizer.step().  Bit-identical to the separate launches and, measured, 4.6 us per
        # optimiser step SLOWER (two device-wide meetings of 1 000 lone waves cost 3.5 us each, the in-kernel fold reads the same 20 MB
        # of partials, and the expensive launch boundaries are the ones behind the two big kernels, which stay:
        # profiles/r06_fused_tail.txt) -- off unless VISFLY_AMD_FUSED_TAIL=1
        self.fused_tail = os.environ.get("VISFLY_AMD_FUSED_TAIL", "0") == "1"
        # gradient exchange of a multi-GPU step.  1 (default): ONE all-reduce of [gradient | 16 loss statistics] between the fold and
        # Adam.  2 (VISFLY_AMD_GRAD_BUCKETS=2): the weight gradients are formed in two launch pairs -- trunks + heads, then the extractor
        # MLPs -- and the first bucket ([trunks | log_std | statistics]) is all-reduced on a second stream UNDER the second pair; the
        # second bucket follows on that stream, Adam waits for both.  Same sums, same bits (tests/test_parallel_*); whether it pays is a
        # question for real xGMI links: 176 KB is latency, not bandwidth (bench.py prints both modes for N > 1)
        self.grad_buckets = int(os.environ.get("VISFLY_AMD_GRAD_BUCKETS", "1"))
        self._xstream, self._xev = None, None
        self._tail_sync = th.zeros(_lib.WGRAD_SYNC_WORDS, dtype=th.int32, device=dev)
        self._tail_launches = 0
        self.index_minibatches = False     # True: train() reads its minibatches through the permutation slice (vf_ppo_loss_cfg.row_index) instead of a shuffled copy -- measured 2 % slower, see train()
        self.logs: Dict[str, float] = {}

    def _stream(self):
        return _lib.current_stream(self.device)

    def _now(self, v):
        """a schedule's value at the current progress_remaining (floats are constant schedules)"""
        return None if v is None else float(v(self._current_progress_remaining) if callable(v) else v)

    @property
    def lr(self):
        return self._now(self.lr_schedule)

    @lr.setter
    def lr(self, v):
        self.lr_schedule = v

    def _bootstrap_list(self):
        """compact list of the rollout's truncated rows (vf_rollout_post_collect): an agent is truncated at most once per
        max_episode_steps steps, plus the episode it is in when the rollout starts"""
        if self._boot is None:
            N, dev = self.n_envs, self.device
            per_agent = self.n_steps // max(int(getattr(self.env, "max_episode_steps", self.n_steps)), 1) + 2
            cap = (N * per_agent + 8191) // 8192 * 8192
            w1 = self.policy.obs_dims.get("target", 0) if "target" in self.obs_keys else 0
            self._boot = {"cap": cap, "cursor": th.zeros(1, dtype=th.int32, device=dev), "idx": th.zeros(cap, dtype=th.int32, device=dev),
                          "rows0": th.zeros((cap, self.policy.obs_dims["state"]), device=dev),
                          "rows1": th.zeros((cap, w1), device=dev) if w1 else None,
                          "stat": th.zeros((N, 4), device=dev)}
        self._boot["cursor"].zero_()
        self._boot["stat"].zero_()
        return self._boot

    def save(self, path: str):
        """zip archive in the layout of SB3's BaseAlgorithm.save (PPO.py:418-430): see checkpoint.py"""
        return checkpoint.save(self, path)

    def set_parameters(self, path: str, load_optimizer: bool = True):
        return checkpoint.load_into(self, path, load_optimizer)

    @classmethod
    def load(cls, path: str, env, **kwargs):
        """PPO.py:432-572: re-create the trainer on `env`, then load policy (and optimiser) state; also reads
        archives written by the reference (policy.pth with the reference's parameter names)"""
        return checkpoint.load_into(cls(env, **checkpoint.ctor_kwargs_from_archive(path, kwargs)), path)

    # ------------------------------------------------------------------------------------------
    def _act(self, obs, deterministic=False):
        """policy.forward (policies.py:195-226): action, value, log_prob"""
        mean, value = self.policy.forward({k: obs[k] for k in self.obs_keys}, save_activations=False)
        M = mean.shape[0]
        action = th.empty((M, 4), device=self.device)
        logp = th.empty(M, device=self.device)
        self._sample_step += 1
        _lib.check(_lib.lib().vf_head_sample(_ptr(mean), _ptr(self.policy.log_std), _ptr(action), _ptr(logp), M,
                                             self._noise_key, self._sample_step, 1 if deterministic else 0,
                                             self._stream()))
        return action, value.view(M), logp

    def predict_values(self, obs):
        _, value = self.policy.forward({k: obs[k] for k in self.obs_keys}, save_activations=False)
        return value.view(-1).clone()

    def predict(self, obs, state=None, episode_start=None, deterministic: bool = False):
        """SB3 ``BaseAlgorithm.predict`` as the evaluation harness calls it (utils/evaluate.py:94): -> (action, None);
        deterministic: a = tanh(mean)"""
        return self._act(obs, deterministic)[0], None

    def collect_rollouts(self):
        """SB3 OnPolicyAlgorithm.collect_rollouts: n_steps of policy -> env.step -> buffer, with the
        TimeLimit bootstrap reward += gamma * V(terminal_obs) for truncated episodes, then GAE."""
        env, buf = self.env, self.buf
        if self._last_obs is None:
            self._last_obs = env.reset()
            self._last_starts = th.ones(self.n_envs, device=self.device)
        obs = self._last_obs
        L, pol, N = _lib.lib(), self.policy, self.n_envs
        buf.episode_starts[0].copy_(self._last_starts)
        bs = self._bootstrap_list() if self.defer_bootstrap else None
        fused = False
        if self.fused_rollout and self.defer_bootstrap and hasattr(env, "collect_policy"):
            # the whole loop below as one persistent launch (vf_ppo_rollout); same buffer rows, same Philox counters
            for k in self.obs_keys:
                if k == "state":
                    buf.obs[k][0].copy_(obs[k])
                else:
                    buf.obs[k].copy_(obs[k].unsqueeze(0).expand_as(buf.obs[k]))        # constant per env ("target")
            fused = env.collect_policy(pol, self.obs_keys, buf, bs, self._noise_key, self._sample_step, self._last_starts)
            if fused is not False:
                obs = fused
                self._sample_step += self.n_steps
            else:
                self.fused_rollout = False
        for t in range(self.n_steps if fused is False else 0):
            # policy.forward (policies.py:195-226) straight into row t of the buffer: value head, sampled action, log-prob
            action, logp = buf.actions[t], buf.log_probs[t]
            mean, _ = pol.forward({k: obs[k] for k in self.obs_keys}, save_activations=False, out_value=buf.values[t])
            self._sample_step += 1
            _lib.check(L.vf_head_sample(_ptr(mean), _ptr(pol.log_std), _ptr(action), _ptr(logp), N, self._noise_key,
                                        self._sample_step, 0, self._stream()))
            for k in self.obs_keys:
                buf.obs[k][t].copy_(obs[k])
            obs, reward, done, _info = env.step(action)
            if not self.defer_bootstrap:
                _lib.check(L.vf_episode_stats(done.data_ptr(), _ptr(env._ep_return), _ptr(env._ep_length), _ptr(env._ep_flags),
                                              self._ep_stats.data_ptr(), N, self._stream()))
            # TimeLimit.truncated bootstrap: SB3 adds gamma * V(terminal_observation) where the info says truncated.  Deferred:
            # the few truncated rows of this step are appended to a compact list, valued once after the loop
            nxt = buf.episode_starts[t + 1] if t + 1 < self.n_steps else self._last_starts
            trows = env._terminal_state_rows()               # (N, w) in the env's observation map, w = the policy's row width
            assert trows.shape[1] == pol.obs_dims["state"], (trows.shape, pol.obs_dims)
            if self.defer_bootstrap:
                o1 = obs["target"] if "target" in self.obs_keys else None
                _lib.check(L.vf_rollout_post_collect(_ptr(reward), done.data_ptr(), _ptr(env._ep_flags), _ptr(buf.rewards[t]), _ptr(nxt),
                                                     _ptr(trows), _ptr(o1), trows.shape[1], 0 if o1 is None else o1.shape[1],
                                                     bs["cursor"].data_ptr(), bs["cap"], bs["idx"].data_ptr(), _ptr(bs["rows0"]),
                                                     _ptr(bs["rows1"]), t * N, N, _ptr(env._ep_return), env._ep_length.data_ptr(),
                                                     _ptr(bs["stat"]), self._stream()))
            else:
                tobs = {"state": trows.contiguous()}
                if "target" in self.obs_keys:
                    tobs["target"] = obs["target"]
                _, tv = pol.forward(tobs, save_activations=False)
                _lib.check(L.vf_rollout_post(_ptr(reward), done.data_ptr(), _ptr(env._ep_flags), _ptr(tv), float(self.gamma),
                                             _ptr(buf.rewards[t]), _ptr(nxt), N, self._stream()))
        if self.defer_bootstrap:
            self._ep_stats += bs["stat"].double().sum(dim=0)     # episode statistics of the rollout, per-agent sums folded once
            cnt = int(bs["cursor"].item())                       # one host sync per rollout
            if cnt > bs["cap"]:
                raise _lib.VisflyError(f"deferred bootstrap list overflow ({cnt} > {bs['cap']} truncated rows in one rollout)")
            CH = 8192                                            # fixed chunk: one set of forward buffers whatever the count
            for s0 in range(0, cnt, CH):
                m = min(CH, cnt - s0)
                rows = {"state": bs["rows0"][s0:s0 + CH]}
                if bs["rows1"] is not None:
                    rows["target"] = bs["rows1"][s0:s0 + CH]
                _, tv = pol.forward(rows, save_activations=False)
                _lib.check(L.vf_bootstrap_scatter(bs["idx"].data_ptr() + 4 * s0, _ptr(tv), m, float(self.gamma), _ptr(buf.rewards),
                                                  self._stream()))
        self._last_obs = obs
        last_values = self.predict_values(obs)
        _lib.check(_lib.lib().vf_gae(_ptr(buf.rewards), _ptr(buf.values), _ptr(buf.episode_starts), _ptr(last_values),
                                     _ptr(self._last_starts), _ptr(buf.advantages), _ptr(buf.returns), self.n_steps,
                                     self.n_envs, float(self.gamma), float(self.gae_lambda), self._stream()))
        self.num_timesteps += self.n_steps * self.n_envs * self.world

    # ------------------------------------------------------------------------------------------
    def _minibatch_update(self, mb, stats_acc=None):
        """one optimiser step on a minibatch {obs:*, actions, old_lp, adv, ret[, old_v]} of contiguous rows (PPO.py:203-292).
        Returns the loss statistics of the (global) minibatch, or None when target_kl stopped the update BEFORE the
        optimiser step (PPO.py:269-282)."""
        L, st, pol = _lib.lib(), self._stream(), self.policy
        rows = mb.get("rows")        # indexed minibatch: the fields are the whole rollout buffer, `rows` the minibatch's row indices
        obs = {k: mb["obs:" + k] for k in self.obs_keys}
        actions, old_lp, adv, ret = mb["actions"], mb["old_lp"], mb["adv"], mb["ret"]
        B = adv.numel()
        gB = B * self.world
        # the loss launch also writes d(loss)/d(log_std) into the tail of the flat gradient and adds the minibatch
        # statistics to the epoch accumulator (no separate copy / add launches)
        vclip = self.clip_range_vf is not None
        cfg = _lib.PpoLossCfg(self._now(self.clip_range), self.ent_coef, self.vf_coef, 1.0 / gB, _ptr(pol.grad, pol.log_std_off),
                              None if stats_acc is None else _ptr(stats_acc),
                              _ptr(mb["old_v"]) if vclip else None, self._now(self.clip_range_vf) if vclip else 0.0, 0)
        # reference-default policy shapes: forward + loss + reverse chain are one launch (vf_ppo_update)
        # single GPU: the weight-gradient fold also leaves the squared gradient norm as partial sums, which Adam adds up itself
        tail = None
        if self.fused_tail and self.world == 1 and self.target_kl is None:
            pmap, packed = pol.pack_map()
            tail = _lib.WgradTail(_ptr(pol.flat), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), pol.n_params, self._adam_cfg(self._opt_step + 1, pmap, packed, None),
                                  self._tail_sync.data_ptr())
        two = self.grad_buckets == 2 and tail is None and (self.world > 1 or os.environ.get("VISFLY_AMD_GRAD_BUCKETS_FORCE") == "1")
        between = None
        if two:
            if self._xstream is None:
                self._xstream, self._xev = th.cuda.Stream(device=self.device), [th.cuda.Event() for _ in range(3)]
            split, xs, ev = pol.bucket_split(), self._xstream, self._xev

            def between():        # trunks' gradients, log_std's and the statistics are final: their all-reduce runs under the second launch pair
                ev[0].record()
                xs.wait_event(ev[0])
                parallel.allreduce_sum_(self._gbuf[split:], stream=xs)
        res = pol.ppo_update(obs, actions, old_lp, adv, ret, cfg, self._stats, self._scratch, want_sumsq=self.world == 1 and not two,
                             row_index=rows, tail=tail, between=between)
        if two and res is True:
            ev[1].record()
            xs.wait_event(ev[1])
            parallel.allreduce_sum_(self._gbuf[:split], stream=xs)
            ev[2].record(xs)
            th.cuda.current_stream(self.device).wait_event(ev[2])
        elif two:       # no fused step for this network: one bucket after the layer-by-layer backward (below)
            two = False
        if res == "adam":        # the optimiser step happened inside the weight-gradient launch
            self._opt_step += 1
            self._tail_launches += 1
            pol.mark_updated(packed_current=pmap is not None)
            return self._stats
        sq = res if isinstance(res, tuple) else None
        if res is False and rows is not None:       # no fused step for this network: materialise the minibatch (what the gather did)
            obs = {k: v.index_select(0, rows) for k, v in obs.items()}
            actions, old_lp, ret = actions.index_select(0, rows), old_lp.index_select(0, rows), ret.index_select(0, rows)
            if vclip:
                old_v_sel = mb["old_v"].index_select(0, rows)       # bound to a name: it must outlive vf_ppo_loss (cfg holds its raw pointer)
                cfg.old_value = _ptr(old_v_sel)
            cfg.row_index, cfg.obs_copy0, cfg.obs_copy1 = None, None, None
        if res is False:
            mean, value = pol.forward(obs)
            d_mean, d_value = th.empty((B, 4), device=self.device), th.empty(B, device=self.device)
            _lib.check(L.vf_ppo_loss(_ptr(mean), _ptr(value), _ptr(pol.log_std), _ptr(actions), _ptr(old_lp), _ptr(adv), _ptr(ret),
                                     _ptr(d_mean), _ptr(d_value), _ptr(self._stats), B, C.byref(cfg), _ptr(self._scratch), st))
            pol.backward(d_mean, d_value, None)
        if self.world > 1 and not two:
            # ONE collective per optimiser step: the flat gradient and the 16 loss statistics are one buffer
            # (sum over ranks: every gradient term is already / global batch, the statistics are per-rank sums)
            parallel.allreduce_sum_(self._gbuf)
        if self.target_kl is not None:
            # PPO.py:269-282: approx_kl of THIS minibatch, checked before optimizer.step() (one host sync per minibatch,
            # only when target_kl is set; with several ranks the statistics came back with the gradient)
            kl = float(self._stats[3].item()) / gB
            if kl > 1.5 * self.target_kl:
                return None
        self._opt_step += 1
        if sq is None:
            _lib.check(L.vf_sumsq(_ptr(pol.grad), pol.n_params, _ptr(self._sumsq), _ptr(self._scratch), st))
        pmap, packed = pol.pack_map()          # Adam refreshes the packed MFMA weight images in the same launch
        acfg = self._adam_cfg(self._opt_step, pmap, packed, sq)
        _lib.check(L.vf_adam_step(_ptr(pol.flat), _ptr(pol.grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), pol.n_params,
                                  _ptr(self._sumsq), C.byref(acfg), st))
        pol.mark_updated(packed_current=pmap is not None)
        return self._stats

    def _adam_cfg(self, step, pmap, packed, sq):
        return _lib.AdamCfg(self.lr, self.betas[0], self.betas[1], self.adam_eps, self.weight_decay,
                            self.max_grad_norm if self.max_grad_norm is not None else 0.0, step, 0,
                            _ptr(pmap), _ptr(packed), None if sq is None else sq[0].data_ptr(), 0 if sq is None else sq[1],
                            self.policy.log_std_off)

    def _check_tail(self):
        """the fused optimiser tail's waves meet at device counters; one that waited past the time limit raised the abort word and the
        update did not happen -- a hard error, reported at the trainer's next host synchronisation"""
        if self._tail_launches and int(self._tail_sync[_lib.WGRAD_SYNC_ABORT].item()):
            self._tail_sync.zero_()
            raise _lib.VisflyError("the fused optimiser tail (vf_mlp_weight_grad_adam) timed out waiting for its waves to become "
                                   "co-resident: another kernel holds part of the device; set VISFLY_AMD_FUSED_TAIL=0")

    def train(self, permutations=None):
        """PPO.train (PPO.py:177-337): n_epochs passes over random minibatches (SB3 RolloutBuffer.get: the trailing
        partial minibatch of an epoch is trained on too).  `permutations`: one row permutation per epoch (rows = step * n_envs + env)
        instead of the trainer's own draws -- how tests replay the minibatches of a recorded run of the reference's train()
        (tests/test_ppo_loop_gpu.py)"""
        total = self.n_steps * self.n_envs
        bs = min(self.batch_size, total)
        g = th.Generator(device=self.device)
        g.manual_seed(self.seed + 7919 * (self._opt_step + 1))
        # loss statistics: one row per evaluated minibatch.  The reference appends every minibatch's MEAN to a list and logs np.mean of
        # the lists -- equal weight per minibatch, whatever its row count (the trailing partial one counts like a full one); approx_kl's
        # list is reset per epoch (PPO.py:197), so its log is the mean over the minibatches of the LAST epoch started, including the one
        # that tripped target_kl (PPO.py:263-282: appended before the check)
        n_mb = (total + bs - 1) // bs
        stats_mb = th.zeros((self.n_epochs * n_mb, 16), device=self.device)
        rows_mb, epoch_mb = [], []
        stop = False
        buf = self.buf
        flat = {"actions": buf.actions.view(-1, 4), "old_lp": buf.log_probs.view(-1), "adv": buf.advantages.view(-1),
                "ret": buf.returns.view(-1)}
        if self.clip_range_vf is not None:
            flat["old_v"] = buf.values.view(-1)
        flat.update({"obs:" + k: buf.obs[k].view(-1, buf.obs[k].shape[-1]) for k in self.obs_keys})
        for _epoch in range(self.n_epochs):
            if permutations is not None:
                perm = th.as_tensor(permutations[_epoch], dtype=th.int64, device=self.device).contiguous()
                assert perm.numel() == total
            else:
                perm = th.randperm(total, device=self.device, generator=g)
            # (preparing epoch e + 1's copy on a side stream under epoch e's optimiser steps gains nothing: the gather takes the HBM
            # bandwidth the weight-gradient kernel runs on -- profiles/r05_side_streams.txt, tools/exp_ppo_prefetch.py)
            # r05 (ABI 9), `index_minibatches`: the fused step can read its minibatch through the permutation slice
            # (vf_ppo_loss_cfg.row_index), like SB3's RolloutBuffer.get indexes the buffer -- only the advantages are gathered then.
            # Bit-identical, and SLOWER: 145.3 vs 142.5 ms per train() of 1280 steps (tools/exp_ppo_indexed.py) -- the 1.14 ms-per-epoch
            # streaming gather (5.7 ms) is replaced by 25 600 random 100-byte rows behind a dependent index load at the head of every
            # fused launch, +6.6 us of a lone wave's latency each (8.5 ms).  Off by default; the shuffled copy stays
            pol = self.policy
            indexed = (self.index_minibatches and pol._fused_ppo is not False and pol._plan is not None and pol.fused and pol.fused_backward)
            shuf = self._prepare_epoch({"adv": flat["adv"]} if indexed else flat, perm, bs)
            for s in range(0, total, bs):
                e = min(s + bs, total)
                if indexed:
                    mb = dict(flat, adv=shuf["adv"][s:e], rows=perm[s:e])
                else:
                    mb = {k: v[s:e] for k, v in shuf.items()}
                st = self._minibatch_update(mb, stats_mb[len(rows_mb)])
                rows_mb.append(e - s)
                epoch_mb.append(_epoch)
                if st is None:
                    stop = True
                    break
            self._n_updates += 1        # PPO.py:294: once per epoch started, the early-stopped one included
            if stop:
                break
        self._check_tail()
        if self.world > 1:
            parallel.allreduce_sum_(stats_mb)                        # log the global means, like a single-process run would
        n_eval = len(rows_mb)
        means = stats_mb[:max(n_eval, 1)].double().cpu().numpy() / (np.asarray(rows_mb or [1], np.float64)[:, None] * self.world)
        last = [i for i in range(n_eval) if epoch_mb[i] == epoch_mb[-1]] or [0]
        ep = parallel.allreduce_sum_(self._ep_stats.clone()).tolist()          # rollout statistics of this iteration (:398-414)
        self._ep_stats.zero_()
        if ep[0] > 0:
            self.logs.update({"rollout/ep_rew_mean": ep[1] / ep[0], "rollout/ep_len_mean": ep[2] / ep[0],
                              "rollout/ep_success_rate": ep[3] / ep[0], "rollout/episodes": ep[0]})
        # PPO.py:322-336
        ret, val = buf.returns.view(-1).double(), buf.values.view(-1).double()
        var_y = float(ret.var(unbiased=False))
        self.logs.update({"train/policy_gradient_loss": float(means[:, 0].mean()), "train/value_loss": float(means[:, 1].mean()),
                          "train/entropy_loss": float(means[:, 2].mean()), "train/approx_kl": float(means[last, 3].mean()),
                          "train/clip_fraction": float(means[:, 4].mean()),
                          "train/loss": float(means[-1, 0] + self.ent_coef * means[-1, 2] + self.vf_coef * means[-1, 1]),
                          "train/explained_variance": float("nan") if var_y == 0 else 1.0 - float((ret - val).var(unbiased=False)) / var_y,
                          "train/std": float(self.policy.log_std.exp().mean()),
                          "train/n_updates": self._n_updates, "train/optimiser_steps": self._opt_step, "train/learning_rate": self.lr,
                          "train/clip_range": self._now(self.clip_range), "train/early_stop": float(stop)})
        if self.clip_range_vf is not None:
            self.logs["train/clip_range_vf"] = self._now(self.clip_range_vf)

    def _prepare_epoch(self, flat, perm, bs, slot=0):
        """the epoch's shuffled copy of the rollout (buffer set `slot`), advantages normalised per minibatch"""
        total = perm.numel()
        n_seg, rem = divmod(total, bs)
        # one gather per epoch instead of one per minibatch: the shuffled copy makes every minibatch a
        # contiguous slice (same rows, same order as indexing the buffer with perm[s:s+bs])
        shuf = self._gather(flat, perm, slot)
        if self.normalize_advantage and bs * self.world > 1:
            # PPO.py:215-220 normalises per minibatch; all minibatches of the epoch in one launch (+ one
            # all-reduce of the per-minibatch sums when the minibatch spans several GPUs)
            scr = self._shuf.setdefault(("advn", slot), {})
            if scr.get("advn") is None or scr["advn"].shape != shuf["adv"].shape or scr["sums"].shape[0] != n_seg + 1:
                scr["advn"] = th.empty_like(shuf["adv"])
                scr["sums"] = th.empty((n_seg + 1, 2), dtype=th.float64, device=self.device)
            advn, sums = scr["advn"], scr["sums"]
            L, stv = _lib.lib(), self._stream()
            calls = [(_ptr(shuf["adv"]), _ptr(advn), n_seg, bs, bs * self.world, sums.data_ptr())]
            if rem > 1 or (rem == 1 and self.world > 1):       # PPO.py:216: only `if len(advantages) > 1`
                calls.append((_ptr(shuf["adv"], n_seg * bs), _ptr(advn, n_seg * bs), 1, rem, rem * self.world,
                              sums.data_ptr() + 16 * n_seg))
            elif rem == 1:
                advn[n_seg * bs:] = shuf["adv"][n_seg * bs:]
            if self.world > 1:
                for args in calls:
                    _lib.check(L.vf_adv_normalize_segments(*args, 0, stv))
                parallel.allreduce_sum_(sums)
                for args in calls:
                    _lib.check(L.vf_adv_normalize_segments(*args, 1, stv))
            else:
                for args in calls:
                    _lib.check(L.vf_adv_normalize_segments(*args, 2, stv))
            shuf["adv"] = advn
        return shuf

    def _gather(self, flat, perm, slot=0):
        """{name: rows of flat[name] in the order of perm} -- all fields in one launch (vf_gather_rows); the destination
        buffers (set `slot`) are kept between epochs"""
        if self._shuf is None:
            self._shuf = {}
        dst = self._shuf.get(slot)
        if dst is None or any(k not in dst or dst[k].shape != v.shape for k, v in flat.items()):
            dst = self._shuf[slot] = {k: th.empty_like(v) for k, v in flat.items()}
        gf = _lib.GatherFields()
        gf.n_fields = len(flat)
        for i, (k, v) in enumerate(flat.items()):
            gf.width[i], gf.src[i], gf.dst[i] = (v.shape[1] if v.dim() == 2 else 1), _ptr(v), _ptr(dst[k])
        _lib.check(_lib.lib().vf_gather_rows(C.byref(gf), perm.data_ptr(), perm.numel(), self._stream()))
        return dict(dst)

    def learn(self, total_timesteps: int, log_interval: Optional[int] = None):
        """PPO.learn (PPO.py:116-175): alternate rollout collection and training"""
        t0 = time.time()
        start = self.num_timesteps
        it = 0
        while self.num_timesteps - start < total_timesteps:
            self.collect_rollouts()
            # SB3 _update_current_progress_remaining (PPO.py:150-152): after the rollout, before train()
            self._current_progress_remaining = 1.0 - float(self.num_timesteps - start) / float(total_timesteps)
            self.train()
            it += 1
            if log_interval and it % log_interval == 0:
                th.cuda.synchronize(self.device)
                fps = (self.num_timesteps - start) / max(time.time() - t0, 1e-9)       # PPO._dump_logs :394-395
                self.logs["time/fps"] = fps
                print(f"[ppo] it {it} steps {self.num_timesteps} fps {fps:.3e} "
                      f"pg {self.logs['train/policy_gradient_loss']:.4f} v {self.logs['train/value_loss']:.4f}")
        th.cuda.synchronize(self.device)
        self.logs["time/fps"] = (self.num_timesteps - start) / max(time.time() - t0, 1e-9)
        return self
