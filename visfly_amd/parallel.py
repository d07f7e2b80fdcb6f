"""One process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on
CPU for the world_size-2 tests).  Agents are independent, so the data path shards them
contiguously by rank and exchanges nothing; the only collectives are the PPO gradient all-reduce
(one flat fp32 buffer per optimiser step) and timing/statistics reductions."""
import os
from typing import Tuple

import torch as th
import torch.distributed as dist


def init(backend: str = None) -> Tuple[int, int, int]:
    """-> (rank, world_size, local_rank); reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if th.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            th.cuda.set_device(local)
            kw["device_id"] = th.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def broadcast_(t: th.Tensor, src: int = 0) -> th.Tensor:
    """rank `src`'s tensor to every rank (initial parameters: only gradients are exchanged afterwards)"""
    if world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def shard(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous agent shard of rank: (first agent id, count); remainders go to the low ranks"""
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


_native = {"comm": None, "tried": False}


def _rccl_path() -> str:
    """the librccl.so this process already maps (PyTorch ships its own copy), else PyTorch's, else the ROCm one"""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl.so" in line:
                    return line.split()[-1]
    except OSError:
        pass
    cand = os.path.join(os.path.dirname(th.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else "/opt/rocm/lib/librccl.so"


def native_comm(force: bool = False):
    """vf_comm handle for vf_allreduce_grads (RCCL straight from C on the caller's stream), created once per process.
    torch.distributed is the bootstrap: rank 0's ncclUniqueId is broadcast through the existing process group.  Only for
    the nccl backend (the gloo test configurations keep torch.distributed); VISFLY_AMD_NATIVE_ALLREDUCE=0 disables it.
    If RCCL refuses the communicator the reason is printed once and torch.distributed's RCCL all-reduce is used."""
    if _native["tried"] and not force:
        return _native["comm"]
    _native["tried"] = True
    if os.environ.get("VISFLY_AMD_NATIVE_ALLREDUCE", "1") == "0":
        return None
    w = world_size()
    if not force and (w == 1 or dist.get_backend() != "nccl"):
        return None
    import ctypes as C
    import warnings
    from . import _lib
    L = _lib.lib()

    # the agreement rounds run over a gloo (CPU) side group: after a timed-out ncclCommInitRank the NCCL backend's own stream may be
    # blocked behind the stuck init on this device, and an all-reduce on it would hang the very ranks the watchdog is protecting
    # (ADVICE r03).  new_group is itself collective: every rank reaches this line (local failures above are caught, not raised)
    # The side group is made once per process (a forced retry reuses it) and used only if EVERY rank got one: new_group can fail on
    # some ranks only, and agreement rounds on different groups would hang -- that availability vote is the one
    # all-reduce that has to run on the default group, before anything was handed to RCCL
    side = None
    if w > 1:
        if "side" not in _native:
            try:
                _native["side"] = dist.new_group(backend="gloo")
            except Exception:  # noqa: BLE001   (gloo not built)
                _native["side"] = None
            got = th.tensor([1 if _native["side"] is not None else 0], dtype=th.int32)
            if dist.get_backend() == "nccl":
                got = got.to(th.device("cuda", th.cuda.current_device()))
            dist.all_reduce(got, op=dist.ReduceOp.MIN)
            if not bool(got.item()):
                _native["side"] = None
        side = _native["side"]

    def agreed(ok: bool) -> bool:
        """every rank must take the same path (one rank on torch.distributed and the others on the native communicator would
        hang in their first all-reduce): logical AND over the ranks through the bootstrap process group"""
        if w == 1:
            return ok
        if side is not None:
            flag = th.tensor([1 if ok else 0], dtype=th.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=side)
        else:
            flag = th.tensor([1 if ok else 0], dtype=th.int32, device=th.device("cuda", th.cuda.current_device()))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    why, h = None, None
    ident = (C.c_uint8 * 128)()
    try:                                    # phase 1, local: find RCCL, rank 0 draws the id
        _lib.check(L.vf_comm_library(_rccl_path().encode()))
        if rank() == 0:
            _lib.check(L.vf_comm_unique_id(ident))
    except Exception as e:  # noqa: BLE001
        why = e
    if agreed(why is None):
        try:                                # phase 2, collective: the id goes to everybody ...
            if w > 1:
                box = [bytes(ident)]
                dist.broadcast_object_list(box, src=0)
                ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
                if not any(ident):
                    raise RuntimeError("rank 0 sent an empty ncclUniqueId")
        except Exception as e:  # noqa: BLE001
            why = e
        # ... and only when EVERY rank holds it does anybody enter ncclCommInitRank (a rank that raised above would leave the
        # others blocked inside the init for ever); the init itself runs under a watchdog for the same reason
        if agreed(why is None):
            import threading
            box2, box_lock = {}, threading.Lock()      # "abandoned" is checked and "h" stored under the lock: a late init either
            dev_index = th.cuda.current_device()       # sees the flag and destroys its handle, or its handle is seen below

            def init():
                try:
                    th.cuda.set_device(dev_index)
                    hh = _lib._vp()
                    _lib.check(L.vf_comm_init(ident, w, rank(), C.byref(hh)))
                    with box_lock:
                        late = box2.get("abandoned", False)
                        if not late:
                            box2["h"] = hh
                    if late:                         # the watchdog gave up on this thread: nobody will use the communicator it got
                        L.vf_comm_destroy(hh)
                except Exception as e:  # noqa: BLE001
                    box2["why"] = e

            t = threading.Thread(target=init, daemon=True)
            t.start()
            t.join(float(os.environ.get("VISFLY_AMD_COMM_INIT_TIMEOUT", "180")))
            with box_lock:
                h = box2.get("h")
                if h is None and "why" not in box2:
                    box2["abandoned"] = True         # if the init ever returns, its thread destroys the handle itself
            if h is None:
                why = box2.get("why") or RuntimeError("ncclCommInitRank did not return (another rank never entered it?)")
            if not agreed(h is not None):
                # a peer is still inside (or never entered) ncclCommInitRank: destroying OUR communicator now would wait for it.
                # The handle is abandoned instead (a few MB until the process exits; no atexit destroy is registered for it)
                h, why = None, why or RuntimeError("another rank could not create its communicator")
        else:
            why = why or RuntimeError("another rank did not receive the communicator id")
    else:
        why = why or RuntimeError("another rank could not load RCCL")
    _native["why"] = None if h is not None else f"{type(why).__name__}: {why}" if why is not None else "unknown"
    if h is None:
        warnings.warn(f"visfly_amd: native RCCL communicator unavailable ({why}); gradient all-reduce goes through "
                      "torch.distributed (same RCCL collective, Python dispatch)")
    _native["comm"] = h
    if h is not None and not _native.get("atexit"):
        import atexit
        _native["atexit"] = True
        atexit.register(_destroy_native)
    return _native["comm"]


def native_comm_error():
    """why the last native_comm() attempt fell back to torch.distributed (None: it did not, or it was never tried / not applicable)"""
    return _native.get("why")


def _destroy_native():
    h, _native["comm"] = _native["comm"], None
    if h is not None:
        try:
            from . import _lib
            _lib.lib().vf_comm_destroy(h)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def allreduce_sum_(t: th.Tensor, stream: th.cuda.Stream = None) -> th.Tensor:
    """in-place sum over the ranks.  fp32 / fp64 device tensors on the nccl backend: one ncclAllReduce enqueued from C on
    torch's current stream (vf_allreduce_grads); anything else: torch.distributed.  ``stream``: enqueue on that stream instead
    (the two-bucket gradient exchange: the caller orders it against its own stream with events)"""
    if world_size() > 1:
        c = native_comm() if (t.is_cuda and t.is_contiguous() and t.dtype in (th.float32, th.float64)) else None
        if c is not None:
            from . import _lib
            L = _lib.lib()
            fn = L.vf_allreduce_grads if t.dtype == th.float32 else L.vf_allreduce_f64
            with th.cuda.device(t.device):
                _lib.check(fn(c, t.data_ptr(), t.numel(), stream.cuda_stream if stream is not None else _lib.current_stream(t.device)))
        elif stream is not None:
            with th.cuda.stream(stream):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def max_over_ranks(x: float, device="cpu") -> float:
    if world_size() == 1:
        return float(x)
    t = th.tensor([x], dtype=th.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if world_size() > 1:
        dist.barrier()
