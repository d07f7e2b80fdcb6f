#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the fused quadrotor step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one control interval (ctrl_dt = 8 sub-steps of dt) for every agent of the batch.
Workload = BASELINE.json configs[1]: 65 536 agents per GPU, visual=False, bodyrate + Euler,
dt=0.0025 / ctrl_dt=0.02, ctrl_delay on, 3-slot comm-delay ring.  Agents shard trivially:
each rank owns its own 65 536 agents (weak scaling), no data-path collective.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory (visfly_amd/__init__.py); before torch
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver

import numpy as np  # noqa: E402
import torch  # noqa: E402

AGENTS_PER_GPU = 65536
DYN_KW = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
BYTES_PER_AGENT_STEP = 244      # 116 B read + 128 B written per agent-step (Dynamics.step)
BYTES_PER_ENV_STEP = 350        # full env.step adds obs, reward, counters and flags
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def host_cores():
    """cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU box advertises 256 CPUs
    but may schedule far fewer; 256 OpenMP threads on a handful of cores only measure context switching)"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(np.ceil(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(np.ceil(q / per))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline_env(consts, kind, N, seconds_target=5.0, threads=None, note=None):
    """TEST/BENCH INFRASTRUCTURE: time the CPU oracle (C restatement of the reference's env.step: dynamics interval + bbox
    collision + reward / counters / done masks) on the host cores of this box on a bounded sample of the same workload.
    The sample runs inside ONE OpenMP region per block of 64 steps (oracle.OracleEnv.run_steps: every thread walks its own
    contiguous agent chunk through the steps -- agents are independent), so 256 threads are not fork-join-bound; no
    auto-reset inside the sample: episodes are 256 steps long and the sample is re-spawned every 192 steps."""
    import oracle
    cores = host_cores()
    threads = min(threads or cores, cores)
    oracle.set_threads(threads)
    gates = [[4, 4, 1.], [8, 0, 2.], [5, -4, 1.], [1, -1, 1.]] if kind == "racing" else None
    target = (1.0, 0.0, 1.5) if kind == "hover" else (9.0, 0.0, 1.0)
    env = oracle.OracleEnv(consts, N, kind, 256, target=target, success_radius=0.3 if kind == "racing" else 0.5, gates=gates)
    rng = np.random.default_rng(0)
    fs = np.zeros((N, 22), np.float32)
    fs[:, 0:3] = (np.array([1, 0, 1.5]) + rng.uniform(-1, 1, (N, 3)) * np.array([1, 1, .5])).astype(np.float32)
    fs[:, 3] = 1.0
    fs[:, 13:17], fs[:, 17:21] = consts["w_init"], consts["T_init"]
    hover = np.array([-1 / 3, 0, 0, 0]) if int(consts["action_type"]) == 1 else np.zeros(4)
    a = np.clip(hover + rng.uniform(-.02, .02, (16, N, 4)), -1, 1).astype(np.float32)
    env.reset_full_state(fs)
    env.run_steps(a, 2)
    block = 64 if threads > 1 else 8
    t0 = time.perf_counter()
    steps = 0
    while True:
        if steps % 192 == 0:
            env.reset_full_state(fs)
        env.run_steps(a, block)
        steps += block
        el = time.perf_counter() - t0
        if el > seconds_target or steps >= 16384:
            break
    assert np.isfinite(env.dyn.S).all()
    oracle.set_threads(cores)
    out = {"value": N * steps / el, "affinity_cpus": len(os.sched_getaffinity(0)), "unit": "agent-steps/s", "cores": threads, "kind": "port",
           "sample": f"oracle/vf_oracle.c {kind} env.step restatement (OpenMP, {threads} thread{'s' if threads > 1 else ''}, one "
                     f"parallel region per {block} steps), N={N}, {steps} control steps, {el:.1f} s"}
    if note:
        out["note"] = note
    return out


# The reference itself (PyTorch CPU) cannot travel to the GPU box; its own numbers were recorded once in the build container
# (SURVEY.md section 6, 8-core Xeon 2.1 GHz, torch 2.10 CPU, 8 threads) and ride along for orientation (last row)
REFERENCE_RECORDED = {"where": "build container, 8-core Xeon 2.1 GHz, torch 2.10 CPU, 8 threads (SURVEY.md section 6); not measured on this box",
                      "Dynamics.step_agent_steps_per_s@N=65536": 1.21e6, "HoverEnv.step_agent_steps_per_s@N=65536": 2.92e5}


def exchange_block(gbuf, n_upd, s_per_iteration, us_per_update, world, dev):
    """the one exchange step of the path, timed on its own: 50 all-reduces of the trainer's gradient + statistics buffer on this stream
    (vf_allreduce_grads = ncclAllReduce from C on the nccl backend), per rank and as a share of the optimiser step / iteration.
    `native_rccl` false = the native communicator could not be created and the all-reduce went through torch.distributed (same RCCL
    collective, Python dispatch): visible in the JSON line, with the reason, not only as a warning."""
    from visfly_amd import parallel
    if world <= 1:
        return None
    import torch.distributed as dist
    ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    scratch = torch.zeros_like(gbuf)
    parallel.allreduce_sum_(scratch)
    torch.cuda.synchronize()
    parallel.barrier()
    ev2[0].record()
    for _ in range(50):
        parallel.allreduce_sum_(scratch)
    ev2[1].record()
    torch.cuda.synchronize()
    mine = ev2[0].elapsed_time(ev2[1]) * 1e3 / 50
    per_rank = [None] * world
    dist.all_gather_object(per_rank, float(mine))
    ar_us = max(per_rank)
    native = parallel.native_comm() is not None
    flags = [None] * world
    dist.all_gather_object(flags, bool(native))
    out = {"allreduce_us": ar_us, "allreduce_us_per_rank": per_rank, "floats": int(scratch.numel()), "bytes": int(scratch.numel() * scratch.element_size()),
           "per_iteration_ms": ar_us * n_upd * 1e-3, "share_of_iteration": ar_us * n_upd * 1e-6 / s_per_iteration,
           "share_of_optimiser_step": (ar_us / us_per_update) if us_per_update else None,
           "native_rccl": all(flags), "native_rccl_per_rank": flags, "backend": dist.get_backend()}
    if not all(flags):
        out["native_rccl_fallback_reason"] = parallel.native_comm_error() or "another rank could not create its communicator"
    return out


def cpu_baseline(consts, seconds_target=5.0, threads=None):
    out = cpu_baseline_env(consts, "hover", AGENTS_PER_GPU, seconds_target, threads)
    out["reference_recorded"] = REFERENCE_RECORDED
    return out


def bench_ppo(args, rank, world, dev, iters=None, cpu_ref=False):
    """BASELINE configs[3] shape: NavigationEnv + PPO full train loop (rollout + GAE + 5 epochs of minibatch updates),
    32 768 agents per GPU (262 144 / 8), agents sharded by rank, one RCCL all-reduce of the flat gradient per
    optimiser step.  Returns the result dict (rank 0 prints it for --workload ppo; main() embeds it as "secondary")."""
    from visfly_amd import parallel
    from visfly_amd.envs import NavigationEnv
    from visfly_amd.ppo import PPO
    N = args.agents if args.agents != AGENTS_PER_GPU else 32768
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
    env = NavigationEnv(num_agent_per_scene=N, seed=42 + rank, dynamics_kwargs=dict(DYN_KW), random_kwargs=spawn,
                        device=dev, max_episode_steps=256)
    # the networks of the reference's experiment YAMLs (exps/examples/alg_cfgs/*/PPO.yaml: activation_fn relu; the policy class's own
    # default would be Tanh trunks, policies.py:108)
    kw = dict(policy_kwargs=dict(activation_fn=getattr(args, "activation", None) or "relu"))
    if getattr(args, "net_arch", None):
        arch = {k: [int(x) for x in v.split(",")] for k, v in (p.split("=") for p in args.net_arch.split(":"))}
        kw = dict(policy_kwargs=dict(features_extractor_class="StateTargetExtractor", activation_fn=getattr(args, "activation", None) or "ReLU", net_arch=arch,
                                     features_extractor_kwargs=dict(net_arch=dict(state=dict(layer=[128, 64]), target=dict(layer=[128, 64])))))
    ppo = PPO(env, n_steps=256, batch_size=25600, n_epochs=5, learning_rate=1e-4, seed=0, **kw)
    ppo.learn(256 * N * world)   # warm-up iteration
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    iters = iters or max(1, args.steps // 256)
    ppo.learn(256 * N * world * iters)
    torch.cuda.synchronize()
    parallel.barrier()
    el = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    # split of one iteration and the MFMA roofline of the optimiser steps (outside the timed region): one more rollout and
    # one more train() pass, each timed with device events; algorithmic flops = 2 x weights x rows for each of forward,
    # data gradient and weight gradient
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n0 = ppo._opt_step
    # the 256 policy -> sample -> env.step -> buffer rounds on their own clock (one persistent launch, vf_ppo_rollout, where the
    # library has the kernel): events around env.collect_policy
    roll_ev, inner = [torch.cuda.Event(enable_timing=True) for _ in range(2)], getattr(env, "collect_policy", None)
    roll = {"fused": False}

    def timed_collect(*a, **k):
        roll_ev[0].record()
        r = inner(*a, **k)
        roll_ev[1].record()
        roll["fused"] = r is not False
        return r

    if inner is not None:
        env.collect_policy = timed_collect
    ev[0].record()
    ppo.collect_rollouts()
    ev[1].record()
    if inner is not None:
        del env.collect_policy
    ppo.train()
    ev[2].record()
    torch.cuda.synchronize()
    n_upd = max(1, ppo._opt_step - n0)
    us_upd = ev[1].elapsed_time(ev[2]) * 1e3 / n_upd
    rows = min(ppo.batch_size, 256 * N)
    flops = 6.0 * ppo.policy.log_std_off * rows          # log_std_off = number of weights + biases of the network
    tfs = flops / (us_upd * 1e-6) / 1e12
    roof = {"bound": "mfma", "achieved": tfs, "peak": 157.3, "unit": "TFLOP/s", "frac": tfs / 157.3, "traffic": None,
            "kernel": ("optimiser step: k_ppo_update_chain<generated class> + k_mlp_wgrad + fold + Adam" if kw else
                       "optimiser step: k_ppo_update_split (k_ppo_update_chain above 32 768 rows) + k_mlp_wgrad + fold + Adam"), "us_per_update": us_upd,
            "rows": rows, "note": "fp32 v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)"}
    exchange = exchange_block(ppo._gbuf, n_upd, el / iters, us_upd, world, dev)
    # the two gradient-exchange modes side by side: one more train() pass in each, device events; N = 1 with
    # VISFLY_AMD_GRAD_BUCKETS_FORCE=1 shows what the split of the weight-gradient launch costs by itself
    if exchange is not None or os.environ.get("VISFLY_AMD_GRAD_BUCKETS_FORCE") == "1":
        modes = {}
        for nb in (1, 2, 1, 2):
            ppo.grad_buckets = nb
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k0 = ppo._opt_step
            parallel.barrier()
            e0.record()
            ppo.train()
            e1.record()
            torch.cuda.synchronize()
            modes.setdefault(f"buckets_{nb}", []).append(e0.elapsed_time(e1) * 1e3 / max(1, ppo._opt_step - k0))
        ab = {k: parallel.max_over_ranks(min(v), dev) for k, v in modes.items()}
        ab["note"] = ("us per optimiser step, best of two train() passes each, max over ranks; buckets_2 = trunks' bucket all-reduced on a second "
                      "stream under the extractors' weight-gradient launches (VISFLY_AMD_GRAD_BUCKETS=2)")
        exchange = dict(exchange or {}, grad_bucket_modes=ab)
        ppo.grad_buckets = int(os.environ.get("VISFLY_AMD_GRAD_BUCKETS", "1"))
    out = {"metric": "PPO env-steps/s (rollout + train, NavigationEnv, StateTarget MLP)",
           "value": 256 * N * world * iters / el, "unit": "agent-steps/s", "n_gpus": world,
           "iterations": iters, "s_per_iteration": el / iters, "dtype": "f32", "data": "synthetic",
           "split_ms": {"collect_rollouts": ev[0].elapsed_time(ev[1]), "train": ev[1].elapsed_time(ev[2]),
                        "rollout_256_steps": roll_ev[0].elapsed_time(roll_ev[1]) if roll["fused"] else None,
                        "rollout_persistent_launch": roll["fused"],
                        "optimiser_steps": n_upd, "note": "rank 0's device clock: env + policy rollout (no collective) vs "
                                                            "the optimiser steps (one all-reduce each)"},
           "exchange": exchange,
           "config": {"workload": f"NavigationEnv {N} agents/GPU, n_steps=256, batch 25600/GPU, 5 epochs "
                                  "(BASELINE configs[3] shard)" + (f", net_arch {args.net_arch} (generated chain class)" if getattr(args, "net_arch", None) else "")
                                  + (f", activation_fn {args.activation}" if getattr(args, "activation", None) else ""),
                      "logs": {k: float(v) for k, v in ppo.logs.items()}},
           "roofline": roof}
    if cpu_ref and rank == 0:
        out["cpu_baseline"] = cpu_baseline_env(env.envs.dynamics.constants, "nav", 32768, seconds_target=3.0,
                                               note="env.step part of the loop only (the reference's PPO runs its MLP on "
                                                    "torch CPU; not ported to C)")
    env.close()
    return out


def _actor_critic_arch(args):
    """--net-arch pi=128,128[:qf=128,128] for the BPTT / SHAC legs: policy_kwargs of the reference's MTDPolicy over its StateExtractor with a
    net_arch the library holds no built-in chain class of -- actor (and twin critic) classes are generated, the horizons run from the BPTT
    plugin (visfly_amd/_jit.py; profiles/r06_generated_horizons.txt)"""
    if not getattr(args, "net_arch", None):
        return None
    arch = {k: [int(x) for x in v.split(",")] for k, v in (p.split("=") for p in args.net_arch.split(":"))}
    arch.setdefault("qf", list(arch["pi"]))
    return dict(features_extractor_class="StateExtractor", features_extractor_kwargs=dict(net_arch=dict(state=dict(layer=[128, 64]))),
                net_arch=dict(pi=arch["pi"], qf=arch["qf"]), activation_fn="relu", share_features_extractor=False)


def bench_bptt(args, rank, world, dev, iters=None, cpu_ref=False):
    """BASELINE configs[4] shape: RacingEnv, thrust actions, BPTT H=64 through the adjoint kernel, 16 384 agents per GPU
    (131 072 / 8), agents sharded by rank, one all-reduce of the flat gradient per update.  The leg's `value` is the loop with the
    actor the reference's own BPTT.learn runs (utils/algorithms/BPTT.py:113: MTDPolicy.actor = td_policies.Actor, two trunks and a
    state-dependent clamped log_std head: `policy="MultiInputPolicy"`); the one-trunk MlpPolicy actor (a log_std parameter, half the
    policy MFMA work per agent-step) is reported next to it as `mlp_policy_actor`."""
    from visfly_amd import parallel
    from visfly_amd.bptt import BPTT
    from visfly_amd.envs import RacingEnv
    N = args.agents if args.agents != AGENTS_PER_GPU else 16384
    dkw = dict(DYN_KW, action_type="thrust")
    iters = iters or max(1, args.steps // 64)

    def run(policy):
        env = RacingEnv(num_agent_per_scene=N, seed=42 + rank, dynamics_kwargs=dkw, device=dev, max_episode_steps=256,
                        requires_grad=True, tensor_output=True)
        pk = _actor_critic_arch(args) if policy else None
        algo = BPTT(env, horizon=64, gamma=0.99, learning_rate=1e-3, seed=0, **({"policy": policy} if policy else {}),
                    **({"policy_kwargs": pk} if pk else {}))
        algo.learn(64 * N * world)     # warm-up update
        torch.cuda.synchronize()
        parallel.barrier()
        # short regions (a few updates of 4 ms) are at the mercy of one host hiccup: three of them, the median
        els = []
        for _ in range(3 if iters < 16 else 1):
            t0 = time.perf_counter()
            algo.learn(64 * N * world * iters)
            torch.cuda.synchronize()
            parallel.barrier()
            els.append(parallel.max_over_ranks(time.perf_counter() - t0, dev))
        return env, algo, sorted(els)[len(els) // 2], len(els)

    env, algo, el, regions = run("MultiInputPolicy")
    # rooflines of the loop as a whole (it is a chain of persistent launches at 1024 waves whose waves alternate between a 16-row MFMA
    # chain and the env step's VALU stream: both fractions are small by construction and are reported so that they can be tracked).
    # MFMA: actor forward + data gradient + weight gradient = 6 flops per weight per agent-step.  HBM: algorithmic bytes per
    # agent-step = the forward env step (350 B) + the state checkpoint written to the tape and read back by the adjoint
    # (2 slabs) + the adjoint slab in / out (2 slabs)
    w_a = int(algo.policy.n_params)
    steps_total = 64 * N * iters                      # this rank
    tfs = 6.0 * w_a * steps_total / el / 1e12
    slab_bytes = env._slab.numel() * 4 / N
    hbm_bytes = BYTES_PER_ENV_STEP + 4 * slab_bytes
    gbs = hbm_bytes * steps_total / el / 1e9
    roof = {"bound": "mfma", "achieved": tfs, "peak": 157.3, "unit": "TFLOP/s", "frac": tfs / 157.3, "traffic": None,
            "kernel": "whole update: k_bptt_rollout (actor chain + env step, 64 x) + k_bptt_reverse (adjoint env step + reverse chain, 64 x) "
                      "+ one weight-gradient launch per horizon",
            "flops_per_agent_step": 6.0 * w_a, "actor_params": w_a,
            "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bytes_per_agent_step": hbm_bytes,
                    "note": "env step 350 B + 4 slabs (tape write / read, adjoint in / out)"},
            "note": "loop-level figures over the timed updates (wall clock, not one kernel) -- profiles/r04_bptt_kernel_stats.txt lists "
                    "the per-kernel times, profiles/r04_pmc_mfma.txt the MFMA-busy share of a wave's life"}
    out = {"metric": "BPTT env-steps/s (H=64 rollout + adjoint + actor update, RacingEnv, the reference's td_policies.Actor)",
           "value": 64 * N * world * iters / el, "unit": "agent-steps/s", "n_gpus": world,
           "iterations": iters, "regions": regions, "s_per_iteration": el / iters, "dtype": "f32", "data": "synthetic", "roofline": roof,
           "policy": "td_policies.Actor (policy=\"MultiInputPolicy\"): extractor [128, 64], latent_pi / log_latent_pi [64, 64], mu / log_std heads",
           "exchange": exchange_block(algo.policy.grad, 1, el / iters, el / iters * 1e6, world, dev),     # one all-reduce of the flat gradient per update
           "config": {"workload": f"RacingEnv {N} agents/GPU, thrust actions, horizon 64 (BASELINE configs[4] shard)" +
                                  (f", net_arch {args.net_arch} (generated actor class, BPTT plugin)" if getattr(args, "net_arch", None) else ""),
                      "logs": {k: float(v) for k, v in algo.logs.items()}}}
    if cpu_ref and rank == 0:
        out["cpu_baseline"] = cpu_baseline_env(env.envs.dynamics.constants, "racing", 16384, seconds_target=3.0,
                                               note="forward env.step only (no adjoint on the CPU side)")
    env.close()
    env, algo, el2, _ = run(None)
    pol = algo.policy
    w_pi = sum(ly.K * ly.No + ly.No for ly in pol.layers if not (ly.dst == "value" or ly.dst.startswith("vf:")))
    out["mlp_policy_actor"] = {"value": 64 * N * world * iters / el2, "unit": "agent-steps/s", "s_per_iteration": el2 / iters,
                               "policy": "MlpPolicy actor (one trunk, log_std parameter): extractor [128, 64], pi [64, 64], 4-wide head",
                               "mfma_frac": 6.0 * w_pi * 64 * N * iters / el2 / 1e12 / 157.3, "actor_params": int(w_pi)}
    env.close()
    return out


def bench_shac(args, rank, world, dev, iters=None, cpu_ref=False):
    """SHAC (utils/algorithms/shac.py:215-278) in the reference's network shapes -- HoverEnv, bodyrate, 16 384 agents per GPU,
    horizon 32, 5 critic steps per iteration (the reference's defaults), agents sharded by rank, one all-reduce of the actor gradient
    and one of the critic gradient per critic step."""
    from visfly_amd import parallel
    from visfly_amd.envs import HoverEnv
    from visfly_amd.shac import SHAC
    N, H = (args.agents if args.agents != AGENTS_PER_GPU else 16384), 32
    env = HoverEnv(num_agent_per_scene=N, seed=42 + rank, dynamics_kwargs=dict(DYN_KW), device=dev, max_episode_steps=256,
                   requires_grad=True, tensor_output=True)
    pk = _actor_critic_arch(args)
    algo = SHAC(env, horizon=H, gamma=0.99, learning_rate=1e-3, gradient_steps=5, seed=0, **({"policy": "MultiInputPolicy", "policy_kwargs": pk} if pk else {}))
    algo.learn(H * N * world)       # warm-up iteration
    torch.cuda.synchronize()
    parallel.barrier()
    iters = iters or max(1, args.steps // H)
    els = []
    for _ in range(3 if iters < 16 else 1):
        t0 = time.perf_counter()
        algo.learn(H * N * world * iters)
        torch.cuda.synchronize()
        parallel.barrier()
        els.append(parallel.max_over_ranks(time.perf_counter() - t0, dev))
    el = sorted(els)[len(els) // 2]
    # MFMA work of one iteration per (step, agent) row: actor forward + data gradient + weight gradient (6 flops per weight), the next
    # action's actor forward (2), the target critics' forward (2), and gradient_steps critic updates over the horizon buffer (6 each)
    wa = algo.policy.n_params
    wc = algo.critic.n_params
    flops_row = 6.0 * wa + 2.0 * wa + 2.0 * wc + algo.gradient_steps * 6.0 * wc
    tfs = flops_row * H * N * iters / el / 1e12
    out = {"metric": "SHAC env-steps/s (H=32 rollout + adjoint + actor update + 5 critic updates, HoverEnv)",
           "value": H * N * world * iters / el, "unit": "agent-steps/s", "n_gpus": world, "iterations": iters, "regions": len(els),
           "s_per_iteration": el / iters, "dtype": "f32", "data": "synthetic",
           "roofline": {"bound": "mfma", "achieved": tfs, "peak": 157.3, "unit": "TFLOP/s", "frac": tfs / 157.3, "traffic": None,
                        "kernel": "whole iteration: actor chain forward / reverse + env step + adjoint, target critics, 5 critic updates",
                        "flops_per_agent_step": flops_row, "actor_params": int(wa), "critic_params": int(wc),
                        "note": "loop-level figure over the timed iterations (wall clock, not one kernel)"},
           "config": {"workload": f"HoverEnv {N} agents/GPU, bodyrate, horizon {H}, 5 critic steps (SURVEY 8f-2; the reference's SHAC defaults)" +
                                  (f", net_arch {args.net_arch} (generated actor / twin-critic classes, BPTT plugin)" if getattr(args, "net_arch", None) else ""),
                      "logs": {k: float(v) for k, v in algo.logs.items()}}}
    env.close()
    return out


def bench_nav_rk4_dr(args, rank, world, dev, iters=None, cpu_ref=False):
    """BASELINE configs[2] (SURVEY 8d input 3): NavigationEnv.step, 65 536 agents per GPU, RK4 integrator + ctrl_delay + domain
    randomisation of the drag coefficients (drag_random = 0.1: two 16-byte per-agent granules read every step, re-drawn by the on-device
    spawner at every reset) -- the same fused launch as the headline, driven and timed the same way (env.step_n, regions of `steps`
    launches between barriers, HIP events on the launch stream for the kernel time)."""
    from visfly_amd import parallel
    from visfly_amd.envs import NavigationEnv
    N = args.agents
    K = iters or max(20, min(args.steps, 2000))
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
    dkw = dict(DYN_KW, integrator="rk4", drag_random=0.1)
    env = NavigationEnv(num_agent_per_scene=N, num_scene=1, seed=42 + rank, visual=False, dynamics_kwargs=dkw, random_kwargs=spawn,
                        device=dev, max_episode_steps=256, tensor_output=True, out_buffers=4)
    env.reset()
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    hover = torch.tensor([-1 / 3, 0, 0, 0], device=dev)
    Kc = min(K, args.chunk)
    seq = (hover + (torch.rand((Kc, N, 4), device=dev, generator=g) * 2 - 1) * 0.02).clamp(-1, 1).contiguous()

    def run_steps(n):
        done = 0
        while done < n:
            k = min(n - done, Kc)
            env.step_n(seq if k == Kc else seq[:k])
            done += k
    run_steps(max(256, 4 * Kc))          # allocation of the output ring + clocks
    torch.cuda.synchronize()
    walls, events = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        parallel.barrier()
        t0 = time.perf_counter()
        run_steps(K)
        torch.cuda.synchronize()
        walls.append(parallel.max_over_ranks(time.perf_counter() - t0, dev))
    for _ in range(3):
        parallel.barrier()
        e0.record()
        run_steps(K)
        e1.record()
        torch.cuda.synchronize()
        events.append(e0.elapsed_time(e1) * 1e-3)
    el, kern_us = statistics.median(walls), statistics.median(events) / K * 1e6
    dn = env._rollouts[Kc]["done"]
    bytes_step = BYTES_PER_ENV_STEP + 32             # + the two drag granules (16 B each) every step reads
    achieved = bytes_step * N / (kern_us * 1e-6) / 1e9
    out = {"metric": "agent-steps/sec (NavigationEnv.step, RK4 + ctrl_delay + drag domain randomisation)",
           "value": world * N * K / el, "unit": "agent-steps/s", "n_gpus": world, "steps": K, "us_per_step": el / K * 1e6,
           "dtype": "f32", "data": "synthetic", "episode_end_rate": float(dn.float().mean()),
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "kernel": "k_env_step<navigation,bodyrate,rk4,ctrl_delay> (per-agent drag granules)",
                        "kernel_us": kern_us, "bytes_per_agent_step": bytes_step,
                        "kernel_us_source": "HIP events on the launch stream over the timed regions / steps (median of 3)",
                        "note": "bound by the contract's definition; like the headline launch it is limited by a lone wave's VALU issue "
                                "(RK4 = four derivative evaluations per sub-step: ~1.5 x the Euler launch), DESIGN.md 4"},
           "config": {"workload": f"NavigationEnv.step, {N} agents/GPU, visual=False, bodyrate + RK4, dt=0.0025/ctrl_dt=0.02, ctrl_delay, "
                                  "drag_random=0.1, max_episode_steps=256 (BASELINE configs[2])"}}
    if cpu_ref and rank == 0:
        out["cpu_baseline"] = cpu_baseline_env(env.envs.dynamics.constants, "nav", N, seconds_target=3.0,
                                               note="oracle env.step with the RK4 integrator and the nominal drag coefficients (the per-agent "
                                                    "draw changes values, not the arithmetic)")
    env.close()
    return out


def _with_watchdog(fn, seconds, what):
    """secondary measurements must never cost the primary line: run fn() and give up after `seconds`"""
    import threading
    box = {}

    dev_index = torch.cuda.current_device()

    def work():
        try:
            torch.cuda.set_device(dev_index)          # the current device is per thread
            box["out"] = fn()
        except Exception as e:  # noqa: BLE001
            box["out"] = {"error": f"{type(e).__name__}: {e}"}

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return {"error": f"{what}: no result after {seconds} s"}, True
    return box["out"], False


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this very command line as N ranks (one per GPU) under
    torch.distributed.run on the loopback address -- what the driver's
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` does.
    Rank 0 of the children prints the ONE JSON line; the parent only forwards the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--agents", type=int, default=AGENTS_PER_GPU, help="agents per GPU")
    ap.add_argument("--chunk", type=int, default=16, help="steps per vf_env_step_n call = depth of the (K,N,...) output ring (16 x 3.6 MB stays in the Infinity Cache; profiles/r02_reset_regime.txt)")
    ap.add_argument("--no-reset-leg", action="store_true", help="skip the U(-1,1) reset-heavy leg (kernel-trace profiles of the headline regime)")
    ap.add_argument("--spinup-ms", type=float, default=25.0, help="untimed device spin-up before the warm-up steps (clock governor); 0 = off")
    ap.add_argument("--repeats", type=int, default=7, help="the timed --steps region is repeated; the median is reported")
    ap.add_argument("--sustain-s", type=float, default=6.0, help="seconds of back-to-back stepping reported as `sustained` (the driver's "
                    "gpu_busy samples see the GPU working; 0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--net-arch", default=None, help="--workload ppo: `pi=128,128:vf=32`; --workload bptt / shac: `pi=128,128[:qf=128,128]` -- a net_arch without a built-in chain class "
                                                     "(kernels compiled on first use, visfly_amd/_jit.py); the default is the reference's [64, 64] / [64, 64]")
    ap.add_argument("--activation", default=None, help="--workload ppo only: the policy's activation_fn (relu | tanh | elu | leaky_relu; default relu = the "
                                                        "reference's YAMLs; tanh = the policy class's own default): a generated chain class")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short PPO / BPTT runs embedded in the line")
    ap.add_argument("--workload", default="env", choices=["env", "ppo", "bptt", "shac", "nav_rk4_dr"],
                    help="env: fused HoverEnv.step (the BASELINE metric, default); ppo / bptt / shac / nav_rk4_dr (BASELINE configs[2]): that leg only")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins the process group, one barrier, rank 0 prints the world size "
                         "(no GPU work; tests/test_parallel_gloo.py runs the plain `--gpus 2` command through it on CPU)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))

    from visfly_amd import parallel
    build_info = None
    if not args.launch_check:
        # did THIS process compile the library or reuse the .so that travelled with the tree?  (recorded in the JSON line)
        from visfly_amd import _build
        stale = _build._stale()
        t_b = time.time()
        _build.build()
        build_info = {"rebuilt": bool(stale), "so_mtime": os.path.getmtime(_build.LIB), "so_bytes": os.path.getsize(_build.LIB),
                      "build_s": round(time.time() - t_b, 1) if stale else 0.0}
    if args.launch_check:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("VISFLY_AMD_DIST_BACKEND", "gloo"))
        assert dist.get_world_size() == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={dist.get_world_size()}"
        dist.barrier()
        if dist.get_rank() == 0:
            print(json.dumps({"launch_check": True, "n_gpus": dist.get_world_size()}), flush=True)
        dist.destroy_process_group()
        return
    # RCCL in production; VISFLY_AMD_DIST_BACKEND=gloo lets the N > 1 control flow be exercised on a box with fewer GPUs than
    # ranks (ranks then share devices round-robin) -- used by tests/test_parallel_gpu.py, never by the driver
    backend = os.environ.get("VISFLY_AMD_DIST_BACKEND", "nccl")
    if backend != "nccl":
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    rank, world, local = parallel.init(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
    if args.workload in ("ppo", "bptt", "shac", "nav_rk4_dr"):
        out = {"ppo": bench_ppo, "bptt": bench_bptt, "shac": bench_shac, "nav_rk4_dr": bench_nav_rk4_dr}[args.workload](args, rank, world, dev, cpu_ref=world == 1)
        if rank == 0:
            out["build"] = build_info
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    from visfly_amd.envs import HoverEnv
    N, K, W = args.agents, args.steps, args.warmup
    env = HoverEnv(num_agent_per_scene=N, num_scene=1, seed=42 + rank, visual=False, dynamics_kwargs=dict(DYN_KW),
                   device=dev, max_episode_steps=256, tensor_output=True, out_buffers=4)   # spawn box: HoverEnv default
    env.reset()
    dyn = env.envs.dynamics
    g = torch.Generator(device=dev).manual_seed(rank)
    hover = torch.tensor([-1 / 3, 0, 0, 0], device=dev)
    pool = (hover + (torch.rand((16, N, 4), device=dev, generator=g) * 2 - 1) * 0.02).clamp(-1, 1).contiguous()
    # the K actions of a timed region: a (K,N,4) sequence cycling through 16 distinct synthetic action batches
    Kc = min(K, args.chunk)                                     # regions longer than 512 steps are driven in chunks (output buffers: 3.6 MB per step)
    seq = pool.repeat(((Kc + 15) // 16, 1, 1))[:Kc].contiguous()
    wseq = pool.repeat(((max(W, 1) + 15) // 16, 1, 1))[:max(W, 1)].contiguous()

    tail = torch.cuda.Event()

    def drain():
        # poll an event first: hipDeviceSynchronize parks the thread and wakes up tens of microseconds late, which a
        # 20-step region (~250 us) would count as step time; the synchronize the contract asks for follows immediately
        tail.record()
        while not tail.query():
            pass
        torch.cuda.synchronize()

    def barrier():
        drain()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(n, s):
        """exactly n env steps, launch loop in C (vf_env_step_n): same kernels, same order, same outputs as n step() calls"""
        done = 0
        while done < n:
            k = min(n - done, s.shape[0])
            env.step_n(s if k == s.shape[0] else s[:k])
            done += k

    # Device spin-up, disclosed as timing.spinup_steps: the driver's W = 5 warm-up steps last 60 us, the GPU's clock governor needs
    # ~20 ms of load to leave its idle state (a 20-step region right after process start runs at 12.85 us per step, the same region
    # after 2 000 untimed steps at 12.1 -- `--spinup-ms 0` switches this off).  Untimed, before the W warm-up steps; the timed
    # regions below still consist of exactly K steps each.
    # The contract read literally -- W untimed warm-up steps, then exactly K timed steps, nothing else before them -- measured once,
    # right after process start and BEFORE the spin-up: reported as value_without_spinup next to the headline
    run_steps(min(K, seq.shape[0]), seq)                  # first use of the K-step output buffers is an allocation (untimed)
    if W > 0:
        run_steps(W, wseq)
    barrier()
    t0 = time.perf_counter()
    run_steps(K, seq)
    torch.cuda.synchronize()
    cold_el = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([cold_el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cold_el = float(t.item())
    spin = int(args.spinup_ms * 1e3 / 11.0)
    if spin > 0:
        run_steps(spin, seq)
        torch.cuda.synchronize()
    if W > 0:
        run_steps(W, wseq)
    walls, hosts, events, done_at = [], [], [], []
    for _ in range(max(1, args.repeats)):          # wall clock: nothing but the K launches between the two barriers
        barrier()
        t0 = time.perf_counter()
        run_steps(K, seq)
        t1 = time.perf_counter()
        tail.record()                              # drain(), with a time stamp in the middle:
        while not tail.query():
            pass
        t2 = time.perf_counter()                   # ... the K steps have completed (the event behind them reports it) ...
        torch.cuda.synchronize()                   # ... and the synchronize the contract asks for has returned:
        el = time.perf_counter() - t0              # this rank's clock stops here
        done_at.append(t2 - t0)
        barrier()                                  # the closing barrier follows, and the MAX over ranks is what is reported
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        walls.append(el)
        hosts.append(t1 - t0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):                              # the same region on the device's clock (HIP events on the launch stream)
        barrier()
        e0.record()
        run_steps(K, seq)
        e1.record()
        barrier()
        events.append(e0.elapsed_time(e1) * 1e-3)
    el = statistics.median(walls)
    # how often the reset path ran in the last chunk: the share of agent-steps that ended an episode and the share of steps in
    # which at least one agent did (one wave per SIMD: the launch lasts as long as its slowest wave, so ANY reset costs the step)
    dn = env._rollouts[seq.shape[0]]["done"]
    end_rate, steps_with_reset = float(dn.float().mean()), float(dn.any(dim=1).float().mean())

    # sustained leg: ~args.sustain_s seconds of the same launches back to back (one synchronize at the end).  Not the headline
    # (its timed region is seconds, not K steps) -- it is there so that a short --steps run leaves a trace an outside observer
    # (rocm-smi's busy counter, the driver's gpu_busy samples) can see, and as the steady-state figure of the same kernel
    sustained = None
    if args.sustain_s > 0:
        per = max(el / K, 1e-6)
        n_chunks = max(1, int(args.sustain_s / per / seq.shape[0]))
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_chunks):
            env.step_n(seq)
        torch.cuda.synchronize()
        sel = time.perf_counter() - t0
        barrier()
        sustained = {"value": world * N * n_chunks * seq.shape[0] / sel, "unit": "agent-steps/s", "steps": n_chunks * seq.shape[0],
                     "seconds": sel, "us_per_step": sel / (n_chunks * seq.shape[0]) * 1e6,
                     "note": "the headline's launches back to back for ~--sustain-s seconds, one synchronize at the end; per rank"}

    # open-loop rollout in ONE launch (vf_env_rollout_fused: agents stay in registers between the steps; same outputs bit for bit).
    # NOT the headline: a policy in the loop needs one launch per step; reported as a separate figure
    fused_walls = []
    run_fused = lambda: [env.step_n(seq[:min(K - d, seq.shape[0])], fused=True) for d in range(0, K, seq.shape[0])]
    run_fused()
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        run_fused()
        barrier()
        fused_walls.append(time.perf_counter() - t0)
    fused_el = statistics.median(fused_walls)

    # the reference-style driver for comparison: one Python env.step() call per step (zero-allocation output ring)
    for k in range(min(W, 50)):
        env.step(pool[k % 16])
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        env.step(pool[k % 16])
    t1 = time.perf_counter()
    barrier()
    per_call = {"ms_per_step": (time.perf_counter() - t0) / K * 1e3, "host_us_per_step": (t1 - t0) / K * 1e6,
                "driver": "Python loop over env.step(), out_buffers=4"}

    # dominant kernel, HIP events on the launch stream (BEFORE the reset-heavy leg below desynchronises the episodes)
    # kernel_us = HIP events over the timed K-step regions (median of the three event-bracketed repeats above): K launches back to
    # back on one stream, so events / K is the average launch duration plus whatever gap the launches leave (>= the kernel alone).
    # The second figure is 300 launches of ONE action from vf_env_time_steps -- same kernel, but the constant action lets agents
    # drift out of the box at different times, so part of it runs in the re-spawn regime (see with_resets).
    kern_us = statistics.median(events) / K * 1e6
    kern_us_300 = env.time_steps(pool[0], iters=300)
    dyn.time_steps(pool[0], iters=20)               # first use of k_dyn_step in this process: code load (~1 ms) stays out of the mean
    dyn_us = dyn.time_steps(pool[0], iters=300)
    assert torch.isfinite(dyn.state).all()

    # SURVEY 8(d) input 2, second run: actions U(-1,1) -- agents crash at random times, so the re-spawn path (device Philox
    # spawner, second collision query and observation row, terminal-observation stores) runs in essentially every step
    with_resets = None
    if not args.no_reset_leg:
        rseq = (torch.rand((seq.shape[0], N, 4), device=dev, generator=g) * 2 - 1).contiguous()
        run_steps(max(256, seq.shape[0]), rseq)         # let the crashes spread over the episode phase
        rwalls, revents = [], []
        for _ in range(max(1, args.repeats)):           # the same clock as the headline: stops when the synchronize returns
            barrier()
            t0 = time.perf_counter()
            run_steps(K, rseq)
            torch.cuda.synchronize()
            rwalls.append(time.perf_counter() - t0)
            barrier()
        for _ in range(3):                              # and the same device-clock figure (HIP events over the K launches / K)
            barrier()
            e0.record()
            run_steps(K, rseq)
            e1.record()
            barrier()
            revents.append(e0.elapsed_time(e1) * 1e-3)
        rel = statistics.median(rwalls)
        dn = env._rollouts[rseq.shape[0]]["done"]
        with_resets = {"value": world * N * K / rel, "unit": "agent-steps/s", "us_per_step": rel / K * 1e6,
                       "kernel_us": statistics.median(revents) / K * 1e6,
                       "vs_headline": el / rel, "kernel_us_vs_headline": (statistics.median(revents) / K * 1e6) / kern_us,
                       "episode_end_rate": float(dn.float().mean()), "steps_with_a_reset": float(dn.any(dim=1).float().mean()),
                       "timing": "median of the repeated --steps regions, clock as for the headline; kernel_us = HIP events over "
                                 "the K launches / K (median of 3)",
                       "actions": "U(-1,1) (SURVEY 8(d) input 2, second run): per rank, not max-reduced over ranks",
                       "spawn_prefetch": bool(env._ecfg.spawn_prefetch)}

    out = None
    if rank == 0:
        value = world * N * K / el
        achieved = BYTES_PER_ENV_STEP * N / (kern_us * 1e-6) / 1e9
        # HBM bytes per launch: NOT measured in this run -- read from the PMC pass committed under profiles/ (rocprofv3 --pmc in its
        # own run, as MI355X_MICROARCH.md prescribes; per-agent figure x N), and labelled as such (roofline.traffic_source)
        traffic, traffic_src = None, None
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                traffic = (pmc["fetch_bytes_per_agent"] + pmc["write_bytes_per_agent"]) * N
                traffic_src = f"profiles/{name} (builder's rocprofv3 --pmc pass on the same kernel; not measured in this run)"
                break
            except Exception:
                continue
        # the unit that is actually busy: fp32 VALU ISSUE of a single wave per SIMD.  Instruction count per wave
        # from the PMC pass committed under profiles/, 4.1 cycles per wave64 VALU instruction for a lone wave
        # (profiles/r02_valu_cost_probe.txt), 2.4 GHz.  Reported next to the HBM roofline the contract asks for.
        valu = None
        try:
            sq_name = "r03_pmc_env_sq.json" if os.path.exists(os.path.join(ROOT, "profiles", "r03_pmc_env_sq.json")) else "r02_pmc_env_sq.json"
            sq = json.load(open(os.path.join(ROOT, "profiles", sq_name)))["k_env_step"]
            per_wave = sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"]
            waves_per_simd = -(-(-(-N // 64)) // 1024)                 # ceil(waves / (256 CUs x 4 SIMDs))
            floor_us = waves_per_simd * per_wave * 4.1 / 2.4e3
            valu = {"valu_instr_per_wave": per_wave, "waves_per_simd": waves_per_simd, "cycles_per_instr_single_wave": 4.1,
                    "floor_us": floor_us, "frac": floor_us / kern_us,
                    "source": f"instruction count from profiles/{sq_name} (builder's PMC pass, not measured in this run); "
                              "kernel time from this run"}
        except Exception:
            pass
        out = {
            "metric": "agent-steps/sec (dynamics.step, visual=False)",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "value_without_spinup": world * N * K / cold_el,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"HoverEnv.step (fused dynamics + collision + reward + done + on-device auto-reset), "
                                   f"{N} agents/GPU, visual=False, bodyrate+euler, dt=0.0025/ctrl_dt=0.02, ctrl_delay, "
                                   "3-slot delay ring, max_episode_steps=256 (BASELINE configs[1])",
                       "agents_per_gpu": N, "parallelism": f"agents sharded x{world}, no data-path collective",
                       "driver": "env.step_n(): the K launches of the timed region are enqueued by one C call "
                                 "(vf_env_step_n); bit-identical to K env.step() calls (tests/test_env_multistep_gpu.py)"},
            "timing": {"repeats": len(walls), "statistic": "median of the repeated --steps regions, each bracketed by barrier + synchronize; "
                                                            "a rank's clock stops when ITS K steps have drained, the closing "
                                                            "barrier follows, max over ranks",
                       "ms_per_step_all": [w / K * 1e3 for w in walls], "ms_per_step_min": min(walls) / K * 1e3,
                       "host_us_per_step": statistics.median(hosts) / K * 1e6,
                       "event_us_per_step": statistics.median(events) / K * 1e6,
                       "wall_over_kernel": el / K * 1e6 / kern_us, "per_call": per_call,
                       "episode_end_rate": end_rate, "steps_with_a_reset": steps_with_reset,
                       "ms_per_step_at_completion": statistics.median(done_at) / K * 1e3,
                       "completion_note": "the same regions with the clock read when the event recorded behind the K launches reports "
                                          "completion, i.e. without the ~20 us torch.cuda.synchronize() takes to return on an already "
                                          "idle device (1 us per step at --steps 20); rank 0's figure",
                       "spinup_steps": spin, "spinup_note": "untimed env steps before the W warm-up steps so that the device is at "
                                                            "its sustained clocks (--spinup-ms, default 25 ms); value_without_spinup "
                                                            "= the first K-step region of the process, timed right after W warm-up "
                                                            "steps and before any spin-up (one region, max over ranks)",
                       "ms_per_step_without_spinup": cold_el / K * 1e3},
            "build": build_info,
            "sustained": sustained,
            "with_resets": with_resets,
            "rollout_fused": {"value": world * N * K / fused_el, "unit": "agent-steps/s", "us_per_step": fused_el / K * 1e6,
                              "driver": "env.step_n(fused=True): the K steps of the region in ONE launch (vf_env_rollout_fused), agents "
                                        "held in registers between the steps; open-loop only (actions known up front), bit-identical "
                                        "outputs (tests/test_env_multistep_gpu.py); per rank, not max-reduced over ranks"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_env_step<hover,bodyrate,euler,ctrl_delay>", "kernel_us": kern_us,
                         "kernel_us_source": "HIP events on the launch stream over the timed K-step regions / K (median of 3)",
                         "kernel_us_300_launches_one_action": kern_us_300,
                         "bytes_per_agent_step": BYTES_PER_ENV_STEP, "valu_issue": valu,
                         "note": "bound by the contract's definition (algorithmic HBM bytes / launch time); the launch is in fact "
                                 "limited by single-wave VALU issue (valu_issue: ~7.5 us of wave life, all but its first 0.7 us VALU) plus "
                                 "~2.5 us of dispatch ramp and completion per launch -- wave timeline in profiles/r04_env_timeline.txt, DESIGN.md 4",
                         "dyn_only": {"kernel": "k_dyn_step<bodyrate,euler,ctrl_delay>", "kernel_us": dyn_us,
                                      "bytes_per_agent_step": BYTES_PER_AGENT_STEP,
                                      "achieved": BYTES_PER_AGENT_STEP * N / (dyn_us * 1e-6) / 1e9}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dyn.constants)
            out["cpu_baseline_1core"] = cpu_baseline(dyn.constants, seconds_target=3.0, threads=1)
    env.close()

    # configs[3] / configs[4] / SHAC under the same clock: ONE PPO iteration, three regions of FOUR BPTT updates and of TWO SHAC iterations (median; after one warm-up each)
    if not args.no_secondary and os.environ.get("VISFLY_BENCH_SECONDARY", "1") != "0":
        sec, dead = {}, False
        for name, fn, iters in (("nav_rk4_dr", bench_nav_rk4_dr, K), ("ppo", bench_ppo, 1), ("bptt", bench_bptt, 4), ("shac", bench_shac, 2)):
            res, dead = _with_watchdog(lambda fn=fn, iters=iters: fn(args, rank, world, dev, iters=iters, cpu_ref=world == 1),
                                       240, name)
            sec[name] = res
            if dead:
                break
        if out is not None:
            out["secondary"] = sec
        if dead:                 # a hung collective: print what we have and leave without joining the stuck thread
            if out is not None:
                print(json.dumps(out), flush=True)
            os._exit(0)
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
