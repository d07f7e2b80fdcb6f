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
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

AGENTS_PER_GPU = 65536
DYN_KW = dict(action_type="bodyrate", integrator="euler", dt=0.0025, ctrl_dt=0.02, ctrl_delay=True)
BYTES_PER_AGENT_STEP = 244      # 116 B read + 128 B written per agent-step (Dynamics.step)
BYTES_PER_ENV_STEP = 350        # full env.step adds obs, reward, counters and flags
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(consts, seconds_target=15.0):
    """TEST/BENCH INFRASTRUCTURE: time the CPU oracle (C restatement, OpenMP over agents) on the host
    cores of this box on a bounded sample of the same workload: HoverEnv.step = dynamics interval +
    bbox collision + reward / counters / done masks (no auto-reset inside the sample: episodes are
    256 steps long and the sample is re-spawned every 192 steps)."""
    import oracle
    N = AGENTS_PER_GPU
    env = oracle.OracleEnv(consts, N, "hover", 256, target=(1.0, 0.0, 1.5))
    rng = np.random.default_rng(0)
    fs = np.zeros((N, 22), np.float32)
    fs[:, 0:3] = (np.array([1, 0, 1.5]) + rng.uniform(-1, 1, (N, 3)) * np.array([1, 1, .5])).astype(np.float32)
    fs[:, 3] = 1.0
    fs[:, 13:17], fs[:, 17:21] = consts["w_init"], consts["T_init"]
    a = np.clip(np.array([-1 / 3, 0, 0, 0]) + rng.uniform(-.02, .02, (N, 4)), -1, 1).astype(np.float32)
    env.reset_full_state(fs)
    for _ in range(2):
        env.step(a)
    t0 = time.perf_counter()
    steps = 0
    while True:
        if steps % 192 == 0:
            env.reset_full_state(fs)
        for _ in range(8):
            env.step(a)
        steps += 8
        el = time.perf_counter() - t0
        if el > seconds_target or steps >= 8192:
            break
    cores = len(os.sched_getaffinity(0))
    return {"value": N * steps / el, "unit": "agent-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/vf_oracle.c HoverEnv.step restatement (OpenMP, {cores} threads), N={N}, "
                      f"{steps} control steps, {el:.1f} s"}


def bench_ppo(args, rank, world, dev):
    """secondary measurement (not the driver's metric): NavigationEnv + PPO full train loop,
    agents sharded by rank, one RCCL all-reduce of the flat gradient per optimiser step."""
    from visfly_amd import parallel
    from visfly_amd.envs import NavigationEnv
    from visfly_amd.ppo import PPO
    N = args.agents if args.agents != AGENTS_PER_GPU else 32768
    spawn = {"state_generator": {"class": "Uniform", "kwargs": [{"position": {"mean": [1., 0., 1.5], "half": [0., 2., 1.]}}]}}
    env = NavigationEnv(num_agent_per_scene=N, seed=42 + rank, dynamics_kwargs=dict(DYN_KW), random_kwargs=spawn,
                        device=dev, max_episode_steps=256)
    ppo = PPO(env, n_steps=256, batch_size=25600, n_epochs=5, learning_rate=1e-4, seed=0)
    ppo.learn(256 * N * world)   # warm-up iteration
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    iters = max(1, args.steps // 256)
    ppo.learn(256 * N * world * iters)
    torch.cuda.synchronize()
    parallel.barrier()
    el = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    # MFMA roofline of the optimiser steps (outside the timed region): one more train() pass over the last rollout,
    # timed with device events; algorithmic flops = 2 x weights x rows for each of forward, data gradient, weight gradient
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n0 = ppo._opt_step
    e0.record()
    ppo.train()
    e1.record()
    torch.cuda.synchronize()
    n_upd = max(1, ppo._opt_step - n0)
    us_upd = e0.elapsed_time(e1) * 1e3 / n_upd
    rows = min(ppo.batch_size, 256 * N)
    flops = 6.0 * ppo.policy.log_std_off * rows          # log_std_off = number of weights + biases of the network
    tfs = flops / (us_upd * 1e-6) / 1e12
    roof = {"bound": "mfma", "achieved": tfs, "peak": 157.3, "unit": "TFLOP/s", "frac": tfs / 157.3, "traffic": None,
            "kernel": "optimiser step: k_ppo_update_chain + k_mlp_wgrad + fold + grad norm + Adam", "us_per_update": us_upd,
            "rows": rows, "note": "fp32 v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md); PMC breakdown in profiles/r01_pmc_mlp.json"}
    if rank == 0:
        print(json.dumps({"metric": "PPO env-steps/s (rollout + train, NavigationEnv, StateTarget MLP)",
                          "value": 256 * N * world * iters / el, "unit": "agent-steps/s", "n_gpus": world,
                          "iterations": iters, "s_per_iteration": el / iters, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"NavigationEnv {N} agents/GPU, n_steps=256, batch 25600/GPU, 5 epochs",
                                     "logs": {k: float(v) for k, v in ppo.logs.items()}},
                          "roofline": roof}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def bench_bptt(args, rank, world, dev):
    """secondary measurement (BASELINE configs[4]): RacingEnv, thrust actions, BPTT H=64 through the adjoint kernel,
    agents sharded by rank, one all-reduce of the flat gradient per update."""
    from visfly_amd import parallel
    from visfly_amd.bptt import BPTT
    from visfly_amd.envs import RacingEnv
    N = args.agents if args.agents != AGENTS_PER_GPU else 16384
    dkw = dict(DYN_KW, action_type="thrust")
    env = RacingEnv(num_agent_per_scene=N, seed=42 + rank, dynamics_kwargs=dkw, device=dev, max_episode_steps=256,
                    requires_grad=True, tensor_output=True)
    algo = BPTT(env, horizon=64, gamma=0.99, learning_rate=1e-3, seed=0)
    algo.learn(64 * N * world)     # warm-up update
    torch.cuda.synchronize()
    parallel.barrier()
    t0 = time.perf_counter()
    iters = max(1, args.steps // 64)
    algo.learn(64 * N * world * iters)
    torch.cuda.synchronize()
    parallel.barrier()
    el = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(json.dumps({"metric": "BPTT env-steps/s (H=64 rollout + adjoint + actor update, RacingEnv)",
                          "value": 64 * N * world * iters / el, "unit": "agent-steps/s", "n_gpus": world,
                          "iterations": iters, "s_per_iteration": el / iters, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"RacingEnv {N} agents/GPU, thrust actions, horizon 64",
                                     "logs": {k: float(v) for k, v in algo.logs.items()}}}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--agents", type=int, default=AGENTS_PER_GPU, help="agents per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="env", choices=["env", "ppo", "bptt"],
                    help="env: fused HoverEnv.step (the BASELINE metric, default); ppo: full PPO loop (configs[3] shape)")
    args = ap.parse_args()

    from visfly_amd import parallel
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
    if args.workload == "bptt":
        return bench_bptt(args, rank, world, dev)
    if args.workload == "ppo":
        return bench_ppo(args, rank, world, dev)

    from visfly_amd.envs import HoverEnv
    N = args.agents
    env = HoverEnv(num_agent_per_scene=N, num_scene=1, seed=42 + rank, visual=False, dynamics_kwargs=dict(DYN_KW),
                   device=dev, max_episode_steps=256, tensor_output=True)   # spawn box: HoverEnv default
    env.reset()
    dyn = env.envs.dynamics
    g = torch.Generator(device=dev).manual_seed(rank)
    hover = torch.tensor([-1 / 3, 0, 0, 0], device=dev)
    pool = [(hover + (torch.rand((N, 4), device=dev, generator=g) * 2 - 1) * 0.02).clamp(-1, 1).contiguous()
            for _ in range(16)]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    step = env.step
    for k in range(args.warmup):
        step(pool[k % 16])
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(pool[k % 16])
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # dominant kernel, HIP events on the launch stream
    kern_us = env.time_steps(pool[0], iters=200)
    dyn_us = dyn.time_steps(pool[0], iters=200)
    assert torch.isfinite(dyn.state).all()

    if rank == 0:
        value = world * N * args.steps / el
        achieved = BYTES_PER_ENV_STEP * N / (kern_us * 1e-6) / 1e9
        traffic = None   # HBM bytes per launch from the PMC passes committed under profiles/ (per-agent figure x N)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic = (pmc["fetch_bytes_per_agent"] + pmc["write_bytes_per_agent"]) * N
        except Exception:
            pass
        valu = None      # second roofline: fp32 VALU issue, from the PMC instruction count committed under profiles/
        try:
            sq = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_env_sq.json")))
            per_wave = sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"]
            waves_per_simd = -(-(-(-N // 64)) // 1024)                 # ceil(waves / (256 CUs x 4 SIMDs))
            floor_us = waves_per_simd * per_wave * 4 / 2.4e3          # 4 cycles per wave64 VALU instruction at 2.4 GHz
            valu = {"valu_instr_per_wave": per_wave, "waves_per_simd": waves_per_simd, "floor_us": floor_us,
                    "frac": floor_us / kern_us}
        except Exception:
            pass
        out = {
            "metric": "agent-steps/sec (dynamics.step, visual=False)",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"HoverEnv.step (fused dynamics + collision + reward + done + on-device auto-reset), "
                                   f"{N} agents/GPU, visual=False, bodyrate+euler, dt=0.0025/ctrl_dt=0.02, ctrl_delay, "
                                   "3-slot delay ring, max_episode_steps=256 (BASELINE configs[1])",
                       "agents_per_gpu": N, "parallelism": f"agents sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_env_step<hover,bodyrate,euler,ctrl_delay>", "kernel_us": kern_us,
                         "bytes_per_agent_step": BYTES_PER_ENV_STEP, "valu": valu,
                         "dyn_only": {"kernel": "k_dyn_step<bodyrate,euler,ctrl_delay>", "kernel_us": dyn_us,
                                      "bytes_per_agent_step": BYTES_PER_AGENT_STEP,
                                      "achieved": BYTES_PER_AGENT_STEP * N / (dyn_us * 1e-6) / 1e9}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dyn.constants)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
