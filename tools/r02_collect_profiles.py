"""turn the outputs of tools/r02_profiles.sh (gpurun_out/r02p/) into the committed files under profiles/"""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, OUT = os.path.join(ROOT, "gpurun_out", "r02p"), os.path.join(ROOT, "profiles")
N = 65536


def counters(path):
    c = {}
    for line in open(path):
        m = re.match(r"(\w+)\s+n=\s*\d+\s+mean=\s*([\d.]+)", line)
        if m:
            c[m.group(1)] = float(m.group(2))
    return c


def per_wave(c):
    w = c["SQ_WAVES"]
    return {"wave_cycles": c["SQ_WAVE_CYCLES"] * 4 / w, "valu_instr": c["SQ_INSTS_VALU"] / w, "valu_busy_cycles": c["SQ_ACTIVE_INST_VALU"] * 4 / w,
            "wait_cycles": c["SQ_WAIT_ANY"] * 4 / w, "issue_stall_cycles": c["SQ_WAIT_INST_ANY"] * 4 / w}


for src, dst in (("r02_env_step_kernel_stats.txt",) * 2, ("r02_ppo_kernel_stats.txt",) * 2, ("r02_bptt_kernel_stats.txt",) * 2,
                 ("valu_cost_probe.txt", "r02_valu_cost_probe.txt"), ("fused_rollout.txt", "r02_fused_rollout.txt"),
                 ("step_n_drivers.txt", "r02_step_n_drivers.txt"), ("configs.txt", "r02_configs.txt")):
    shutil.copy(os.path.join(P, src), os.path.join(OUT, dst))
step, fused = counters(os.path.join(P, "pmc_step.txt")), counters(os.path.join(P, "pmc_fused.txt"))
fetch, write = step["FETCH_SIZE"] * 1024 * 2, step["WRITE_SIZE"] * 1024
json.dump({
    "source": "rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains) on "
              "`python tools/exp_env_one.py 65536 12 [fused]`, MI355X, r02 (tools/r02_profiles.sh); means per launch",
    "kernel": "k_env_step<hover,bodyrate,euler,ctrl_delay>", "agents": N,
    "FETCH_SIZE_KB_raw": step["FETCH_SIZE"], "WRITE_SIZE_KB_raw": step["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced stream -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE used as is",
    "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "fetch_bytes_per_agent": fetch / N, "write_bytes_per_agent": write / N,
    "reading": "traffic = %.1f MB/launch against 22.9 MB algorithmic (350 B/agent-step): no re-reads.  The kernel itself loads 144 B "
               "(7 granules + ring slot + action) and stores 201 B per agent." % ((fetch + write) / 1e6),
    "fused_rollout": {"kernel": "k_env_rollout<hover,bodyrate,euler,ctrl_delay>, 8 steps per launch", "FETCH_SIZE_KB_raw": fused["FETCH_SIZE"],
                      "WRITE_SIZE_KB_raw": fused["WRITE_SIZE"], "fetch_bytes_per_agent_step": fused["FETCH_SIZE"] * 1024 * 2 / N / 8,
                      "write_bytes_per_agent_step": fused["WRITE_SIZE"] * 1024 / N / 8,
                      "reading": "per agent-step ~34 B read (action 16 + ring slot 16 + 1/8 of the state) and ~80 B written (obs 52, reward 4, "
                                 "done 1, ring slot 16, 1/8 of the state): the state round trip of the per-step launch is gone"}},
    open(os.path.join(OUT, "r02_pmc_traffic.json"), "w"), indent=1)
ps, pf = per_wave(step), per_wave(fused)
old = json.load(open(os.path.join(OUT, "r02_pmc_env_sq.json")))
json.dump({
    "source": "rocprofv3 --pmc <SQ counters> --kernel-trace (two passes) on `python tools/exp_env_one.py 65536 12 [fused]`, MI355X, r02; means per "
              "launch (quad-cycle counters x4 in the per-wave figures)",
    "k_env_step": dict(step, per_wave=ps, reading="65536 agents = 1024 waves, one per SIMD.  A wave lives %.1fk cycles: %.0f VALU instructions busy for "
                       "%.1fk cycles (%.0f %%: 4.1 cycles per instruction, the single-wave issue rate measured by tools/valu_cost_probe), %.1fk cycles "
                       "(%.0f %%) parked at s_waitcnt (initial loads, LDS transpose, store queue), %.1f %% issue stalls." % (
                           ps["wave_cycles"] / 1e3, ps["valu_instr"], ps["valu_busy_cycles"] / 1e3, 100 * ps["valu_busy_cycles"] / ps["wave_cycles"],
                           ps["wait_cycles"] / 1e3, 100 * ps["wait_cycles"] / ps["wave_cycles"], 100 * ps["issue_stall_cycles"] / ps["wave_cycles"])),
    "k_env_rollout_8_steps": dict(fused, per_wave=pf, per_step={k: v / 8 for k, v in pf.items()},
                                  reading="per step %.1fk wave cycles, %.0f %% of them VALU-busy, %.0f %% parked: with the state in registers across the "
                                          "steps the launch is within ~25 %% of its VALU-issue floor" % (
                                              pf["wave_cycles"] / 8e3, 100 * pf["valu_busy_cycles"] / pf["wave_cycles"], 100 * pf["wait_cycles"] / pf["wave_cycles"])),
    "split_vs_plain_r02e": old.get("split_vs_plain_r02e")},
    open(os.path.join(OUT, "r02_pmc_env_sq.json"), "w"), indent=1)
print(ps, pf)
