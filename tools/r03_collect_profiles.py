"""turn the outputs of tools/r03_profiles.sh (gpurun_out/r03p/) into the committed files under profiles/"""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, OUT = os.path.join(ROOT, "gpurun_out", "r03p"), os.path.join(ROOT, "profiles")
N = 65536


def counters(path):
    c = {}
    for line in open(path):
        m = re.match(r"(\w+)\s+n=\s*\d+\s+mean=\s*([\d.]+)", line)
        if m:
            c[m.group(1)] = float(m.group(2))
    return c


for src, dst in (("r03_env_step_kernel_stats.txt",) * 2, ("r03_ppo_kernel_stats.txt",) * 2, ("r03_bptt_kernel_stats.txt",) * 2):
    shutil.copy(os.path.join(P, src), os.path.join(OUT, dst))
off, on = counters(os.path.join(P, "pmc_mode0.txt")), counters(os.path.join(P, "pmc_mode3.txt"))


def traffic(c):
    fetch, write = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
    return fetch, write


f0, w0 = traffic(off)
f3, w3 = traffic(on)
json.dump({
    "source": "rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains) on "
              "`python tools/exp_env_one.py 65536 12`, MI355X, r03 (tools/r03_profiles.sh); means per launch; the headline launch = "
              "prefetched re-spawn ON (1024 main + 1024 helper waves)",
    "kernel": "k_env_step<hover,bodyrate,euler,ctrl_delay>", "agents": N,
    "FETCH_SIZE_KB_raw": on["FETCH_SIZE"], "WRITE_SIZE_KB_raw": on["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced stream -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE used as is",
    "fetch_bytes_per_launch": f3, "write_bytes_per_launch": w3, "fetch_bytes_per_agent": f3 / N, "write_bytes_per_agent": w3 / N,
    "prefetch_off": {"fetch_bytes_per_agent": f0 / N, "write_bytes_per_agent": w0 / N},
    "reading": "traffic = %.1f MB/launch against 22.9 MB algorithmic (350 B/agent-step): no re-reads.  The helper blocks of the prefetched "
               "re-spawn add %.1f B/agent of loads (flag word + tag of the copy they guard) and, in the first launches after a reset, the "
               "stores of the copies they refill (prefetch off: %.1f MB/launch)." % ((f3 + w3) / 1e6, (f3 - f0) / N, (f0 + w0) / 1e6)},
    open(os.path.join(OUT, "r03_pmc_traffic.json"), "w"), indent=1)


def per_wave(c, waves):
    return {"wave_cycles": c["SQ_WAVE_CYCLES"] * 4 / waves, "valu_instr": c["SQ_INSTS_VALU"] / waves, "valu_busy_cycles": c["SQ_ACTIVE_INST_VALU"] * 4 / waves,
            "wait_cycles": c["SQ_WAIT_ANY"] * 4 / waves, "issue_stall_cycles": c["SQ_WAIT_INST_ANY"] * 4 / waves}


ps = per_wave(off, off["SQ_WAVES"])
# helper waves = the difference between the two launches, spread over the 1024 extra waves
helper = {k: (on[k] - off[k]) for k in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_SALU")}
json.dump({
    "source": "rocprofv3 --pmc <SQ counters> --kernel-trace (two passes) on `python tools/exp_env_one.py 65536 12`, MI355X, r03; means per launch "
              "(quad-cycle counters x4 in the per-wave figures).  k_env_step = the MAIN waves (pass with VISFLY_AMD_PREFETCH_MODE=0: 1024 waves, "
              "what bench.py's valu_issue figure uses); with_helper_blocks = the launch as the bench runs it (2048 waves)",
    "k_env_step": dict(off, per_wave=ps, reading="65536 agents = 1024 main waves, one per SIMD.  A wave lives %.1fk cycles: %.0f VALU instructions busy for "
                       "%.1fk cycles (%.0f %%), %.1fk cycles (%.0f %%) parked at s_waitcnt, %.1f %% issue stalls -- unchanged from r02." % (
                           ps["wave_cycles"] / 1e3, ps["valu_instr"], ps["valu_busy_cycles"] / 1e3, 100 * ps["valu_busy_cycles"] / ps["wave_cycles"],
                           ps["wait_cycles"] / 1e3, 100 * ps["wait_cycles"] / ps["wave_cycles"], 100 * ps["issue_stall_cycles"] / ps["wave_cycles"])),
    "with_helper_blocks": dict(on, helper_waves_total=helper,
                               reading="the 1024 helper waves add %.0f VALU instructions and %.0f quad-cycles of wave life in total per launch "
                                       "(%.0f instructions per helper wave: two loads, a compare, exit) in the steady state where no copy is stale" % (
                                           helper["SQ_INSTS_VALU"], helper["SQ_WAVE_CYCLES"], helper["SQ_INSTS_VALU"] / 1024))},
    open(os.path.join(OUT, "r03_pmc_env_sq.json"), "w"), indent=1)
print(ps, helper)
