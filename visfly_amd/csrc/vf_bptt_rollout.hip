// vf_bptt_rollout.hip -- the forward half of a BPTT horizon as ONE persistent launch (gfx950).
//
// BPTT.learn's inner loop (utils/algorithms/BPTT.py:107-124) is closed: policy(obs_t) -> action_t -> env.step -> obs_{t+1}.
// As separate launches that is, per control step, the register-chained policy forward + action head, the fused env step and
// the loss / discount bookkeeping + state checkpoint: three launches of 256-1024 waves each, every one of them latency-bound
// (~4 us of launch boundary, first-touch misses on kernel arguments and weights, the state round trip through HBM), 32-35 us per
// step at 16 384 agents for ~17 us of work.  Here a wave owns 16 agents for the whole horizon:
//   * the policy forward is the 16-rows-per-wave chain of vf_mlp_chain.hpp (v_mfma_f32_16x16x4_f32, activations in accumulator
//     registers, weights from the transposed image of the packed buffer -- L2-resident after the first step);
//   * the env step is the same per-agent code as k_env_step (control_interval + env_epilogue), the agent's state staying in
//     registers from step to step like in k_env_rollout.  One lane per agent: lanes 16..63 replicate the agent of lane & 15
//     (same loads, same arithmetic, same stores of the same values) -- the step is bound by single-wave instruction issue,
//     not by lanes, so the idle three quarters of the wave cost nothing, and no code path needs a 16-lane special case;
//   * observation rows and actions travel through the per-step slot buffers the reverse sweep needs anyway (the observation
//     copy of slot t + 1 IS what step t's epilogue writes; the action row IS the tape's action), so nothing is copied;
//   * per step the wave also writes the adjoint's checkpoint (its agents' granules of tape row t) and advances the loss /
//     discount recurrence in registers.
// Everything the per-step path leaves behind -- saved activations of every slot, actions, tape rows, done flags, d_reward
// rows, loss, episode outputs, final slab -- is bit-identical (tests/test_bptt_gpu.py), so the reverse sweep is unchanged.
#include "vf_chain_plugin.hpp"
#include "vf_bptt_rollout_kernel.hpp"

extern "C" int vf_bptt_rollout(vf_env* h, const vf_mlp_desc* desc, const float* params, const float* packed, const float* obs_slots0,
                               const float* obs_slots1, const float* log_std, const float* eps, float* actions,
                               const vf_env_out* out, float* obs_final, float* tape, int64_t tape_stride, uint8_t* tape_done,
                               float* d_reward, float* loss, float* disc, float gamma, float scale, int32_t H, float* substep_tape,
                               float* mean_rows, float* log_std_rows, float* reward_rows, uint8_t* ep_flag_rows, vf_stream_t stream)
{
    if (!h || !desc || !params || !obs_slots0 || !eps || !actions || !out || !obs_final || !tape || !tape_done || !d_reward ||
        !loss || !disc || H <= 0)
        return vf::fail(VF_EINVAL, "vf_bptt_rollout: bad argument");
    if (!out->reward) return vf::fail(VF_EINVAL, "vf_bptt_rollout: out->reward (N floats of scratch) is required");
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_bptt_rollout: vf_env_bind has not been called");
    if (h->dyn.wind) return vf::fail(VF_EUNSUPPORTED, "vf_bptt_rollout: per-agent wind rows are set");
    if (tape_stride < (int64_t)h->dyn.Npad * h->dyn.G * 4) return vf::fail(VF_EINVAL, "vf_bptt_rollout: tape rows are shorter than the slab");
    if (reinterpret_cast<uintptr_t>(substep_tape) & 15) return vf::fail(VF_EINVAL, "vf_bptt_rollout: substep_tape must be 16-byte aligned");
    if ((int64_t)H * h->dyn.N * 128 * 4 >= (1ll << 32))      // the chain addresses its activation copies with 32-bit byte offsets
        return vf::fail(VF_EUNSUPPORTED, "vf_bptt_rollout: H x N rows of activation copies pass 4 GiB per buffer");
    const bool race2 = h->cfg.kind == VF_ENV_RACING && h->cfg.obs_mode == VF_OBS_RACE2;      // RacingEnv2: 16 gate-relative columns
    const int OW = race2 ? 16 : 13;
    if (desc->in_dim[0] != OW) return vf::fail(VF_EUNSUPPORTED, "vf_bptt_rollout: the first observation must be the %d-wide state row", OW);
    // (RacingEnv2: out->obs / out->terminal_obs / obs_slots0 / obs_final rows are 16 wide)
    const int cls = vf::chain16_policy_class(desc, params);
    // td_policies.Actor: the second head is the state-dependent log_std.  cls == 0: not a built-in class -- a generated one runs from its
    // BPTT plugin below (visfly_amd/_jit.py: ensure_bptt), told apart by the width of the table's second head
    bool sac = cls >= 3;
    if (cls == 0)
        for (int l = 0; l < desc->n_layers; ++l)
            if (desc->layer[l].dst == VF_MLP_OUT1) sac = desc->layer[l].No == 4;
    if (sac ? !log_std_rows : !log_std)
        return vf::fail(VF_EINVAL, sac ? "vf_bptt_rollout: log_std_rows (H N, 4) is required for the two-headed actor classes"
                                       : "vf_bptt_rollout: log_std is required for the state-independent-log_std classes");
    if ((reinterpret_cast<uintptr_t>(mean_rows) | reinterpret_cast<uintptr_t>(log_std_rows)) & 15)
        return vf::fail(VF_EINVAL, "vf_bptt_rollout: mean_rows / log_std_rows must be 16-byte aligned");
    vf::RollKernel k = nullptr;
    if (cls == 0) k = nullptr;
    else if (race2) k = obs_slots1 ? nullptr : vf::pick_roll_race2(cls, h->dyn.cfg);
    else if ((cls == 1 || cls == 3) && h->cfg.kind == VF_ENV_NAV) k = obs_slots1 ? nullptr : vf::pick_roll_nav2(cls, h->dyn.cfg);
    else if (!h->dyn.cfg.ctrl_delay) k = (cls == 2 || cls == 4) && !obs_slots1 ? nullptr : vf::pick_roll_nodelay(cls, h->cfg.kind, h->dyn.cfg);
    else if (cls == 1 && h->cfg.kind == VF_ENV_HOVER) k = vf::pick_roll<vf::NetHoverPi, VF_ENV_HOVER>(h->dyn.cfg);
    else if (cls == 1 && h->cfg.kind == VF_ENV_RACING) k = vf::pick_roll<vf::NetHoverPi, VF_ENV_RACING>(h->dyn.cfg);
    else if (cls == 2 && h->cfg.kind == VF_ENV_NAV && obs_slots1) k = vf::pick_roll<vf::NetNavPi, VF_ENV_NAV>(h->dyn.cfg);
    else if (sac && (cls == 3 || obs_slots1)) k = vf::pick_roll_sac(cls, h->cfg.kind, h->dyn.cfg);
    if (!k && cls != 0)
        return vf::fail(VF_EUNSUPPORTED, "vf_bptt_rollout: no persistent roll-out for this network class / env kind / dynamics "
                                         "configuration (policy trunk or td_policies.Actor over [128, 64] x [64, 64], thrust / bodyrate, Euler / RK4)");
    const int N = h->dyn.N;
    vf::EnvArgs ge{vf::DynArgs{N, h->dyn.G, h->dyn.g_drag, h->dyn.S, reinterpret_cast<const float4*>(actions), nullptr,
                               vf::ring_head(&h->dyn), nullptr, h->dyn.vel_strided},
                   *out, h->g_race, 1};
    ge.out.done = tape_done;
    ge.out.done_list = ge.out.done_count = nullptr;
    // slot t + 1's observation rows are what step t writes; slot 0 holds the current observation (caller)
    ge.out.obs = H > 1 ? const_cast<float*>(obs_slots0) + (size_t)N * OW : obs_final;
    // (the policy-only classes do not run the second trunk: log_std_rows is written by the two-headed classes only)
    vf::ChainArgs gc{*desc, params, packed, vf::ChainIo{{obs_slots0, obs_slots1}, mean_rows, sac ? log_std_rows : nullptr}, H * N, log_std,
                     reinterpret_cast<const float4*>(eps), reinterpret_cast<float4*>(actions), {nullptr, nullptr}, VF_SAC_LOG_STD_MIN,
                     VF_SAC_LOG_STD_MAX};
    vf::RollArgs r{H, N, tape, tape_stride, tape_done, d_reward, loss, disc, const_cast<float*>(obs_slots0), obs_final, gamma, scale, reinterpret_cast<float4*>(substep_tape), reward_rows,
                   ep_flag_rows};
    if (k) {
        hipLaunchKernelGGL(k, dim3((N + 15) / 16), dim3(64), 0, vf::as_stream(stream), h->dyn.d_cfg, h->d_cfg, ge, gc, r);
        VF_HIP(hipGetLastError());
    } else {
        // a generated actor class: its BPTT plugin, compiled on first use for this env kind / dynamics configuration (visfly_amd/_jit.py)
        int rc = 0;
        const int kkind = race2 ? vf::VF_ENV_RACING2 : h->cfg.kind;
        for (int i = 0; i < vf::chain_plugin_count() && rc == 0; ++i) {
            const vf::ChainPlugin* p = vf::chain_plugin(i);
            if (p->bptt_rollout && p->bptt_roll_abi == vf::kBpttRollPluginAbi)
                rc = p->bptt_rollout(desc, params, kkind, &h->dyn.cfg, obs_slots1 != nullptr, h->dyn.d_cfg, h->d_cfg, &ge, &gc, &r, N, vf::as_stream(stream));
        }
        if (rc <= -1000) return vf::fail(VF_EHIP, "vf_bptt_rollout (chain plugin) failed: %s", hipGetErrorString((hipError_t)(-rc - 1000)));
        if (rc == 0)
            return vf::fail(VF_EUNSUPPORTED, "vf_bptt_rollout: the policy's layer table is not one of the built-in register-chained classes and no "
                                             "BPTT plugin of a generated class serves it under this env kind / dynamics configuration");
        vf::chain_plugin_count_launch();
    }
    h->dyn.tick += H;
    h->stale_all = 1;       // agents re-spawned inside the launch: the prefetched copies' stale bits no longer cover them
    return VF_OK;
}
