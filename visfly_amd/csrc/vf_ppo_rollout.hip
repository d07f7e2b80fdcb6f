// vf_ppo_rollout.hip -- PPO's collect_rollouts as ONE persistent launch (gfx950).
//
// SB3 OnPolicyAlgorithm.collect_rollouts (run by utils/algorithms/PPO.py:146) is the closed loop policy.forward(obs_t) ->
// distribution.sample / log_prob -> env.step -> RolloutBuffer.add, n_steps times.  Launch by launch that is, per control step
// at 32 768 agents: the register-chained forward (30 us), the head sampler (5 us), two observation copies (2 x 5 us), the env
// step (11 us) and the TimeLimit bookkeeping (4 us) -- 60 us, of which the launch boundaries and the cold first touches of
// every launch are a third (profiles/r03_ppo_kernel_stats.txt).  Here, as in vf_bptt_rollout.hip, a wave owns its agents for
// all n_steps:
//   * ROWS = 32 agents per wave with the 32-row chain (v_mfma_f32_32x32x2_f32) or 16 with the 16-row chain -- the SAME choice
//     vf_mlp_forward makes from the row count (chain16_ok), so heads, values, actions and log-probs are the per-step path's
//     to the bit.  Lanes ROWS..63 replicate lane & (ROWS - 1) outside the chain (k_bptt_rollout's construction);
//   * nothing of a step travels through memory: the accumulator lane that holds the heads of row m IS lane m (replicas take
//     them by ds_bpermute), the sampled action feeds the env step in registers, the reward / done of the epilogue feed the buffer
//     rows and the TimeLimit list, and the next forward reads its "state" row from the LDS tile the epilogue staged it through
//     on its way to RolloutBuffer.obs["state"][t + 1].  Global memory sees only the buffer rows, written once, never re-read.
#include "vf_chain_plugin.hpp"
#include "vf_ppo_rollout_kernel.hpp"

namespace {

using PpoRollKernel = void (*)(const vf_dyn_cfg*, const vf_env_cfg*, const vf::EnvArgs, const vf::ChainArgs, const vf::PpoRollArgs);

template <class Net, int ROWS, int KIND, bool DELAY>
PpoRollKernel pick_ppo_roll2(const vf_dyn_cfg& c)
{
    if (c.integrator == VF_INT_RK4) {      // BASELINE configs[2]'s dynamics (utils/maths.py:353-386, repaired): bodyrate; thrust too
        if (c.action_type == VF_ACT_THRUST) return vf::k_ppo_rollout<Net, ROWS, KIND, VF_ACT_THRUST, VF_INT_RK4, DELAY>;
        if (c.action_type == VF_ACT_BODYRATE) return vf::k_ppo_rollout<Net, ROWS, KIND, VF_ACT_BODYRATE, VF_INT_RK4, DELAY>;
        return nullptr;
    }
    if (c.action_type == VF_ACT_THRUST) return vf::k_ppo_rollout<Net, ROWS, KIND, VF_ACT_THRUST, VF_INT_EULER, DELAY>;
    if (c.action_type == VF_ACT_BODYRATE) return vf::k_ppo_rollout<Net, ROWS, KIND, VF_ACT_BODYRATE, VF_INT_EULER, DELAY>;
    return nullptr;
}

// motor lag (ctrl_delay, the reference's default) or the direct form of the interval (envs/base/dynamics.py:510-554; r05)
template <class Net, int ROWS, int KIND>
PpoRollKernel pick_ppo_roll(const vf_dyn_cfg& c)
{
    return c.ctrl_delay ? pick_ppo_roll2<Net, ROWS, KIND, true>(c) : pick_ppo_roll2<Net, ROWS, KIND, false>(c);
}

}  // namespace

extern "C" int vf_ppo_rollout(vf_env* h, const vf_mlp_desc* desc, const float* params, const float* packed, const vf_ppo_rollout_args* a,
                              vf_stream_t stream)
{
    if (!h || !desc || !params || !packed || !a || !a->out || !a->obs_state || !a->values || !a->actions || !a->log_probs || !a->rewards ||
        !a->episode_starts || !a->last_starts || !a->obs_final || !a->log_std || !a->cursor || !a->idx_list || !a->rows0 ||
        !a->stat || a->T <= 0 || a->w1 < 0 || (a->w1 > 0 && (!a->rows1 || !a->obs_target_row)))
        return vf::fail(VF_EINVAL, "vf_ppo_rollout: bad argument");
    const vf_env_out* out = a->out;
    if (!out->reward || !out->done || !out->ep_return || !out->ep_length || !out->ep_flags || !out->terminal_obs)
        return vf::fail(VF_EINVAL, "vf_ppo_rollout: out needs reward / done scratch and the episode outputs");
    for (int l = 0; l < desc->n_layers; ++l)
        if (desc->layer[l].save) return vf::fail(VF_EINVAL, "vf_ppo_rollout: the layer table must not keep activation copies (layer %d has a save pointer)", l);
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_ppo_rollout: vf_env_bind has not been called");
    if (h->dyn.wind) return vf::fail(VF_EUNSUPPORTED, "vf_ppo_rollout: per-agent wind rows are set");
    // (observation / reward variants -- HoverEnv2, NavigationEnv2 -- are the epilogue's own: the rows it stages through the wave's
    // tile for the next forward are already in the env's obs_mode, the terminal rows too; r05)
    if ((reinterpret_cast<uintptr_t>(a->means) | reinterpret_cast<uintptr_t>(a->actions) | reinterpret_cast<uintptr_t>(a->stat)) & 15)
        return vf::fail(VF_EINVAL, "vf_ppo_rollout: means / actions / stat must be 16-byte aligned");
    const int N = h->dyn.N, T = a->T;
    // the rows-per-wave choice of vf_mlp_forward for N rows (chain16_ok), so that the heads are the per-step path's to the bit
    const int cls = vf::chain_full_class(desc, params, N);        // 0 none; 1 NetHover, 2 NetNav; + 16 when the 16-row chain runs N rows
    const bool r16 = (cls & 16) != 0;
    PpoRollKernel k = nullptr;
    if ((cls & 15) == 1 && h->cfg.kind == VF_ENV_HOVER)
        k = r16 ? pick_ppo_roll<vf::NetHover, 16, VF_ENV_HOVER>(h->dyn.cfg) : pick_ppo_roll<vf::NetHover, 32, VF_ENV_HOVER>(h->dyn.cfg);
    else if ((cls & 15) == 2 && h->cfg.kind == VF_ENV_NAV && a->obs_target)
        k = r16 ? pick_ppo_roll<vf::NetNav, 16, VF_ENV_NAV>(h->dyn.cfg) : pick_ppo_roll<vf::NetNav, 32, VF_ENV_NAV>(h->dyn.cfg);
    else if ((cls & 15) == 1 && h->cfg.kind == VF_ENV_NAV && !a->obs_target)      // NavigationEnv2: the target is inside the "state" row
        k = r16 ? pick_ppo_roll<vf::NetHover, 16, VF_ENV_NAV>(h->dyn.cfg) : pick_ppo_roll<vf::NetHover, 32, VF_ENV_NAV>(h->dyn.cfg);
    // r06: RacingEnv (the raw state row) and RacingEnv2 (obs_mode VF_OBS_RACE2: the 16 gate-relative columns of the agent's current gate,
    // formed by the epilogue -- kernel-side kind VF_ENV_RACING2) under the one-observation class; the motor-lag form of the interval
    const bool race2 = h->cfg.kind == VF_ENV_RACING && h->cfg.obs_mode == VF_OBS_RACE2;
    const int OW = race2 ? 16 : 13;
    if (desc->in_dim[0] != OW) k = nullptr;
    else if ((cls & 15) == 1 && h->cfg.kind == VF_ENV_RACING && !a->obs_target && h->dyn.cfg.ctrl_delay) {
        if (race2) k = r16 ? pick_ppo_roll2<vf::NetHover, 16, vf::VF_ENV_RACING2, true>(h->dyn.cfg) : pick_ppo_roll2<vf::NetHover, 32, vf::VF_ENV_RACING2, true>(h->dyn.cfg);
        else k = r16 ? pick_ppo_roll2<vf::NetHover, 16, VF_ENV_RACING, true>(h->dyn.cfg) : pick_ppo_roll2<vf::NetHover, 32, VF_ENV_RACING, true>(h->dyn.cfg);
    }
    const int rows = r16 ? 16 : 32;
    vf::EnvArgs ge{vf::DynArgs{N, h->dyn.G, h->dyn.g_drag, h->dyn.S, nullptr, nullptr, vf::ring_head(&h->dyn), nullptr, h->dyn.vel_strided},
                   *out, h->g_race, 1};
    ge.out.done_list = ge.out.done_count = nullptr;
    ge.out.obs = T > 1 ? a->obs_state + (size_t)N * OW : a->obs_final;
    vf::ChainArgs gc{*desc, params, packed, vf::ChainIo{{a->obs_state, a->obs_target}, a->means, a->values}, T * N, nullptr, nullptr, nullptr,
                     {nullptr, nullptr}};
    vf::PpoRollArgs r{T, N, reinterpret_cast<float4*>(a->actions), a->log_probs, a->rewards, a->episode_starts, a->last_starts,
                      a->obs_state, a->obs_final, a->log_std, a->noise_key, a->sample_step, a->obs_target_row, a->w1, a->capacity, a->cursor,
                      a->idx_list, a->rows0, a->rows1, a->stat};
    if (k) {
        hipLaunchKernelGGL(k, dim3((N + rows - 1) / rows), dim3(64), 0, vf::as_stream(stream), h->dyn.d_cfg, h->d_cfg, ge, gc, r);
        VF_HIP(hipGetLastError());
    } else {
        // a generated class: its roll-out plugin, compiled on first use for this env kind / dynamics configuration (visfly_amd/_jit.py)
        int rc = 0;
        // (only for the observation width the env kind's epilogue forms: a class padded to the same 16 input columns matches 13- and 16-wide rows alike)
        for (int i = 0; i < vf::chain_plugin_count() && rc == 0 && desc->in_dim[0] == OW; ++i) {
            const vf::ChainPlugin* p = vf::chain_plugin(i);
            if (p->ppo_rollout && p->rollout_abi == vf::kRolloutPluginAbi)
                rc = p->ppo_rollout(desc, race2 ? vf::VF_ENV_RACING2 : h->cfg.kind, &h->dyn.cfg, a->obs_target != nullptr, h->dyn.d_cfg, h->d_cfg,
                                    &ge, &gc, &r, N, vf::as_stream(stream));      // (the KERNEL-side kind: RacingEnv2's 16 columns are formed by the epilogue)
        }
        if (rc <= -1000) return vf::fail(VF_EHIP, "vf_ppo_rollout (chain plugin) failed: %s", hipGetErrorString((hipError_t)(-rc - 1000)));
        if (rc == 0)
            return vf::fail(VF_EUNSUPPORTED, "vf_ppo_rollout: no persistent roll-out for this network class / env kind / dynamics "
                                             "configuration (built in: [128, 64] x [64, 64] actor-critic, Hover / Navigation, thrust / bodyrate, "
                                             "Euler / RK4; generated classes: through their roll-out plugin)");
        vf::chain_plugin_count_launch();
    }
    h->dyn.tick += T;
    h->stale_all = 1;       // agents re-spawned inside the launch: the prefetched copies' stale bits no longer cover them
    return VF_OK;
}

#ifdef VF_PPO_TRACE
extern "C" int vf_debug_ppo_trace(long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vf::vf_ppo_trace), sizeof(long long) * 8, 0, hipMemcpyDeviceToHost);
}
#endif
