// vf_bptt_reverse.hip -- the reverse half of a BPTT horizon as ONE persistent launch (gfx950).
//
// loss.backward() over a horizon (utils/algorithms/BPTT.py:127-129) is a strictly serial chain: the adjoint of env step t needs
// dLoss/d obs_{t+1} from the policy's reverse pass of step t + 1, which needs dLoss/d action_{t+1} from the adjoint of step
// t + 1.  As separate launches that is 2 H launches of 256-512 waves, each latency-bound (~20 us at 16 384 agents).  Agents are
// independent, so here a wave owns ROWS = 16 or 32 agents (the rows-per-wave choice of vf_mlp_backward_data for N rows) for the
// whole sweep t = H-1 .. 0:
//   * adjoint of the control interval + env epilogue for its agents (env_step_bwd_agent, one lane per agent; lanes ROWS..63
//     replicate lane & (ROWS - 1) -- same loads, same arithmetic, same stores of the same values, so no special case);
//   * action head reverse + reverse register chain for the same rows (vf_mlp_chain_bwd.hpp): masked layer gradients into
//     the slot's dZ buffers for the horizon-wide weight-gradient launch, dLoss/d observation for the next adjoint step;
// d_action and the observation gradient travel through per-step rows of scratch (the same wave reads what it wrote).
// Bit-identical to the launch-by-launch sweep (tests/test_bptt_gpu.py).
#include "vf_chain_plugin.hpp"
#include "vf_bptt_reverse_kernel.hpp"

extern "C" int vf_bptt_reverse(vf_env* h, const vf_mlp_bwd_desc* desc, const float* packed, const float* log_std, const float* eps,
                               const float* actions, const float* tape, int64_t tape_stride, const uint8_t* tape_done,
                               const float* d_reward, float* adj_slab, float* d_action, const float* g_obs, float* g_log_std, int32_t H,
                               const float* substep_tape, const float* log_std_rows, vf_stream_t stream)
{
    if (!h || !desc || !packed || !eps || !actions || !tape || !tape_done || !d_reward || !adj_slab || !d_action || !g_obs || H <= 0)
        return vf::fail(VF_EINVAL, "vf_bptt_reverse: bad argument");
    if (!h->dyn.S) return vf::fail(VF_ESTATE, "vf_bptt_reverse: vf_env_bind has not been called");
    if (tape_stride < (int64_t)h->dyn.Npad * h->dyn.G * 4) return vf::fail(VF_EINVAL, "vf_bptt_reverse: tape rows are shorter than the slab");
    // rows the kernel reads / writes as float4
    auto misaligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
    if (misaligned(actions) || misaligned(d_action) || misaligned(eps) || misaligned(g_log_std) || misaligned(log_std_rows) || misaligned(tape) ||
        misaligned(adj_slab) || (tape_stride & 3))
        return vf::fail(VF_EINVAL, "vf_bptt_reverse: actions / d_action / eps / g_log_std / log_std_rows / tape / adj_slab must be 16-byte aligned (float4 rows)");
    if (h->dyn.wind) return vf::fail(VF_EUNSUPPORTED, "vf_bptt_reverse: per-agent wind rows are set");
    const int S = h->dyn.cfg.interval_steps;
    if (S > 10) return vf::fail(VF_EUNSUPPORTED, "vf_bptt_reverse: at most 10 sub-steps per control interval");
    const int N = h->dyn.N;
    if ((int64_t)H * N * 128 * 4 >= (1ll << 32))             // the reverse chain addresses its dZ stores with 32-bit byte offsets
        return vf::fail(VF_EUNSUPPORTED, "vf_bptt_reverse: H x N rows of layer gradients pass 4 GiB per buffer");
    // rows per wave: the choice vf_mlp_backward_data makes for N rows, so that the sweep equals the launch-by-launch one to the bit
    const int cls = vf::bwd_chain_policy_class(desc, N), net = cls & 15;
    // net == 0: not a built-in class -- a generated one runs from its BPTT plugin below (16 agents per wave with the sub-step tape only);
    // its caller says which head form it has by the rows it passes
    const bool r16 = net ? (cls & 16) != 0 : N <= 16384;
    const bool sac = net ? net >= 3 : log_std_rows != nullptr;          // td_policies.Actor: both head gradients come from d_action and the saved log_std rows
    if (sac ? !log_std_rows : (!log_std || !g_log_std))
        return vf::fail(VF_EINVAL, sac ? "vf_bptt_reverse: log_std_rows (H N, 4) is required for the two-headed actor classes"
                                       : "vf_bptt_reverse: log_std / g_log_std are required for the state-independent-log_std classes");
    // the tape is read only by the 16-agents-per-wave sweep (its records are the forward launch's waves); with 32 agents per wave
    // (N > 16 384 per launch) the interval is replayed -- same results to the bit
    // ... and with at most kRingRegs delay-ring slots (the sweep keeps the ring's adjoints in registers)
    const bool ckpt = substep_tape != nullptr && r16 && h->dyn.cfg.delay_steps <= vf::kRingRegs;
    if (reinterpret_cast<uintptr_t>(substep_tape) & 15) return vf::fail(VF_EINVAL, "vf_bptt_reverse: substep_tape must be 16-byte aligned");
    vf::RevKernel k = nullptr;
    const bool race2 = h->cfg.kind == VF_ENV_RACING && h->cfg.obs_mode == VF_OBS_RACE2;      // RacingEnv2: 16 gate-relative columns
    const int OW = race2 ? 16 : 13;
    if (net == 0) k = nullptr;
    else if (race2) k = vf::pick_rev_race2(net, r16, h->dyn.cfg, ckpt);
    else if ((net == 1 || net == 3) && h->cfg.kind == VF_ENV_NAV) k = vf::pick_rev_nav2(net, r16, h->dyn.cfg, ckpt);
    else if (!h->dyn.cfg.ctrl_delay) k = vf::pick_rev_nodelay(net, r16, h->cfg.kind, h->dyn.cfg, ckpt);
    else if (sac) k = r16 ? vf::pick_rev_sac(net, h->cfg.kind, h->dyn.cfg, ckpt) : vf::pick_rev_sac32(net, h->cfg.kind, h->dyn.cfg);
    else if (net == 1 && h->cfg.kind == VF_ENV_HOVER) k = r16 ? vf::pick_rev<vf::NetHover, 16, VF_ENV_HOVER>(h->dyn.cfg, ckpt) : vf::pick_rev<vf::NetHover, 32, VF_ENV_HOVER>(h->dyn.cfg, ckpt);
    else if (net == 1 && h->cfg.kind == VF_ENV_RACING) k = r16 ? vf::pick_rev<vf::NetHover, 16, VF_ENV_RACING>(h->dyn.cfg, ckpt) : vf::pick_rev<vf::NetHover, 32, VF_ENV_RACING>(h->dyn.cfg, ckpt);
    else if (net == 2 && h->cfg.kind == VF_ENV_NAV) k = r16 ? vf::pick_rev<vf::NetNav, 16, VF_ENV_NAV>(h->dyn.cfg, ckpt) : vf::pick_rev<vf::NetNav, 32, VF_ENV_NAV>(h->dyn.cfg, ckpt);
    if (!k && net != 0) return vf::fail(VF_EUNSUPPORTED, "vf_bptt_reverse: no persistent reverse sweep for this network class / env kind / dynamics configuration");
    if (net == 0 && !ckpt)
        return vf::fail(VF_EUNSUPPORTED, "vf_bptt_reverse: a generated actor class sweeps 16 agents per wave from the sub-step tape only (N <= 16 384, "
                                         "substep_tape of the forward launch, delay ring <= %d slots)", vf::kRingRegs);
    bool found = false;        // g_obs must be the (rows, 13) buffer the "state" branch's first layer writes its data gradient to
    for (int l = 0; l < desc->n_layers; ++l) found = found || (desc->layer[l].need_dx && desc->layer[l].dX == g_obs && desc->layer[l].ld_dx == OW);
    if (!found) return vf::fail(VF_EINVAL, "vf_bptt_reverse: g_obs is not the (rows, %d) observation-gradient buffer of the layer table", OW);
    vf::BwdArgsChain gb{*desc, packed, H * N, reinterpret_cast<const float4*>(d_action), reinterpret_cast<const float4*>(actions), log_std,
                        reinterpret_cast<const float4*>(eps), reinterpret_cast<float4*>(g_log_std), reinterpret_cast<const float4*>(log_std_rows),
                        VF_SAC_LOG_STD_MIN, VF_SAC_LOG_STD_MAX};
    vf::RevArgs r{H, N, h->dyn.G, h->dyn.g_drag, h->g_race, tape, tape_stride, reinterpret_cast<const float4*>(actions), tape_done, d_reward,
                  adj_slab, reinterpret_cast<float4*>(d_action), g_obs, reinterpret_cast<const float4*>(substep_tape)};
    const size_t lds = ckpt ? (size_t)2 * (S + 3) * 64 * sizeof(float4) + (64 + 256) * sizeof(float) : (size_t)S * vf::kSave * 64 * sizeof(float);
    const int rows = r16 ? 16 : 32;
    if (k) {
        hipLaunchKernelGGL(k, dim3((N + rows - 1) / rows), dim3(64), lds, vf::as_stream(stream), h->dyn.d_cfg, h->d_cfg, gb, r);
        VF_HIP(hipGetLastError());
        return VF_OK;
    }
    int rc = 0;
    const int kkind = race2 ? vf::VF_ENV_RACING2 : h->cfg.kind;
    for (int i = 0; i < vf::chain_plugin_count() && rc == 0; ++i) {
        const vf::ChainPlugin* p = vf::chain_plugin(i);
        if (p->bptt_reverse && p->bptt_rev_abi == vf::kBpttRevPluginAbi)
            rc = p->bptt_reverse(desc, kkind, &h->dyn.cfg, h->dyn.d_cfg, h->d_cfg, &gb, &r, N, lds, vf::as_stream(stream));
    }
    if (rc <= -1000) return vf::fail(VF_EHIP, "vf_bptt_reverse (chain plugin) failed: %s", hipGetErrorString((hipError_t)(-rc - 1000)));
    if (rc == 0)
        return vf::fail(VF_EUNSUPPORTED, "vf_bptt_reverse: the policy's layer table is not one of the built-in register-chained classes and no BPTT "
                                         "plugin of a generated class serves it under this env kind / dynamics configuration");
    vf::chain_plugin_count_launch();
    return VF_OK;
}

#ifdef VF_PPO_TRACE
extern "C" int vf_debug_rev_trace(long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vf::vf_rev_trace), sizeof(long long) * 8, 0, hipMemcpyDeviceToHost);
}
#endif
