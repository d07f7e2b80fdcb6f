"""register-chained backward (reverse chain + row-slab weight gradients) vs torch autograd and vs the block-tile kernel
usage: python tools/exp_chain_bwd.py [M ...]   (VISFLY_AMD_MLP_CHAIN=0 runs the block-tile kernel)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from visfly_amd.ppo import MlpPolicy
    DEV = "cuda:0"
    for net, mode in (("nav", "ppo"), ("hover", "ppo"), ("nav", "bptt"), ("hover", "bptt")):
        for M in [int(a) for a in sys.argv[2:]]:
            dims = {"state": 13, "target": 3} if net == "nav" else {"state": 13}
            pol = MlpPolicy(dims, {k: [128, 64] for k in dims}, [64, 64], [64, 64], DEV, seed=9)
            pol.lazy_pack = True
            g = torch.Generator(device=DEV).manual_seed(M)
            obs = {k: torch.randn((M, d), device=DEV, generator=g) for k, d in dims.items()}
            d_mean = torch.randn((M, 4), device=DEV, generator=g) / M
            d_value = torch.randn(M, device=DEV, generator=g) / M if mode == "ppo" else None
            ig = mode == "bptt"
            ref = pol.to_torch().to(DEV)
            xs = {k: v.clone().requires_grad_(ig) for k, v in obs.items()}
            m0, v0 = ref(xs)
            loss = (m0 * d_mean).sum() + ((v0.view(-1) * d_value).sum() if d_value is not None else 0.0) + 0.0 * ref.log_std.sum()
            loss.backward()
            pol.grad.zero_()
            pol.forward(obs)
            d_in = pol.backward(d_mean, d_value, None, need_input_grad=ig)
            torch.cuda.synchronize()
            for mod in ref.lin:
                for prm in mod.parameters():
                    if prm.grad is None:
                        prm.grad = torch.zeros_like(prm)
            gref = ref.flat_grad().to(DEV)
            gref[pol.log_std_off:] = 0
            live = gref != 0 if d_value is None else torch.ones_like(gref, dtype=torch.bool)
            err = float((pol.grad - gref)[live].abs().max() / gref.abs().max())
            ein = max([float((d_in[k] - xs[k].grad).abs().max() / xs[k].grad.abs().max()) for k in d_in] or [0.0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                pol.backward(d_mean, d_value, None, need_input_grad=ig)
            e1.record()
            torch.cuda.synchronize()
            print(f"chain={os.environ.get('VISFLY_AMD_MLP_CHAIN', '1')} {net} {mode} M={M}: bwd {e0.elapsed_time(e1) / 30 * 1e3:.1f} us  "
                  f"rel err dW {err:.2e}  d_obs {ein:.2e}", flush=True)
    sys.exit(0)
for chain in ("1", "0"):
    subprocess.run([sys.executable, __file__, "--child"] + (sys.argv[1:] or ["25600", "16384", "777", "1"]),
                   env=dict(os.environ, VISFLY_AMD_MLP_CHAIN=chain))
