"""register-chained forward (vf_mlp_chain.hip) vs the LDS forward: max deviation from a torch fp32 module and launch time.
usage: python tools/exp_chain.py [M ...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from visfly_amd.ppo import MlpPolicy
    DEV = "cuda:0"
    for net in ("nav", "hover"):
        for M in [int(a) for a in sys.argv[2:]]:
            if net == "nav":
                pol = MlpPolicy({"state": 13, "target": 3}, {"state": [128, 64], "target": [128, 64]}, [64, 64], [64, 64], DEV, seed=9)
                obs = {"state": torch.randn((M, 13), device=DEV), "target": torch.randn((M, 3), device=DEV)}
            else:
                pol = MlpPolicy({"state": 13}, {"state": [128, 64]}, [64, 64], [64, 64], DEV, seed=9)
                obs = {"state": torch.randn((M, 13), device=DEV)}
            pol.lazy_pack = True
            ref = pol.to_torch().to(DEV)
            with torch.no_grad():
                m0, v0 = ref(obs)
            for save in (False, True):
                m1, v1 = pol.forward(obs, save_activations=save)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    pol.forward(obs, save_activations=save)
                e1.record()
                torch.cuda.synchronize()
                err = max(float((m0 - m1).abs().max()), float((v0 - v1.view_as(v0)).abs().max()))
                print(f"chain={os.environ.get('VISFLY_AMD_MLP_CHAIN', '1')} {net} M={M} save={save}: {e0.elapsed_time(e1) * 20:.1f} us  max err {err:.2e}", flush=True)
    sys.exit(0)
for chain in ("1", "0"):
    subprocess.run([sys.executable, __file__, "--child"] + (sys.argv[1:] or ["25600", "32768", "16384", "1000"]),
                   env=dict(os.environ, VISFLY_AMD_MLP_CHAIN=chain))
