"""GPU diagnostic: full-size cfg2 step with the batch sums through the fixed-point accumulators against the partial
arrays (MMVAE_BN_PARTIALS=1), both engines: run-to-run bit-identity, distance of every gradient from the fp64 oracle,
and the hidden units whose ReLU decision differs between the two schemes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402
A, B, D = 2, 5000, 5000
h = R.Hyper(input_dim=D, n_arm=A)
seed = int(os.environ.get("SEED", "546"))
sd = R.init_state_dict(h, seed)
x = R.synthetic_batch(B, D, seed=seed + 1)
noise = R.draw_noise(h, B, seed=seed + 2)
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
n64 = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in noise.items()}
_, _, g64 = R.grads_autograd(sd64, [x.double()] * A, h, n64)
acts = {}
for eng in ("fp32x3", "fp32_mfma"):
    for part in ("0", "1"):
        os.environ["MMVAE_BN_PARTIALS"] = part
        runs = []
        for rep in range(2):
            m = U.build_model(h, sd); m.train(); m.gemm_dtype = eng
            m.set_explicit_noise(U.noise_to_device(noise))
            buf = m.fused_train_step(x.to(U.DEV).expand(A, -1, -1), 1.0, None, do_adam=False).clone()
            torch.cuda.synchronize()
            g = {k: gv.detach().clone() for (k, _), gv in zip(m.named_parameters(), m._grad_views)}
            e = m._engine
            act = {n: e.ws_view(n, h.fc_dim).clone() for n in ("r1",)}
            runs.append((buf, g, act))
            del m
        same = all(torch.equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1]) and torch.equal(runs[0][0], runs[1][0])
        g = {k: v.cpu().double() for k, v in runs[0][1].items()}
        acts[eng, part] = runs[0][2]
        line = []
        for k in g64:
            ref = g64[k].double(); sc = float(ref.abs().max()) + 1e-300
            q = float(torch.quantile(((g[k] - ref).abs() / sc).flatten()[:4_000_000], 0.9))
            line.append((k, q))
        worst = sorted(line, key=lambda t: -t[1])[:4]
        print(f"{eng:10s} partials={part} bit-identical runs: {same}   worst p90: " + "  ".join(f"{k} {q:.1e}" for k, q in worst), flush=True)
for eng in ("fp32x3", "fp32_mfma"):
    a0, a1 = acts[eng, "0"]["r1"], acts[eng, "1"]["r1"]
    print(eng, "r1 max |acc - partials|:", float((a0 - a1).abs().max()), " ReLU decisions that differ:", int(((a0 > 0) != (a1 > 0)).sum()))
