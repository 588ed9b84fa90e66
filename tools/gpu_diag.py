"""GPU diagnostic (not a test): HIP path vs the oracle's staged analytic backward on the golden
cases, every intermediate, one table.  Run on the GPU box: python tools/gpu_diag.py [case ...]"""
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402
from tests import golden_util as G  # noqa: E402
from tests import gpu_util as U  # noqa: E402


def rel(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    if a.shape != b.shape:
        return float("nan")
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def diag(name, h, sd, x, noise, ref64=False):
    print(f"==== {name}: A={h.n_arm} B={x.shape[0]} D={h.input_dim} H={h.fc_dim} L={h.lowD_dim} C={h.n_categories} "
          f"S={h.state_dim} hard={h.hard} s_drop={h.s_drop}", flush=True)
    A = h.n_arm
    sd_ref = {k: v.clone() for k, v in sd.items()}
    if ref64:   # reference in fp64: shows the HIP path's own rounding error
        sd_ref = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        n64 = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in noise.items()}
        out_r, lt_r, g_r, st = R.grads_manual(sd_ref, [x.double()] * A, h, n64)
        _, saved = R.forward({k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()},
                             [x.double()] * A, h, n64, keep=True, update_running=False)
    else:
        out_r, lt_r, g_r, st = R.grads_manual(sd_ref, [x] * A, h, noise)
        _, saved = R.forward({k: v.clone() for k, v in sd.items()}, [x] * A, h, noise, keep=True, update_running=False)
    m = U.build_model(h, sd)
    m.train()
    out, lt, grads = U.run_step(m, x.to(U.DEV), noise)
    H, L, C, S = h.fc_dim, h.lowD_dim, h.n_categories, h.state_dim
    rows = []
    def add(tag, got, ref):
        rows.append((tag, rel(got, ref)))
    for i in range(1, 5):
        add(f"r{i}", U.ws(m, f"r{i}", H), torch.stack([s[f"r{i}"] for s in saved]))
    add("r5", U.ws(m, "r5", L), torch.stack([s["r5"] for s in saved]))
    add("bn_mean1", m._engine.ws_raw("bn_mean1", A * H).cpu().view(A, H), torch.stack([s["mean1"] for s in saved]))
    names = {3: "x_low", 9: "c_prob", 4: "c", 6: "c_smp", 7: "s_mean", 8: "s_logvar", 5: "s_smp", 0: "x_rec"}
    for i, nm in names.items():
        add(nm, torch.stack([t.cpu() for t in out[i]]), torch.stack(list(out_r[i])))
    add("y_soft", U.ws(m, "y_soft", C), torch.stack([s["y_soft"] for s in saved]))
    add("zin", U.ws(m, "zin", C + S), torch.stack([s["z"] for s in saved]))
    add("d6", U.ws(m, "d6", L), torch.stack([s["d6"] for s in saved]))
    for i in range(7, 11):
        add(f"d{i}", U.ws(m, f"d{i}", H), torch.stack([s[f"d{i}"] for s in saved]))
    lnames = ["total", "rec", "joint", "c_ent", "c_dist", "c_l2"]
    for i, nm in enumerate(lnames):
        add("loss/" + nm, torch.as_tensor(lt[i]).cpu(), torch.as_tensor(lt_r[i]))
    add("loss/kl", torch.stack([t.cpu() for t in lt[6]]), torch.stack(lt_r[6]))
    add("loss/ll", torch.stack([t.cpu() for t in lt[8]]), torch.stack(lt_r[8]))
    add("dz11", U.ws(m, "dz11", h.input_dim), torch.stack([s["gz11"] for s in st]))
    add("gzin", U.ws(m, "gzin", C + S), torch.stack([s["gzin"] for s in st]))
    add("gzc", U.ws(m, "gzc", C), torch.stack([s["gzc"] for s in st]))
    add("g5", U.ws(m, "g5", L), torch.stack([s["g5"] for s in st]))
    add("dz1", U.ws(m, "dz1", H), torch.stack([s["dz1"] for s in st]))
    for k in R.param_keys(h):
        add("grad/" + k, grads[k], g_r[k])
    for k in sd_ref:
        if "running" in k:
            add("bn/" + k, m.state_dict()[k].cpu(), sd_ref[k])
    worst = 0.0
    for tag, r in rows:
        flag = "" if r < 1e-3 else "   <<<<<<"
        worst = max(worst, r if r == r else 1e9)
        print(f"  {tag:28s} {r:10.3e}{flag}")
    print(f"  worst {worst:.3e}", flush=True)
    return worst


def main():
    cases = sys.argv[1:] or G.SMALL_CASES
    worst = 0.0
    for name in cases:
        try:
            if name == "full64":
                h = R.Hyper()
                sd = R.init_state_dict(h, 546)
                x = R.synthetic_batch(5000, h.input_dim)
                noise = R.draw_noise(h, 5000, seed=7)
                worst = max(worst, diag(name, h, sd, x, noise, ref64=True))
                continue
            if name == "mid":
                g = G.load("mid_a2")
                h = G.hyper_of(g)
                sd = R.init_state_dict(h, int(g["seed"]))
                x = R.synthetic_batch(G.batch_of(g), h.input_dim)
                noise = R.draw_noise(h, G.batch_of(g), seed=int(g["noise_seed"]))
            else:
                g = G.load(name)
                h = G.hyper_of(g)
                sd = G.state_dict_of(g)
                x = torch.from_numpy(g["x"])
                noise = G.noise_of(g)
            worst = max(worst, diag(name, h, sd, x, noise))
        except Exception:
            traceback.print_exc()
            worst = 1e9
    print("OVERALL WORST", worst)


if __name__ == "__main__":
    main()
