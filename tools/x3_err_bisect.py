"""GPU diagnostic: which product of the fp32x3 engine moves the encoder gradients of the cfg2 full-size case?
MMVAE_X3_OFF keeps single products on the fp32 matrix instruction."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402
from tests import gpu_util as U  # noqa: E402
A, B, D = 2, 5000, 5000
h = R.Hyper(input_dim=D, n_arm=A)
sd = R.init_state_dict(h, 546 + A)
x = R.synthetic_batch(B, D, seed=546 + D)
noise = R.draw_noise(h, B, seed=7 + A)
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
n64 = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in noise.items()}
_, _, g64 = R.grads_autograd(sd64, [x.double()] * A, h, n64)
for name, eng, off in (("fp32_mfma", "fp32_mfma", 0), ("fp32x3", "fp32x3", 0), ("x3 but fc1", "fp32x3", 1), ("x3 but fc11", "fp32x3", 2),
                       ("x3 but dW1", "fp32x3", 4), ("x3 but dW11", "fp32x3", 8), ("x3 only fc11", "fp32x3", 13), ("x3 again", "fp32x3", 0)):
    os.environ["MMVAE_X3_OFF"] = str(off)
    m = U.build_model(h, sd); m.train(); m.gemm_dtype = eng
    m.set_explicit_noise(U.noise_to_device(noise))
    m.fused_train_step(x.to(U.DEV).expand(A, -1, -1), 1.0, None, do_adam=False)
    torch.cuda.synchronize()
    g = {k: gv.detach().cpu().double() for (k, _), gv in zip(m.named_parameters(), m._grad_views)}
    del m
    worst = {}
    for k in g64:
        ref = g64[k].double(); sc = float(ref.abs().max()) + 1e-30
        worst[k] = float(((g[k] - ref).abs() / sc).max())
    enc = max(v for k, v in worst.items() if k.split(".")[0] in ("fc1", "fc2", "fc3", "fc4", "fc5"))
    dec = max(v for k, v in worst.items() if k.split(".")[0] in ("fc6", "fc7", "fc8", "fc9", "fc10", "fc11"))
    lat = max(v for k, v in worst.items() if k.split(".")[0] in ("fcc", "fc_mu", "fc_sigma"))
    print(f"{name:14s} worst entry error: encoder {enc:.2e}  latent heads {lat:.2e}  decoder {dec:.2e}", flush=True)
