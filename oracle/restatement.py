"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the cpl-mixVAE train step.

This is the *oracle* for the HIP path: a plain-torch (CPU) restatement of the
reference algorithm with an explicit-noise interface.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``distributed-vae_amd``) never does and fails loudly
when its HIP library is missing.

Pinned against the real reference (``/root/reference/mmidas/nn_model.py``) by
``tests/test_oracle_vs_reference.py`` in the build container and against the
committed fixtures in ``tests/golden/`` everywhere (fixtures are generated from
the reference itself by ``oracle/gen_golden.py``).

Each function cites the reference lines it restates (paths relative to the
reference root):

* encoder / BN ordering          mmidas/nn_model.py:263-269, BN defs :208-255
* double softmax                  mmidas/nn_model.py:337
* Gumbel-softmax                  mmidas/nn_model.py:430-493
* state head + reparameterise     mmidas/nn_model.py:271-275, :347-351, :413-428
* decoder                         mmidas/nn_model.py:277-287
* loss                            mmidas/nn_model.py:495-598, helpers :39-86
* step driver / Adam              mmidas/cpl_mixvae.py:434-463, :274

Two backward paths are offered: ``autograd`` (what the reference does) and
``manual_backward`` (the analytic derivation the HIP kernels implement, stage
for stage, so a failing GPU test can be localised to a kernel).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# state_dict layer order of one arm, as the reference registers it
# (mmidas/nn_model.py:184-203).
LINEAR_NAMES = [
    "fc1", "fc2", "fc3", "fc4", "fc5", "fcc", "fc_mu", "fc_sigma",
    "fc6", "fc7", "fc8", "fc9", "fc10", "fc11",
]
BN_NAMES = ["batch_l1", "batch_l2", "batch_l3", "batch_l4", "batch_l5", "batch_s"]


@dataclass
class Hyper:
    """Hyper-parameters of ``mixVAE_model.__init__`` (nn_model.py:112-134)."""

    input_dim: int = 5000
    fc_dim: int = 100
    n_categories: int = 92
    state_dim: int = 2
    lowD_dim: int = 10
    x_drop: float = 0.5
    s_drop: float = 0.0
    n_arm: int = 2
    lam: float = 1.0
    tau: float = 0.005
    beta: float = 1.0
    hard: bool = False
    eps: float = 1e-8
    momentum: float = 0.01
    temp: float = 1.0

    def linear_shapes(self) -> Dict[str, tuple]:
        D, H, L, C, S = (self.input_dim, self.fc_dim, self.lowD_dim,
                         self.n_categories, self.state_dim)
        return {
            "fc1": (H, D), "fc2": (H, H), "fc3": (H, H), "fc4": (H, H), "fc5": (L, H),
            "fcc": (C, L), "fc_mu": (S, L + C), "fc_sigma": (S, L + C),
            "fc6": (L, S + C), "fc7": (H, L), "fc8": (H, H), "fc9": (H, H),
            "fc10": (H, H), "fc11": (D, H),
        }

    def bn_dims(self) -> Dict[str, int]:
        H, L, S = self.fc_dim, self.lowD_dim, self.state_dim
        return {"batch_l1": H, "batch_l2": H, "batch_l3": H, "batch_l4": H,
                "batch_l5": L, "batch_s": S}


def init_state_dict(h: Hyper, seed: int, dtype=torch.float32) -> Dict[str, Tensor]:
    """Parameters drawn exactly as the reference constructor draws them.

    nn_model.py:184-203 builds, for each layer name in LINEAR_NAMES order, a
    ModuleList of ``n_arm`` ``nn.Linear`` (default kaiming-uniform(a=sqrt 5)
    weight then U(+-1/sqrt(fan_in)) bias).  Re-creating ``nn.Linear`` objects in
    the same order under the same ``torch.manual_seed`` gives identical values.
    """
    torch.manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    shapes = h.linear_shapes()
    for name in LINEAR_NAMES:
        out_f, in_f = shapes[name]
        for a in range(h.n_arm):
            lin = torch.nn.Linear(in_f, out_f)
            sd[f"{name}.{a}.weight"] = lin.weight.detach().to(dtype).clone()
            sd[f"{name}.{a}.bias"] = lin.bias.detach().to(dtype).clone()
    for name, n in h.bn_dims().items():
        for a in range(h.n_arm):
            sd[f"{name}.{a}.running_mean"] = torch.zeros(n, dtype=dtype)
            sd[f"{name}.{a}.running_var"] = torch.ones(n, dtype=dtype)
            sd[f"{name}.{a}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return sd


def param_keys(h: Hyper) -> List[str]:
    """Order of ``model.parameters()`` (== state_dict order of the Linear entries)."""
    keys = []
    for name in LINEAR_NAMES:
        for a in range(h.n_arm):
            keys += [f"{name}.{a}.weight", f"{name}.{a}.bias"]
    return keys


def synthetic_batch(n: int, d: int, seed: int = 546, dtype=torch.float32) -> Tensor:
    """"synthetic-10x-v1" input of SURVEY.md section 8(d) / BASELINE.md section 3."""
    g = torch.Generator("cpu").manual_seed(seed)
    x = (torch.rand(n, d, generator=g) < 0.2).float() * torch.randn(n, d, generator=g).abs() * 3.0
    return x.to(dtype)


def draw_noise(h: Hyper, batch: int, seed: int, training: bool = True, eval_flag: bool = False):
    """Explicit noise in the reference's consumption order.

    Values are independent of the reference's global-generator stream (a GPU
    kernel cannot replay mt19937); parity tests feed the *same* explicit noise
    to both sides.
    """
    g = torch.Generator("cpu").manual_seed(seed)
    A, D, C, S = h.n_arm, h.input_dim, h.n_categories, h.state_dim
    noise = {"x_mask": [], "u_gumbel": [], "u_state": [], "s_mask": []}
    for _ in range(A):
        if training and h.x_drop > 0:
            noise["x_mask"].append((torch.rand(batch, D, generator=g) >= h.x_drop).to(torch.uint8))
        if not eval_flag:
            noise["u_gumbel"].append(torch.rand(batch, C, generator=g))
        noise["u_state"].append(torch.rand(batch, S, generator=g))
        if training and h.s_drop > 0:
            noise["s_mask"].append((torch.rand(batch, S, generator=g) >= h.s_drop).to(torch.uint8))
    return noise


# --------------------------------------------------------------------------- forward

def _bn(r: Tensor, sd, key: str, h: Hyper, training: bool, update: bool):
    """BatchNorm1d(affine=False, eps=h.eps, momentum=h.momentum) -- nn_model.py:208-255."""
    if training:
        mean = r.mean(0)
        var_b = r.var(0, unbiased=False)
        if update:
            n = r.shape[0]
            var_u = var_b * (n / max(n - 1, 1))
            sd[key + ".running_mean"] = (1 - h.momentum) * sd[key + ".running_mean"] + h.momentum * mean.detach()
            sd[key + ".running_var"] = (1 - h.momentum) * sd[key + ".running_var"] + h.momentum * var_u.detach()
            sd[key + ".num_batches_tracked"] = sd[key + ".num_batches_tracked"] + 1
    else:
        mean = sd[key + ".running_mean"]
        var_b = sd[key + ".running_var"]
    rstd = 1.0 / torch.sqrt(var_b + h.eps)
    return (r - mean) * rstd, mean, rstd


def forward(sd: Dict[str, Tensor], xs: Sequence[Tensor], h: Hyper, noise, *,
            temp: Optional[float] = None, training: bool = True, eval_flag: bool = False,
            update_running: bool = True, keep: bool = False, relu_override=None, mask=None):
    """``mixVAE_model.forward`` (nn_model.py:297-368) with explicit noise.

    Returns the reference's 10-tuple; with ``keep`` also a per-arm dict of the
    intermediates the HIP kernels save for backward (and the pre-activation
    ``z<site>`` of every ReLU, sites ``r1..r5``, ``d6..d10``, ``x_rec``).

    ``relu_override``: ``{(arm, site): bool tensor}`` -- decisions forced on the
    ReLU of that site (``relu(z)`` becomes ``z * mask``, so the gradient flows
    exactly where the mask says).  Test infrastructure for full-size parity: a
    pre-activation within fp32 rounding of zero is decided either way by ANY
    fp32 evaluation, and ONE differing decision moves every bias gradient below
    it; a test that has read the device's decisions evaluates the reference
    arithmetic (nn_model.py:263-287) on exactly those decisions and can then
    hold every tensor to the tight gate.

    ``mask``: indices of the kept categories (the pruning-time forward, nn_model.py:332-335): the second softmax runs
    over ``c_prob[:, mask]`` and ``c`` is zero elsewhere.
    """
    temp = h.temp if temp is None else temp
    eps = h.eps
    if mask is not None:
        mask = torch.as_tensor(mask, dtype=torch.long)

    def _relu(z, a, site, iv):
        if keep:
            iv["z" + site] = z.detach()
        if relu_override is not None and (a, site) in relu_override:
            return z * relu_override[(a, site)].to(z.dtype)
        return F.relu(z)
    x_recs, x_lows, cs, s_smps, c_smps, s_means, s_logvars, c_probs = ([] for _ in range(8))
    saved = []
    for a, x in enumerate(xs):
        W = lambda n: sd[f"{n}.{a}.weight"]
        b = lambda n: sd[f"{n}.{a}.bias"]
        iv = {}
        # encoder, nn_model.py:263-269 (dropout -> Linear -> ReLU -> BN, five times)
        if training and h.x_drop > 0:
            xt = x * noise["x_mask"][a].to(x.dtype) / (1.0 - h.x_drop)
        else:
            xt = x
        hcur = xt
        for i, name in enumerate(["fc1", "fc2", "fc3", "fc4", "fc5"], start=1):
            r = _relu(F.linear(hcur, W(name), b(name)), a, f"r{i}", iv)
            hcur, mean, rstd = _bn(r, sd, f"batch_l{i}.{a}", h, training, update_running)
            iv[f"r{i}"], iv[f"mean{i}"], iv[f"rstd{i}"] = r, mean, rstd
        x_low = hcur
        zc = F.linear(x_low, W("fcc"), b("fcc"))
        c_prob = F.softmax(zc, dim=-1)
        # nn_model.py:332-337
        if mask is not None:
            c = torch.zeros_like(c_prob).index_copy(1, mask, F.softmax(c_prob[:, mask] / h.tau, dim=-1))
        else:
            c = F.softmax(c_prob / h.tau, dim=-1)
        # nn_model.py:339-345 / :430-493
        if eval_flag:
            y_soft = c
            hard = True
        else:
            U = noise["u_gumbel"][a].to(x.dtype)
            g = -torch.log(-torch.log(U + eps) + eps)
            y_soft = F.softmax((torch.log(c + eps) + g) / temp, dim=-1)
            hard = h.hard
        if hard:
            ind = y_soft.argmax(dim=-1, keepdim=True)
            y_hard = torch.zeros_like(y_soft).scatter_(1, ind, 1.0)
            c_smp = (y_hard - y_soft).detach() + y_soft
        else:
            c_smp = y_soft
        # nn_model.py:347-351
        y = torch.cat((x_low, c_smp), dim=1)
        s_mean = F.linear(y, W("fc_mu"), b("fc_mu"))
        s_var = torch.sigmoid(F.linear(y, W("fc_sigma"), b("fc_sigma")))
        s_logvar = torch.log(s_var + eps)
        # nn_model.py:426-428 -- uniform noise, as the reference draws it
        s_smp = noise["u_state"][a].to(x.dtype) * torch.sqrt(torch.exp(s_logvar)) + s_mean
        # nn_model.py:277-287
        if training and h.s_drop > 0:
            s_in = s_smp * noise["s_mask"][a].to(x.dtype) / (1.0 - h.s_drop)
        else:
            s_in = s_smp
        z = torch.cat((c_smp, s_in), dim=1)
        d = z
        for i, name in enumerate(["fc6", "fc7", "fc8", "fc9", "fc10"], start=6):
            d = _relu(F.linear(d, W(name), b(name)), a, f"d{i}", iv)
            iv[f"d{i}"] = d
        x_rec = _relu(F.linear(d, W("fc11"), b("fc11")), a, "x_rec", iv)

        x_recs.append(x_rec); x_lows.append(x_low); cs.append(c); s_smps.append(s_smp)
        c_smps.append(c_smp); s_means.append(s_mean); s_logvars.append(s_logvar); c_probs.append(c_prob)
        if keep:
            iv.update(xt=xt, zc=zc, y_soft=y_soft, y=y, s_var=s_var, z=z)
            saved.append(iv)
    out = (x_recs, [], [], x_lows, cs, s_smps, c_smps, s_means, s_logvars, c_probs)
    return (out, saved) if keep else out


# --------------------------------------------------------------------------- loss

def loss(out, xs: Sequence[Tensor], h: Hyper):
    """``mixVAE_model.loss`` (nn_model.py:495-598), MSE mode, variational, no ref prior.

    Returns the reference's 9-tuple: (total, loss_recs[detached tensor], loss_joint,
    mean neg-joint-entropy, mean simplex distance, mean l2 distance, [kl_a], [], [ll_a]).
    """
    x_recs, _, _, _, cs, _, c_smps, s_means, s_logvars, _ = out
    A, C, eps = h.n_arm, h.n_categories, h.eps
    B = xs[0].shape[0]
    lls, loss_recs, loss_inds, kls = [], [], [], []
    c_ents, c_l2, c_dists = [], [], []
    logc = [torch.log(c + eps) for c in cs]
    # nn_model.py:75-77 -- unbiased batch variance, differentiable
    ivar = [torch.sqrt(1.0 / (c.var(0) + eps)) for c in cs]
    for a in range(A):
        x, xr = xs[a], x_recs[a]
        se = ((xr - x) ** 2).sum()
        lls.append(se / x.numel() + B * math.log(2 * math.pi))              # :542
        mism = ((xr > 0.1) != (x > 0.1)).to(x.dtype).mean()
        # :544-546 -- BCE of two {0,1} tensors = 100 * mismatch fraction (torch clamps log at -100)
        rec = 0.5 * se / B + 0.5 * (100.0 * mism)
        kl = (-0.5 * torch.mean(1 + s_logvars[a] - s_means[a] ** 2 - torch.exp(s_logvars[a]), dim=0)).sum()
        kls.append(kl); loss_recs.append(rec); loss_inds.append(rec + h.beta * kl)
        for b in range(a + 1, A):
            ent = (cs[a] * logc[a]).sum(-1).mean() + (cs[b] * logc[b]).sum(-1).mean()    # :565
            c_ents.append(ent)
            c_l2.append(((c_smps[a] - c_smps[b]) ** 2).sum(-1).mean())                    # :566
            c_dists.append(((logc[a] * ivar[a] - logc[b] * ivar[b]) ** 2).sum(-1).mean())  # :567-569
    n_pairs = max(A * (A - 1) / 2, 1)
    if not c_ents:
        raise ZeroDivisionError("n_arm == 1: reference loss divides by len([]) (nn_model.py:592)")
    joint = (h.lam * sum(c_dists) + sum(c_ents)
             + n_pairs * ((C / 2) * math.log(2 * math.pi) - 0.5 * math.log(2 * h.lam)))
    total = max(A - 1, 1) * sum(loss_inds) + joint
    return (total, torch.stack([r.detach() for r in loss_recs]), joint,
            sum(c_ents) / len(c_ents), sum(c_dists) / len(c_dists),
            sum(c_l2) / len(c_l2), kls, [], lls)


# --------------------------------------------------------------------------- step

def grads_autograd(sd, xs, h: Hyper, noise, **fw):
    """forward + loss + ``backward()`` as cpl_mixvae.py:434-462; returns (loss tuple, grads)."""
    keys = param_keys(h)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaves)
    out = forward(work, xs, h, noise, **fw)
    lt = loss(out, xs, h)
    gs = torch.autograd.grad(lt[0], [leaves[k] for k in keys])
    for k in sd:  # running stats updated by forward
        if k not in leaves:
            sd[k] = work[k]
    return out, lt, dict(zip(keys, gs))


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, t: int, lr=1e-3, b1=0.9, b2=0.999,
              eps=1e-8, weight_decay=0.0, decoupled=False):
    """torch.optim.Adam / AdamW single-tensor update (cpl_mixvae.py:274, train.py:144-147)."""
    if weight_decay != 0.0:
        if decoupled:
            p = p * (1 - lr * weight_decay)
        else:
            g = g + weight_decay * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def train_steps(sd, batches, h: Hyper, noises, lr=1e-3, opt_state=None, weight_decay=0.0,
                decoupled=False, backward="autograd"):
    """k optimiser steps; returns per-step loss tuples and the Adam state."""
    keys = param_keys(h)
    if opt_state is None:
        opt_state = {"t": 0, "m": {k: torch.zeros_like(sd[k]) for k in keys},
                     "v": {k: torch.zeros_like(sd[k]) for k in keys}}
    hist = []
    for x, nz in zip(batches, noises):
        xs = [x] * h.n_arm                                    # cpl_mixvae.py:425
        if backward == "autograd":
            _, lt, gs = grads_autograd(sd, xs, h, nz)
        else:
            _, lt, gs, _ = grads_manual(sd, xs, h, nz)
        opt_state["t"] += 1
        for k in keys:
            sd[k], opt_state["m"][k], opt_state["v"][k] = adam_step(
                sd[k], gs[k], opt_state["m"][k], opt_state["v"][k], opt_state["t"], lr,
                weight_decay=weight_decay, decoupled=decoupled)
        hist.append(lt)
    return hist, opt_state


# --------------------------------------------------------------------------- analytic backward

def _bn_bwd(g: Tensor, r: Tensor, mean: Tensor, rstd: Tensor) -> Tensor:
    """d/dr of ((r-mean)*rstd) with batch statistics (biased variance)."""
    xh = (r - mean) * rstd
    return rstd * (g - g.mean(0) - xh * (g * xh).mean(0))


def grads_manual(sd, xs, h: Hyper, noise, **fw):
    """Analytic backward, stage for stage as the HIP kernels compute it.

    Training mode only (batch statistics).  Returns (out, loss tuple, grads, stages)
    where ``stages`` holds every intermediate gradient for kernel-level debugging.
    """
    with torch.no_grad():
        out, saved = forward(sd, xs, h, noise, keep=True, **fw)
        lt = loss(out, xs, h)
        x_recs, _, _, x_lows, cs, s_smps, c_smps, s_means, s_logvars, c_probs = out
        A, eps, B = h.n_arm, h.eps, xs[0].shape[0]
        ovr = fw.get("relu_override") or {}

        def on(a_, site, val):   # ReLU decisions: the forced ones where given, else the sign of the value
            return (ovr[(a_, site)] if (a_, site) in ovr else (val > 0)).to(val.dtype)
        temp = fw.get("temp") or h.temp
        am1 = float(max(A - 1, 1))
        L, C, S = h.lowD_dim, h.n_categories, h.state_dim
        # ---- coupling: column statistics and pair sums (nn_model.py:558-569)
        logc = [torch.log(c + eps) for c in cs]
        cmean = [c.mean(0) for c in cs]
        cvar = [c.var(0) for c in cs]
        iv = [1.0 / torch.sqrt(v + eps) for v in cvar]
        u = [logc[a] * iv[a] for a in range(A)]
        usum = sum(u)
        grads: Dict[str, Tensor] = {}
        stages = []
        for a in range(A):
            st = {}
            sv = saved[a]
            W = lambda n: sd[f"{n}.{a}.weight"]
            x = xs[a]
            # ---- reconstruction: d total / d x_rec = (A-1)' (x_rec - x)/B, through ReLU
            gz11 = am1 * (x_recs[a] - x) / B * on(a, "x_rec", x_recs[a])
            grads[f"fc11.{a}.weight"] = gz11.t() @ sv["d10"]
            grads[f"fc11.{a}.bias"] = gz11.sum(0)
            gd = gz11 @ W("fc11")
            st["gz11"] = gz11
            # ---- decoder fc10..fc6
            for i in (10, 9, 8, 7, 6):
                name = f"fc{i}"
                dz = gd * on(a, f"d{i}", sv[f"d{i}"])
                xin = sv[f"d{i-1}"] if i > 6 else sv["z"]
                grads[f"{name}.{a}.weight"] = dz.t() @ xin
                grads[f"{name}.{a}.bias"] = dz.sum(0)
                gd = dz @ W(name)
                st[f"dz{i}"] = dz
            gzin = gd                                           # [B, C+S]
            g_csmp = gzin[:, :C].clone()
            gs_in = gzin[:, C:]
            if h.s_drop > 0 and fw.get("training", True):
                gs = gs_in * noise["s_mask"][a].to(x.dtype) / (1.0 - h.s_drop)
            else:
                gs = gs_in
            # ---- state head (nn_model.py:347-351, :426-428; kl :43-44)
            mu, lv, var = s_means[a], s_logvars[a], sv["s_var"]
            Us = noise["u_state"][a].to(x.dtype)
            gmu = gs + am1 * h.beta * mu / B
            glv = gs * Us * 0.5 * torch.sqrt(torch.exp(lv)) + am1 * h.beta * (-0.5 / B) * (1 - torch.exp(lv))
            gvar = glv / (var + eps)
            gsig = gvar * var * (1 - var)
            gms = torch.cat((gmu, gsig), dim=1)                 # [B, 2S]
            Wms = torch.cat((W("fc_mu"), W("fc_sigma")), dim=0)  # [2S, L+C]
            grads[f"fc_mu.{a}.weight"] = gmu.t() @ sv["y"]
            grads[f"fc_mu.{a}.bias"] = gmu.sum(0)
            grads[f"fc_sigma.{a}.weight"] = gsig.t() @ sv["y"]
            grads[f"fc_sigma.{a}.bias"] = gsig.sum(0)
            gy = gms @ Wms                                       # [B, L+C]
            g_csmp = g_csmp + gy[:, L:]
            g_xlow = gy[:, :L].clone()
            st.update(gzin=gzin, gms=gms, gy=gy)
            # ---- Gumbel-softmax backward (soft sample; straight-through when hard)
            ys = sv["y_soft"]
            c = cs[a]
            if fw.get("eval_flag", False):
                gc = g_csmp.clone()                              # y_soft == c, no noise
            else:
                glg = ys * (g_csmp - (ys * g_csmp).sum(-1, keepdim=True)) / temp
                gc = glg / (c + eps)
            # ---- coupling terms on c (entropy, distance, variance path)
            gc = gc + (A - 1) * (logc[a] + c / (c + eps)) / B
            G = (2.0 * h.lam / B) * (A * u[a] - usum)            # d joint / d u_a
            T = (G * logc[a]).sum(0)                             # d joint / d iv_a  [C]
            gc = gc + G * iv[a] / (c + eps)
            gc = gc + (T * (-0.5) * iv[a] ** 3) * 2.0 * (c - cmean[a]) / (B - 1)
            st.update(G=G, T=T, gc=gc)
            # ---- double softmax backward (nn_model.py:337, :269)
            gq = c * (gc - (c * gc).sum(-1, keepdim=True)) / h.tau
            cp = c_probs[a]
            gzc = cp * (gq - (cp * gq).sum(-1, keepdim=True))
            grads[f"fcc.{a}.weight"] = gzc.t() @ x_lows[a]
            grads[f"fcc.{a}.bias"] = gzc.sum(0)
            g_h = g_xlow + gzc @ W("fcc")                        # grad wrt BN5 output
            st.update(gzc=gzc, g5=g_h)
            # ---- encoder fc5..fc1 with BatchNorm backward
            for i in (5, 4, 3, 2, 1):
                name = f"fc{i}"
                r = sv[f"r{i}"]
                gr = _bn_bwd(g_h, r, sv[f"mean{i}"], sv[f"rstd{i}"])
                dz = gr * on(a, f"r{i}", r)
                if i > 1:
                    rp = sv[f"r{i-1}"]
                    xin = (rp - sv[f"mean{i-1}"]) * sv[f"rstd{i-1}"]
                else:
                    xin = sv["xt"]
                grads[f"{name}.{a}.weight"] = dz.t() @ xin
                grads[f"{name}.{a}.bias"] = dz.sum(0)
                st[f"dz{i}"] = dz
                if i > 1:
                    g_h = dz @ W(name)
            stages.append(st)
    return out, lt, grads, stages
