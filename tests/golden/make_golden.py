#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small input/output vectors from an INDEPENDENT
implementation (PyTorch-CPU 2.10 in the authoring container, float64 numpy for
Adam) for the parts of the hot path the reference's own tests do not pin
(SURVEY.md section 4 "Not pinned by any reference test": conv2d, pools,
transpose, add_broadcast, sigmoid, Adam).  Mappings:
  reference conv  == F.conv2d(x, w.flatten().reshape(9*C_in, C_out).T.reshape(C_out, C_in, 3, 3), b, padding=1)   (Q3)
  max_pool2d      == F.max_pool2d(return_indices=True); reference index = plane base + torch's per-plane index
  avg_pool2d      == F.avg_pool2d(count_include_pad=True)                                                       (Q6)
  cross_entropy   == F.cross_entropy(reduction="mean")
Run:  python tests/golden/make_golden.py      (needs torch; the fixtures are committed, torch never travels)
"""
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

OUT = Path(__file__).resolve().parent
rng = np.random.default_rng(20250928)
torch.manual_seed(0)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv_case(n, ci, h, w, co, name):
    x = f32(rng.uniform(-1, 1, (n, ci, h, w)))
    wt = f32(rng.uniform(-0.5, 0.5, (co, ci, 3, 3)))
    b = f32(rng.uniform(-0.5, 0.5, co))
    w_eff = torch.from_numpy(wt).double().flatten().reshape(9 * ci, co).T.reshape(co, ci, 3, 3)   # Q3 reinterpretation
    y_taper = F.conv2d(torch.from_numpy(x).double(), w_eff, torch.from_numpy(b).double(), padding=1)
    y_std = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=1)
    np.savez_compressed(OUT / f"{name}.npz", x=x, w=wt, b=b, y_taper=f32(y_taper.numpy()), y_standard=f32(y_std.numpy()))


def pool_case(n, c, h, w, k, s, p, name):
    x = f32(rng.integers(-4, 5, (n, c, h, w)))          # ties on purpose: first max must win
    xt = torch.from_numpy(x).double()
    y, idx = F.max_pool2d(xt, k, s, p, return_indices=True)
    base = (np.arange(n * c) * h * w).reshape(n, c, 1, 1)
    avg = F.avg_pool2d(torch.from_numpy(f32(rng.uniform(-1, 1, (n, c, h, w)))).double(), k, s, p, count_include_pad=True)
    xa = f32(rng.uniform(-1, 1, (n, c, h, w)))
    avg = F.avg_pool2d(torch.from_numpy(xa).double(), k, s, p, count_include_pad=True)
    np.savez_compressed(OUT / f"{name}.npz", x=x, k=np.array(k), s=np.array(s), p=np.array(p), y=f32(y.numpy()),
                        argmax=(idx.numpy() + base).astype(np.int64), xa=xa, avg=f32(avg.numpy()))


def xent_case(b, c, name):
    logits = f32(rng.standard_normal((b, c)) * 3)
    t = rng.integers(0, c, b)
    lt = torch.from_numpy(logits).double().requires_grad_()
    loss = F.cross_entropy(lt, torch.from_numpy(t), reduction="mean")
    loss.backward()
    np.savez_compressed(OUT / f"{name}.npz", logits=logits, targets=f32(t), loss=f32(loss.item()), dlogits=f32(lt.grad.numpy()),
                        logp=f32(F.log_softmax(lt, 1).detach().numpy()), argmax=f32(logits.argmax(1)))


def adam_case(name, steps=6):
    """SURVEY.md A.3 (src/optim.rs:83-113) in float64: eps added to sqrt(v) BEFORE the bias fold (Q10)"""
    n = 257
    p = rng.uniform(-1, 1, n)
    grads = rng.standard_normal((steps, n)) * 0.1
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 1e-4
    p0 = f32(p)
    p = p0.astype(np.float64)
    m, v = np.zeros(n), np.zeros(n)
    traj = []
    for t in range(1, steps + 1):
        g = f32(grads[t - 1]).astype(np.float64) + wd * p
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        step = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        p = p - step * m / (np.sqrt(v) + eps)
        traj.append(p.copy())
    np.savez_compressed(OUT / f"{name}.npz", p0=p0, grads=f32(grads), traj=f32(np.array(traj)), m=f32(m), v=f32(v),
                        hyper=np.array([lr, b1, b2, eps, wd]))


def misc_case(name):
    x = f32(rng.uniform(-3, 3, (37, 53)))
    bias = f32(rng.uniform(-1, 1, 53))
    np.savez_compressed(OUT / f"{name}.npz", x=x, bias=bias, transpose=x.T.copy(), add_broadcast=x + bias,
                        sigmoid=f32(torch.sigmoid(torch.from_numpy(x).double()).numpy()), relu=np.maximum(x, 0),
                        colsum=f32(x.astype(np.float64).sum(0)), rowmax=x.max(1), rowargmax=f32(x.argmax(1)))


def extra_cases():
    """SURVEY.md 8(f) rows 2-4 (added later; own generator so the fixtures above keep their bytes):
      bce_loss                  == F.binary_cross_entropy(reduction="mean") for p in (1e-6, 1 - 1e-6)   (loss.rs:6-73)
      cross_entropy_loss_onehot == -(t * log_softmax(x)).sum() / B with one-hot rows                     (loss.rs:201-245)
      StepLR / ExponentialLR / CosineAnnealingLR == torch.optim.lr_scheduler of the same names           (optim.rs:190-288)
    """
    r2 = np.random.default_rng(20250929)
    n = 300
    p = f32(r2.uniform(1e-4, 1 - 1e-4, n))
    y = f32(r2.uniform(0, 1, n) > 0.5)
    pt = torch.from_numpy(p).double().requires_grad_()
    loss = F.binary_cross_entropy(pt, torch.from_numpy(y).double(), reduction="mean")
    loss.backward()
    b, c = 48, 10
    logits = f32(r2.standard_normal((b, c)) * 2.5)
    onehot = np.eye(c, dtype=np.float32)[r2.integers(0, c, b)]
    lt = torch.from_numpy(logits).double().requires_grad_()
    l2 = -(torch.from_numpy(onehot).double() * F.log_softmax(lt, 1)).sum() / b
    l2.backward()
    sched = {}
    for name, mk in (("step", lambda o: torch.optim.lr_scheduler.StepLR(o, 3, 0.5)),
                     ("exp", lambda o: torch.optim.lr_scheduler.ExponentialLR(o, 0.9)),
                     ("cos", lambda o: torch.optim.lr_scheduler.CosineAnnealingLR(o, 10, 1e-4))):
        opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=0.1)
        sc, lrs = mk(opt), []
        for _ in range(10):
            opt.step()
            sc.step()
            lrs.append(sc.get_last_lr()[0])
        sched[name] = np.array(lrs)
    np.savez_compressed(OUT / "losses_extra.npz", p=p, y=y, bce=f32(loss.item()), dbce=f32(pt.grad.numpy()), logits=logits, onehot=onehot,
                        ce=f32(l2.item()), dce=f32(lt.grad.numpy()), lr_step=sched["step"], lr_exp=sched["exp"], lr_cos=sched["cos"])


def grouped_conv_case():
    """Conv2d with groups (nn.rs:289-332) == per-group F.conv2d on the channel slices, each
    group's weight slice reinterpreted per Q3 (the slice is a tensor of its own when it reaches conv2d), cat on dim 1.
    torch's own groups= convolution (standard layout) is stored next to it to show the two differ."""
    r3 = np.random.default_rng(20250930)
    n, ci, h, w, co, g = 3, 8, 9, 7, 12, 4
    x = f32(r3.uniform(-1, 1, (n, ci, h, w)))
    wt = f32(r3.uniform(-0.5, 0.5, (co, ci // g, 3, 3)))
    b = f32(r3.uniform(-0.5, 0.5, co))
    cig, cog = ci // g, co // g
    outs = []
    for k in range(g):
        wg = torch.from_numpy(wt[k * cog:(k + 1) * cog]).double()
        w_eff = wg.flatten().reshape(9 * cig, cog).T.reshape(cog, cig, 3, 3)
        outs.append(F.conv2d(torch.from_numpy(x[:, k * cig:(k + 1) * cig]).double(), w_eff, torch.from_numpy(b[k * cog:(k + 1) * cog]).double(),
                             padding=1))
    y_std = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=1, groups=g)
    np.savez_compressed(OUT / "conv_grouped.npz", x=x, w=wt, b=b, groups=np.array(g), y_taper=f32(torch.cat(outs, 1).numpy()),
                        y_standard=f32(y_std.numpy()))


def general_conv_cases():
    """conv2d through the reference's general im2col (tensor.rs:1805-1970), dilation 1.
    torch F.conv2d with the Q3 weight mapping; for n > 1 the reference reads plane batch*ch + ch instead of
    batch*c + ch (Q9), so torch is fed the planes the reference actually reads (for n == 1 that is the input itself)."""
    r4 = np.random.default_rng(20250931)
    out = {}
    cases = [(1, 3, 12, 12, 6, 5, 5, (1, 1), (2, 2)), (1, 4, 11, 9, 5, 3, 3, (2, 2), (1, 1)), (1, 2, 16, 16, 4, 7, 7, (2, 2), (3, 3)),
             (1, 3, 8, 8, 4, 2, 2, (2, 2), (0, 0)), (3, 4, 9, 10, 5, 5, 5, (1, 1), (2, 2)), (2, 3, 10, 10, 4, 3, 5, (1, 2), (1, 2))]
    for i, (n, c, h, w, co, kh, kw, st, pd) in enumerate(cases):
        x = f32(r4.uniform(-1, 1, (n, c, h, w)))
        wt = f32(r4.uniform(-0.5, 0.5, (co, c, kh, kw)))
        b = f32(r4.uniform(-0.5, 0.5, co))
        planes = x.reshape(n * c, h, w)
        read = np.stack([np.stack([planes[bi * ch + ch] for ch in range(c)]) for bi in range(n)])
        w_eff = torch.from_numpy(wt).double().flatten().reshape(c * kh * kw, co).T.reshape(co, c, kh, kw)
        y = F.conv2d(torch.from_numpy(read).double(), w_eff, torch.from_numpy(b).double(), stride=st, padding=pd)
        out.update({f"x{i}": x, f"w{i}": wt, f"b{i}": b, f"geo{i}": np.array([st[0], st[1], pd[0], pd[1]]), f"y{i}": f32(y.numpy())})
    np.savez_compressed(OUT / "conv_general.npz", count=np.array(len(cases)), **out)


if __name__ == "__main__":
    import sys
    if "--general" in sys.argv:
        general_conv_cases()
        print("wrote conv_general.npz")
        sys.exit(0)
    if "--grouped" in sys.argv:
        grouped_conv_case()
        print("wrote conv_grouped.npz")
        sys.exit(0)
    if "--extra" in sys.argv:
        extra_cases()
        print("wrote losses_extra.npz")
        sys.exit(0)
    conv_case(2, 1, 28, 28, 8, "conv_c1")
    conv_case(2, 16, 14, 14, 24, "conv_c16")
    conv_case(3, 5, 7, 7, 3, "conv_odd")
    pool_case(2, 3, 28, 28, (2, 2), (2, 2), (0, 0), "pool_2x2")
    pool_case(2, 2, 9, 8, (3, 3), (2, 2), (1, 1), "pool_3x3_pad")
    pool_case(3, 4, 7, 7, (7, 7), (7, 7), (0, 0), "pool_global")
    xent_case(64, 10, "xent_64x10")
    xent_case(7, 3, "xent_7x3")
    adam_case("adam_traj")
    misc_case("misc_2d")
    print("wrote", sorted(p.name for p in OUT.glob("*.npz")))
