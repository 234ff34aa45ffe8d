"""Randomised check of the point-major FP stage (csrc/channels_last.hip, gather_rows_csr_*_cl) against float64 torch:
random batch / point / table / channel / skip sizes (channel counts that do not fill a wave, tables of 3 rows, clouds
smaller than a tile), neighbour ids from three_nn or adversarial (hubs, untouched targets, two thirds of all pairs on one target), with
and without a row order, BatchNorm in training and in eval mode, ReLU on / off.

For every case: forward of fp_stage_cl, gradients w.r.t. the table, the skip weights, gamma and beta -- each against the
same function written with torch ops in float64 on the CPU; the fused backward against the two-node form; a rebuilt
reverse index + rerun must give the same bits."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd import fused_norm as fn  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402
from geot_amd.pointnet2 import pointnet2_utils as pu  # noqa: E402

DEV = torch.device("cuda:0")
CASES = int(os.environ.get("CASES", "60"))
rng = np.random.default_rng(int(os.environ.get("SEED", "7")))


def rel(got, want):
    want = want.detach().double().cpu()
    return float((got.detach().double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-30))


def ref64(a_cl, idx, w, skip, wb, bn, relu, up):
    a64 = a_cl.double().cpu().requires_grad_(True)
    wb64 = None if wb is None else wb.double().cpu().requires_grad_(True)
    b, m, c = a64.shape
    n = idx.shape[1]
    g = torch.gather(a64, 1, idx.cpu().long().reshape(b, n * 3, 1).expand(-1, -1, c)).view(b, n, 3, c)
    y = (g * w.double().cpu().unsqueeze(-1)).sum(2)
    if skip is not None:
        y = y + torch.matmul(skip.double().cpu().transpose(1, 2), wb64.t())
    bn64 = torch.nn.BatchNorm1d(c).double()
    bn64.load_state_dict({k: v.double().cpu() if v.dtype.is_floating_point else v.cpu() for k, v in bn.state_dict().items()})
    bn64.train(bn.training)
    pre = bn64(y.transpose(1, 2)).transpose(1, 2)
    z = torch.relu(pre) if relu else pre
    # a ReLU input within fp32 rounding of zero has no defined mask at this precision (any two fp32 evaluations may disagree
    # and move a whole gradient element): those elements get no upstream gradient, in the referee and on the GPU alike
    keep = (pre.detach().abs() > 1e-5) if relu else torch.ones_like(pre, dtype=torch.bool)
    up64 = up.double().cpu() * keep
    (z * up64).sum().backward()
    return z.detach(), a64.grad, None if wb64 is None else wb64.grad, bn64.weight.grad, bn64.bias.grad, bn64, keep


worst = {}
skipped = 0
done = 0
case = 0
while done < CASES:
    case += 1
    b = int(rng.integers(1, 4))
    n = int(rng.choice([33, 257, 1000, 3001, 6000]))
    m = int(rng.choice([3, 17, 200, 1500]))
    c = int(rng.choice([256, 260, 384, 512, 1028]))
    cs = int(rng.choice([0, 1, 3, 5, 8]))
    relu = bool(rng.integers(0, 2))
    training = bool(rng.integers(0, 4) > 0)
    ordered = bool(rng.integers(0, 2))
    mode = str(rng.choice(["nn", "hub", "one", "random"]))
    pos = torch.from_numpy(make_batch(b, max(n, m), start_index=case)[0]).to(DEV)
    unknown, known = pos[:, :n].contiguous(), pos[:, :m].contiguous()
    d2, idx = pu._ext.three_nn(unknown, known)
    w = pu._ext.fp_weights(d2)
    if mode == "hub":
        idx = idx.clone()
        idx[:, ::2, 0] = 0
        if m > 2:
            idx[idx == m - 1] = 1
    elif mode == "one":            # the two FARTHER slots of every point on target 0: one list holds two thirds of all pairs.  (All three
        idx = idx.clone()          # slots, or the nearest ones, would make y (nearly) constant per channel: a BatchNorm input whose
        idx[:, :, 1:] = 0          # variance is rounding noise, where fp32 and fp64 differ by construction.)
    elif mode == "random":
        idx = torch.from_numpy(rng.integers(0, m, size=(b, n, 3)).astype(np.int32)).to(DEV)
    idx = idx.contiguous()
    torch.manual_seed(case)
    a_cl = torch.randn(b, m, c, device=DEV)
    skip = torch.randn(b, cs, n, device=DEV) if cs else None
    wb = torch.randn(c, cs, device=DEV) if cs else None
    up = torch.randn(b, n, c, device=DEV)
    bn = torch.nn.BatchNorm1d(c).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.2, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
    bn.train(training)
    z64, ga64, gwb64, gg64, gb64, bn64, keep = ref64(a_cl, idx, w, skip, wb, bn, relu, up)
    skipped += int((~keep).sum())
    up = up * keep.to(DEV)
    order_u = fn.local_spatial_order(unknown) if ordered else None
    order_k = fn.local_spatial_order(known) if ordered else None
    res = {}
    for form in ("fused", "two-node"):
        import copy
        bn_f = copy.deepcopy(bn)
        a_r = a_cl.clone().requires_grad_(True)
        wb_r = None if wb is None else wb.clone().requires_grad_(True)
        rix = fn.ReverseIndex(idx, w, m, order_k)
        if form == "fused":
            z = fn.fp_stage_cl(bn_f, a_r, idx, w, skip, wb_r, relu, order_u, rix)
        else:
            y, part = fn.fp_front_cl(a_r, idx, w, skip, wb_r, order_u, rix)
            z = fn.bn_act_cl(bn_f, y, relu=relu, partial=part)
        (z * up).sum().backward()
        res[form] = (z.detach(), a_r.grad, None if wb_r is None else wb_r.grad, bn_f.weight.grad, bn_f.bias.grad, bn_f)
    errs = {}
    for form, (z, ga, gwb, gg, gb, bn_f) in res.items():
        errs[form + " z"] = rel(z, z64)
        errs[form + " dA"] = rel(ga, ga64)
        if gwb is not None:
            errs[form + " dWb"] = rel(gwb, gwb64)
        errs[form + " dgamma"] = rel(gg, gg64)
        errs[form + " dbeta"] = rel(gb, gb64)
        if training:
            errs[form + " running_var"] = rel(bn_f.running_var, bn64.running_var)
    # rebuilt index, rerun: same bits
    rix2 = fn.ReverseIndex(idx, w, m, order_k)
    a_r = a_cl.clone().requires_grad_(True)
    z2 = fn.fp_stage_cl(copy.deepcopy(bn), a_r, idx, w, skip, None if wb is None else wb.clone().requires_grad_(True), relu, order_u, rix2)
    (z2 * up).sum().backward()
    # (lists longer than RIX_SORT_MAX = 4096 pairs -- degenerate hubs -- keep the arrival order of their pairs: correct, not
    # reproducible bit for bit; geot_common.h)
    longest = int(torch.bincount((idx.long() + torch.arange(b, device=DEV).view(b, 1, 1) * m).reshape(-1), minlength=b * m).max())
    same = torch.equal(z2, res["fused"][0]) and (longest > 4096 or torch.equal(a_r.grad, res["fused"][1]))
    # a list of N pairs is summed in fp32 in list order: error grows like sqrt(N) eps x (sum |terms| / |result|)
    tol = 1e-4 if longest > 2000 else (5e-5 if longest > 256 else 2e-5)
    bad = {k: v for k, v in errs.items() if v > (1e-5 if k.endswith(" z") else tol)}
    for k, v in errs.items():
        key = k.split(" ", 1)[1]
        worst[key] = max(worst.get(key, 0.0), v)
    done += 1
    status = "ok" if (not bad and same) else "FAIL %s same=%s" % (bad, same)
    print("case %3d b=%d n=%4d m=%4d c=%4d cs=%d relu=%d train=%d ordered=%d ids=%-6s worst %.1e  %s" %
          (done, b, n, m, c, cs, relu, training, ordered, mode, max(errs.values()), status), flush=True)
    if bad or not same:
        sys.exit(1)
print("%d cases, 0 failures (%d ReLU inputs within 1e-5 of zero given no upstream gradient); worst relative errors vs float64: %s" %
      (done, skipped, {k: "%.1e" % v for k, v in sorted(worst.items())}))
