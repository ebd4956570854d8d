#!/usr/bin/env python3
"""CPU emulation of the split-fp16 MMD arithmetic variants against fp64, on benign and adversarial inputs (design tool).

    python tools/gram_numerics_cpu.py N D [case ...]

Variants of the Gram that feeds t = (|a|^2 + |b|^2 - 2 a.b) / D^2 (fp16 values, fp32 accumulation, as the matrix cores do):
    g2   rows hi + lo, columns hi      (round-2 default, TVAE_GRAM_PRODUCTS = 2)
    g1   hi x hi                       (one product)
each with the loss sums taken as  Q = E + t  (round 2)  or as  R3 = Q - t^2/2  with sum t^2 in closed form from moments,
and each on the raw points or on points centred by the pooled column mean (the MMD is translation invariant).
Printed: relative error of the loss, and of the Q-form gradient (one product X_hi Q_hi) as max|d| / max|g| and relative to
its kernel-dependent part."""
import sys

import torch

torch.set_num_threads(8)
N, D = int(sys.argv[1]), int(sys.argv[2])
cases = sys.argv[3:] or ["iid", "dup64", "tern", "lev8", "collapsed", "prior+3", "outlier50"]
inv = 1.0 / float(D * D)


def make(case):
    g = torch.Generator().manual_seed(7)
    z = torch.tanh(torch.randn((N, D), generator=g))
    y = torch.randn((N, D), generator=g)
    if case == "dup64":
        z = z[: N // 64].repeat_interleave(64, 0)
        y = y[: N // 64].repeat_interleave(64, 0)
    elif case == "tern":
        z = torch.round(z)
    elif case == "lev8":
        z = torch.round(z * 4) / 4
    elif case == "collapsed":
        z = z[:1].repeat(N, 1).contiguous()
    elif case == "prior+3":
        y = y + 3.0
    elif case == "outlier50":
        z = z / z.norm(dim=1, keepdim=True)
        y = y / y.norm(dim=1, keepdim=True)
        z[17] *= 50.0
    elif case == "z+0.6":
        z = torch.tanh(torch.randn((N, D), generator=g) + 0.6)
    return z.float().contiguous(), y.float().contiguous()


def split16(x, s):
    hi = (x * s).half()
    lo = (x * s - hi.float()).half()
    return hi.float(), lo.float()


def r3_poly(x):          # x = -t:  R3 = x^3 (1/6 + x/24 + x^2/120 + x^3/720)
    return x * x * x * (1.0 / 6 + x * (1.0 / 24 + x * (1.0 / 120 + x * (1.0 / 720))))


def run(case):
    z, y = make(case)
    z64, y64 = z.double(), y.double()
    out = {}
    for centred in (False, True):
        if centred:
            mu = ((z64.sum(0) + y64.sum(0)) / (2 * N)).float()
            zc, yc = z - mu, y - mu          # fp32 subtraction, as a prep pass would do
        else:
            zc, yc = z, y
        zc64, yc64 = zc.double(), yc.double()
        amax = float(torch.maximum(zc.abs().max(), yc.abs().max()))
        import math
        e = math.frexp(amax)[1]
        s = 2.0 ** (12 - e)
        zh, zl = split16(zc, s)
        yh, yl = split16(yc, s)
        nz32, ny32 = (zc * zc).sum(1), (yc * yc).sum(1)                # fp32 norms as stored by the prep pass
        nz, ny = nz32.double(), ny32.double()
        # fp64 truth on the ORIGINAL points
        if not centred:
            tnz, tny = (z64 * z64).sum(1), (y64 * y64).sum(1)
            tot = {}
            gq = torch.zeros((N, D), dtype=torch.float64)
            rq = torch.zeros(N, dtype=torch.float64)
            for name, (a, na, b, nb) in {"zz": (z64, tnz, z64, tnz), "zy": (z64, tnz, y64, tny), "yy": (y64, tny, y64, tny)}.items():
                acc = 0.0
                for r0 in range(0, N, 1024):
                    t = ((na[r0:r0 + 1024, None] + nb[None, :] - 2.0 * (a[r0:r0 + 1024] @ b.T)) * inv).clamp_min(0)
                    E = torch.expm1(-t)
                    acc += float(E.sum())
                    if name == "zz":
                        gq[r0:r0 + 1024] += E @ b
                        rq[r0:r0 + 1024] += E.sum(1)
                    elif name == "zy":
                        gq[r0:r0 + 1024] -= E @ b
                        rq[r0:r0 + 1024] -= E.sum(1)
                tot[name] = acc
            loss64 = (tot["yy"] + tot["zz"] - 2 * tot["zy"]) / (N * N)
            grad64 = (-(z64 * rq[:, None]) + (z64.sum(0) - y64.sum(0))[None, :] + gq) * (4.0 * inv / (N * N))
            shift_part = (z64.sum(0) - y64.sum(0))[None, :] * (4.0 * inv / (N * N))
            out["loss64"] = loss64
            truth = (loss64, grad64, shift_part)
        loss64, grad64, shift_part = truth
        # closed-form pieces (fp64 from the stored fp32 values)
        cz, cy = zc64.sum(0), yc64.sum(0)

        def lin_sum(na, ca, nb, cb):      # sum_ij (a_i + b_j - 2 u_i.v_j)
            return N * na.sum() + N * nb.sum() - 2.0 * (ca @ cb)

        def sq_sum(na, ua, nb, ub):       # sum_ij (a_i + b_j - 2 u_i.v_j)^2, the Gram matrices in fp32 like the GEMM kernel
            ca, cb = ua.sum(0), ub.sum(0)
            au, bv = (na[:, None] * ua).sum(0), (nb[:, None] * ub).sum(0)
            ga = (ua.float().T @ ua.float()).double()
            gb = (ub.float().T @ ub.float()).double()
            return (N * (na * na).sum() + 2 * na.sum() * nb.sum() + N * (nb * nb).sum()
                    - 4.0 * (au @ cb + ca @ bv) + 4.0 * (ga * gb).sum())

        for gv in ("g2", "g1"):
            qs, r3s = {}, {}
            gacc = torch.zeros((N, D), dtype=torch.float32)
            racc = torch.zeros(N, dtype=torch.float64)
            for name, (ah, al, na, bh, nb, sign) in {"zz": (zh, zl, nz32, zh, nz32, 1.0), "zy": (zh, zl, nz32, yh, ny32, -1.0),
                                                     "yy": (yh, yl, ny32, yh, ny32, 0.0)}.items():
                qa = ra = 0.0
                for r0 in range(0, N, 1024):
                    S = ah[r0:r0 + 1024] @ bh.T
                    if gv == "g2":
                        S = S + al[r0:r0 + 1024] @ bh.T
                    t = ((na[r0:r0 + 1024, None] + nb[None, :] - (2.0 / (s * s)) * S) * inv).clamp_min(0)
                    x = -t
                    r3 = r3_poly(x)
                    q = 0.5 * x * x + r3
                    qa += float(q.double().sum())
                    ra += float(r3.double().sum())
                    if sign != 0.0:
                        qh = (q * 4096.0).half().float()
                        gacc[r0:r0 + 1024] += sign * (qh @ bh) / (4096.0 * s)
                        racc[r0:r0 + 1024] += sign * q.double().sum(1)
                qs[name], r3s[name] = qa, ra
            lin = {"zz": lin_sum(nz, cz, nz, cz), "zy": lin_sum(nz, cz, ny, cy), "yy": lin_sum(ny, cy, ny, cy)}
            sq = {"zz": sq_sum(nz, zc64, nz, zc64), "zy": sq_sum(nz, zc64, ny, yc64), "yy": sq_sum(ny, yc64, ny, yc64)}
            for form in ("Q", "R3"):
                S3 = {}
                for k in ("zz", "zy", "yy"):
                    second = qs[k] if form == "Q" else r3s[k] + 0.5 * float(sq[k]) * inv * inv
                    S3[k] = second - float(lin[k]) * inv
                loss = (S3["yy"] + S3["zz"] - 2 * S3["zy"]) / (N * N)
                out[(centred, gv, form)] = abs(loss - loss64) / abs(loss64)
            # Q-form gradient: closed-form linear part + the one-product Q sums
            shiftv = cz - cy
            m3 = (nz[:, None] * zc64).sum(0) - (ny[:, None] * yc64).sum(0)
            M2 = (zc.T @ zc).double() - (yc.T @ yc).double()
            Pm = (zc @ M2.float()).double()
            sz = zc64 @ shiftv
            glin = -inv * (m3[None, :] + nz[:, None] * shiftv[None, :] - 2.0 * Pm)
            rs = racc - inv * ((nz.sum() - ny.sum()) - 2.0 * sz)
            grad = (-(zc64 * rs[:, None]) + (z64.sum(0) - y64.sum(0))[None, :] + gacc.double() + glin) * (4.0 * inv / (N * N))
            # centred gradient is the gradient wrt the original z (translation invariance)
            d = (grad - grad64).abs().max()
            out[(centred, gv, "grad")] = (float(d / grad64.abs().max()), float(d / (grad64 - shift_part).abs().max()))
    return out


for case in cases:
    o = run(case)
    print("== %s  N=%d D=%d  loss64 %.6e" % (case, N, D, o["loss64"]), flush=True)
    for centred in (False, True):
        for gv in ("g2", "g1"):
            ge = o[(centred, gv, "grad")]
            print("   %-8s %s  loss err: Q sums %.2e  R3 sums %.2e   grad err %.2e (kernel part %.2e)" % (
                "centred" if centred else "raw", gv, o[(centred, gv, "Q")], o[(centred, gv, "R3")], ge[0], ge[1]), flush=True)
