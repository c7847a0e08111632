"""Per-shape A/B of the backbone's pointwise (1 x 1) convolutions at the bench shape (4 x 3 x 800 x 800, fp32, channels-last):
vendor convolution (MIOpen through PyTorch) + the one-pass epilogue kernel  vs  ttdg_mm_f32 with the epilogue fused
(csrc/pointwise.hip), forward and the two backward products, every tile code.  Checks every arm against float64.

    python tools/bench_pointwise.py [out.json] [fwd|bwd|all] [reps]
"""
import json
import sys

sys.path.insert(0, ".")
import torch
import torch.nn.functional as F

from ttdg_mgm_amd import ops

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = False
CL = torch.channels_last
B = 4

# (name, H_in, Cin, Cout, stride, residual, relu, count per forward)   H_in = W_in
FWD = [
    ("res2.b0.conv1", 200, 64, 64, 1, False, True, 1),
    ("res2.conv3+res", 200, 64, 256, 1, True, True, 3),
    ("res2.b0.shortcut", 200, 64, 256, 1, False, False, 1),
    ("res2.conv1", 200, 256, 64, 1, False, True, 2),
    ("res3.b0.conv1/s2", 200, 256, 128, 2, False, True, 1),
    ("res3.b0.shortcut/s2", 200, 256, 512, 2, False, False, 1),
    ("res3.conv3+res", 100, 128, 512, 1, True, True, 4),
    ("res3.conv1", 100, 512, 128, 1, False, True, 3),
    ("res4.b0.conv1/s2", 100, 512, 256, 2, False, True, 1),
    ("res4.b0.shortcut/s2", 100, 512, 1024, 2, False, False, 1),
    ("res4.conv3+res", 50, 256, 1024, 1, True, True, 6),
    ("res4.conv1", 50, 1024, 256, 1, False, True, 5),
    ("res5.b0.conv1/s2", 50, 1024, 512, 2, False, True, 1),
    ("res5.b0.shortcut/s2", 50, 1024, 2048, 2, False, False, 1),
    ("res5.conv3+res", 25, 512, 2048, 1, True, True, 3),
    ("res5.conv1", 25, 2048, 512, 1, False, True, 2),
    ("fpn.lateral5", 25, 2048, 256, 1, False, False, 1),
    ("fpn.lateral4+up", 50, 1024, 256, 1, "up", False, 1),
    ("fpn.lateral3+up", 100, 512, 256, 1, "up", False, 1),
    ("fpn.lateral2+up", 200, 256, 256, 1, "up", False, 1),
]


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def relerr(got, want64):
    return float((got.double() - want64).abs().max() / want64.abs().max().clamp_min(1e-30))


def bench_fwd(reps, rows):
    g = torch.Generator(device="cpu").manual_seed(1)
    for name, H, Cin, Cout, s, res, relu, cnt in FWD:
        Ho = (H - 1) // s + 1
        x = torch.randn(B, Cin, H, H, generator=g).to(dev).contiguous(memory_format=CL)
        w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).to(dev).contiguous(memory_format=CL)
        b = torch.randn(Cout, generator=g).to(dev)
        r = None
        if res is True:
            r = torch.randn(B, Cout, Ho, Ho, generator=g).to(dev).contiguous(memory_format=CL)
        elif res == "up":
            r = torch.randn(B, Cout, Ho // 2, Ho // 2, generator=g).to(dev).contiguous(memory_format=CL)
        M = B * Ho * Ho
        flops = 2.0 * M * Cin * Cout
        byts = 4.0 * (M * Cin + M * Cout * (2 if r is not None and res is True else 1) + Cin * Cout)

        def vendor():
            y = F.conv2d(x, w, None, s, 0)
            rr = F.interpolate(r, scale_factor=2.0, mode="nearest") if res == "up" else r
            return ops.bias_act_(y, b, rr, None, relu=relu)

        out = torch.empty(B, Cout, Ho, Ho, device=dev).contiguous(memory_format=CL)

        def own(tile):
            return ops.mm(x, w, out, M, Cout, Cin, Cin, Cin, Cout, bias=b, res=r, ldres=Cout, relu=relu, a_stride=s, a_hw=(H, H),
                          res_up=(res == "up"), res_hw=(Ho, Ho), tile=tile)

        want = F.conv2d(x.double(), w.double(), b.double(), s, 0)
        if r is not None:
            want = want + (F.interpolate(r.double(), scale_factor=2.0, mode="nearest") if res == "up" else r.double())
        if relu:
            want = want.relu()
        row = {"name": name, "M": M, "K": Cin, "N": Cout, "stride": s, "count": cnt, "gflop": flops / 1e9, "mbytes": byts / 1e6}
        tv = timed(vendor, reps)
        row["vendor_us"], row["vendor_err"] = tv, relerr(vendor(), want)
        best = None
        for tile in (1, 2, 3, 4):
            if tile in (1, 3) and Cout <= 64:
                continue
            own(tile)
            err = relerr(out, want)
            t = timed(lambda: own(tile), reps)
            row["own_t%d_us" % tile], row["own_t%d_err" % tile] = t, err
            if best is None or t < best[0]:
                best = (t, tile)
        own(0)
        row["own_auto_us"] = timed(lambda: own(0), reps)
        row["own_best_us"], row["own_best_tile"] = best
        row["own_tflops"], row["own_gbs"] = flops / best[0] / 1e6, byts / best[0] / 1e3
        row["vendor_tflops"] = flops / tv / 1e6
        rows.append(row)
        print("%-22s M=%6d K=%4d N=%4d | vendor+epi %7.1f us (%5.1f TF, err %.1e) | own %s | best t%d %7.1f us (%5.1f TF, %4.0f GB/s) auto %7.1f | x%.2f" % (
            name, M, Cin, Cout, tv, row["vendor_tflops"], row["vendor_err"],
            " ".join("t%d %7.1f (%.0e)" % (t, row["own_t%d_us" % t], row["own_t%d_err" % t]) for t in (1, 2, 3, 4) if "own_t%d_us" % t in row),
            best[1], best[0], row["own_tflops"], row["own_gbs"], row["own_auto_us"], tv / best[0]), flush=True)
    tot_v = sum(r_["vendor_us"] * r_["count"] for r_ in rows if "vendor_us" in r_)
    tot_o = sum(r_["own_best_us"] * r_["count"] for r_ in rows if "own_best_us" in r_)
    print("forward, weighted by layer count: vendor + epilogue %.1f us, own (best tile) %.1f us" % (tot_v, tot_o), flush=True)


BWD = [  # (name, H_out, Cin, Cout, count)  stride-1 layers of the adapted stages + FPN laterals
    ("res3.conv3", 100, 128, 512, 4), ("res3.conv1", 100, 512, 128, 3),
    ("res4.conv3", 50, 256, 1024, 6), ("res4.conv1", 50, 1024, 256, 5),
    ("res5.conv3", 25, 512, 2048, 3), ("res5.conv1", 25, 2048, 512, 2),
    ("fpn.lateral5", 25, 2048, 256, 1), ("fpn.lateral4", 50, 1024, 256, 1), ("fpn.lateral3", 100, 512, 256, 1), ("fpn.lateral2", 200, 256, 256, 1),
]


def bench_bwd(reps, rows):
    g = torch.Generator(device="cpu").manual_seed(2)
    for name, H, Cin, Cout, cnt in BWD:
        M = B * H * H
        x = torch.randn(B, Cin, H, H, generator=g).to(dev).contiguous(memory_format=CL)
        w = (torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5).to(dev).contiguous(memory_format=CL)
        go = torch.randn(B, Cout, H, H, generator=g).to(dev).contiguous(memory_format=CL)

        def vendor(mask):
            return torch.ops.aten.convolution_backward(go, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, mask)

        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        wt = w.view(Cout, Cin).t().contiguous()

        def own_dx(tile, transposed):
            if transposed:      # B(n' = cin, k' = cout) = W^T[cin, cout]: k-contiguous
                return ops.mm(go, wt, dx, M, Cin, Cout, Cout, Cout, Cin, tile=tile)
            return ops.mm(go, w, dx, M, Cin, Cout, Cout, Cin, Cin, b_layout=1, tile=tile)

        def own_dw(tile, ks):
            return ops.mm(go, x, dw, Cout, Cin, M, Cout, Cin, Cin, a_layout=1, b_layout=1, kslices=ks, tile=tile)

        want_dx = torch.ops.aten.convolution_backward(go.double(), x.double(), w.double(), None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, True, False))
        row = {"name": name, "M": M, "K": Cin, "N": Cout, "count": cnt, "gflop": 2.0 * M * Cin * Cout / 1e9}
        vdx, vdw = vendor((True, False, False))[0], vendor((False, True, False))[1]
        row["vendor_dx_us"], row["vendor_dw_us"] = timed(lambda: vendor((True, False, False)), reps), timed(lambda: vendor((False, True, False)), reps)
        row["vendor_both_us"] = timed(lambda: vendor((True, True, False)), reps)
        row["vendor_dx_err"], row["vendor_dw_err"] = relerr(vdx, want_dx[0]), relerr(vdw, want_dx[1])
        bx = None
        for tr in (False, True):
            for tile in (1, 2, 3, 4):
                if tile in (1, 3) and Cin <= 64:
                    continue
                own_dx(tile, tr)
                err = relerr(dx, want_dx[0])
                t = timed(lambda: own_dx(tile, tr), reps)
                row["own_dx_%s_t%d" % ("T" if tr else "N", tile)] = (t, err)
                if bx is None or t < bx[0]:
                    bx = (t, tile, tr, err)
        bw = None
        for tile in (1, 2, 3, 4):
            if tile in (1, 3) and Cin <= 64:
                continue
            bm, bn = (64 if (tile - 1) >> 1 else 128), (64 if (tile - 1) & 1 else 128)
            tiles = ((Cout + bm - 1) // bm) * ((Cin + bn - 1) // bn)
            for target in (256, 512, 1024):
                ks = max(2, min(M // 256, (target + tiles - 1) // tiles))
                own_dw(tile, ks)
                err = relerr(dw, want_dx[1])
                t = timed(lambda: own_dw(tile, ks), reps)
                row["own_dw_t%d_ks%d" % (tile, ks)] = (t, err)
                if bw is None or t < bw[0]:
                    bw = (t, tile, ks, err)
        row["own_dx_best"], row["own_dw_best"] = bx, bw
        rows.append(row)
        print("%-14s M=%6d K=%4d N=%4d | vendor dx %7.1f (%.0e) dw %7.1f (%.0e) both %7.1f | own dx %7.1f t%d %s (%.0e) %5.1f TF | own dw %7.1f t%d ks%d (%.0e) %5.1f TF | x%.2f" % (
            name, M, Cin, Cout, row["vendor_dx_us"], row["vendor_dx_err"], row["vendor_dw_us"], row["vendor_dw_err"], row["vendor_both_us"],
            bx[0], bx[1], "W^T" if bx[2] else "W", bx[3], row["gflop"] / bx[0] * 1e3, bw[0], bw[1], bw[2], bw[3], row["gflop"] / bw[0] * 1e3,
            row["vendor_both_us"] / (bx[0] + bw[0])), flush=True)
    tot_v = sum(r_["vendor_both_us"] * r_["count"] for r_ in rows if "vendor_both_us" in r_)
    tot_o = sum((r_["own_dx_best"][0] + r_["own_dw_best"][0]) * r_["count"] for r_ in rows if "own_dx_best" in r_)
    print("backward, weighted by layer count: vendor %.1f us, own (best) %.1f us" % (tot_v, tot_o), flush=True)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else None
    what = sys.argv[2] if len(sys.argv) > 2 else "all"
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows = []
    if what in ("fwd", "all"):
        bench_fwd(reps, rows)
    if what in ("bwd", "all"):
        bench_bwd(reps, rows)
    if out:
        with open(out, "w") as f:
            json.dump(rows, f, indent=1)
