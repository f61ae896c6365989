#!/usr/bin/env python3
"""Times every workgroup-tile / split-K decomposition of the convolution GEMM (csrc/conv.hip) on every layer shape of the
depth network at the benchmark size (batch 2, 480x640), forward and backward-data, in ONE process (HIP events, interleaved
rounds), next to the decomposition the built-in cost model picks -- the calibration data of `gemm_cost` in conv.hip.

    python tools/gemm_tune.py [fwd|bwd|both] > gpurun_out/gemm_tune.txt        (on an MI355X)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
from e2ehip import _lib as L  # noqa: E402

DEV = "cuda:0"
# name, Cx, Cskip, up, H, W (full-res input of the conv), Cout, k, stride, pad, pad_mode
LAYERS = [
    ("layer1", 64, 0, 1, 120, 160, 64, 3, 1, 1, 0), ("l2.0.c1/2", 64, 0, 1, 120, 160, 128, 3, 2, 1, 0), ("layer2", 128, 0, 1, 60, 80, 128, 3, 1, 1, 0),
    ("l3.0.c1/2", 128, 0, 1, 60, 80, 256, 3, 2, 1, 0), ("layer3", 256, 0, 1, 30, 40, 256, 3, 1, 1, 0), ("l4.0.c1/2", 256, 0, 1, 30, 40, 512, 3, 2, 1, 0),
    ("layer4", 512, 0, 1, 15, 20, 512, 3, 1, 1, 0), ("up(4,0)", 512, 0, 1, 15, 20, 256, 3, 1, 1, 1), ("up(4,1)", 256, 256, 2, 30, 40, 256, 3, 1, 1, 1),
    ("up(3,0)", 256, 0, 1, 30, 40, 128, 3, 1, 1, 1), ("up(3,1)", 128, 128, 2, 60, 80, 128, 3, 1, 1, 1), ("up(2,0)", 128, 0, 1, 60, 80, 64, 3, 1, 1, 1),
    ("up(2,1)", 64, 64, 2, 120, 160, 64, 3, 1, 1, 1), ("up(1,0)", 64, 0, 1, 120, 160, 32, 3, 1, 1, 1), ("up(1,1)", 32, 64, 2, 240, 320, 32, 3, 1, 1, 1),
    ("up(0,0)", 32, 0, 1, 240, 320, 16, 3, 1, 1, 1), ("up(0,1)", 16, 0, 2, 480, 640, 16, 3, 1, 1, 1),
]
TILES = [(64, 64), (128, 64), (128, 128), (128, 32), (32, 128), (32, 64), (64, 32), (32, 32)]


def ld(n):
    return (n + 3) // 4 * 4


def timeit(fn, n=12):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    lib = L.load()
    B = 2
    st = L.stream()
    only = [x for x in os.environ.get("GEMM_TUNE_LAYERS", "").split(",") if x]
    product_only = os.environ.get("GEMM_TUNE_PRODUCT_ONLY") == "1"
    for name, Cx, Cs, up, H, W, Cout, k, s, p, pm in LAYERS:
        if only and name not in only:
            continue
        Cin = Cx + Cs
        x = torch.randn(B, H // up, W // up, Cx, device=DEV)
        skip = torch.randn(B, H, W, Cs, device=DEV) if Cs else None
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        wf = torch.randn(k * k * Cin, ld(Cout), device=DEV) * 0.05
        wb = torch.randn(k * k * Cout, ld(Cin), device=DEV) * 0.05
        bias = torch.randn(Cout, device=DEV)
        out = torch.empty(B, Ho, Wo, Cout, device=DEV)
        dz = torch.randn(B, Ho, Wo, Cout, device=DEV)
        pp = p if pm == 1 else 0
        dxp = torch.empty(B, H + 2 * pp, W + 2 * pp, Cin, device=DEV)
        rows_f, rows_b = B * Ho * Wo, B * (H + 2 * pp) * (W + 2 * pp)
        ws = torch.zeros(max(lib.e2e_conv_tuned_workspace_floats(rows_f, Cout), lib.e2e_conv_tuned_workspace_floats(rows_b, Cin)), device=DEV)
        gf = 2.0 * B * Ho * Wo * Cout * Cin * k * k / 1e9
        tune = [0, 0, 0]

        def fwd():
            L.call("e2e_conv2d_fwd_tuned", L.ptr(x), L.ptr(skip), Cx, up, L.ptr(wf), ld(Cout), None, L.ptr(bias), None, L.ptr(out), B, H, W, Cin, Cout, k, k, s, p, pm,
                   2, 0.0, 1.0, L.ptr(ws), tune[0], tune[1], tune[2], st)

        def bwd():
            L.call("e2e_conv2d_bwd_data_fused_tuned", L.ptr(dz), L.ptr(wb), ld(Cin), L.ptr(dxp), B, H, W, Cin, Cout, Ho, Wo, k, k, s, p, pm, 0, None, 0, None,
                   L.ptr(ws), tune[0], tune[1], tune[2], st)

        if which in ("both", "wgrad"):
            dw = torch.empty(Cout, Cin, k, k, device=DEV)
            db = torch.empty(Cout, device=DEV)
            res = []
            for tgt in (() if product_only else (256, 384, 512, 768, 1024, 1536, 2048)):
                wsw = torch.empty(lib.e2e_conv2d_wgrad_tuned_workspace_floats(B, Ho, Wo, Cin, Cout, k, k, 1, tgt), device=DEV)

                def wg():
                    L.call("e2e_conv2d_bwd_weight_scaled_tuned", L.ptr(dz), None, L.ptr(x), L.ptr(skip), Cx, up, L.ptr(dw), L.ptr(db), L.ptr(wsw), B, H, W, Cin, Cout,
                           Ho, Wo, k, k, s, p, pm, 0, 0.0, 1.0, tgt, st)
                res.append((timeit(wg), tgt))
            # what the product launches (the library's own choice: the tap-reuse patch kernel where the layer is eligible, else its GEMM rule)
            wsd = torch.empty(lib.e2e_conv2d_wgrad_workspace_floats(B, Ho, Wo, Cin, Cout, k, k, 1), device=DEV)

            def wd():
                L.call("e2e_conv2d_bwd_weight_scaled", L.ptr(dz), None, L.ptr(x), L.ptr(skip), Cx, up, L.ptr(dw), L.ptr(db), L.ptr(wsd), B, H, W, Cin, Cout,
                       Ho, Wo, k, k, s, p, pm, 0, 0.0, 1.0, st)
            t_prod = timeit(wd)
            best = f"  (best {min(res)[1]}: {gf / min(res)[0] * 1e3:.1f} TF/s)" if res else ""
            print(f"{name:10s} wgrad {gf:5.2f} GF | " + " ".join(f"{tgt}:{t:.1f}" for t, tgt in res) + best +
                  f"  || product default {t_prod:.1f} us ({gf / t_prod * 1e3:.1f} TF/s)", flush=True)
        for tag, fn, ncols, K in (("fwd", fwd, Cout, k * k * Cin), ("bwd", bwd, Cin, k * k * Cout)):
            if which not in ("both", tag):
                continue
            tune[:] = [0, 0, 0]
            t_auto = timeit(fn)
            res = []
            if Cin % 16 == 0 and (Cs == 0 or Cx % 16 == 0) and not (s == 2 and tag == "bwd"):
                for G in (256, 384, 512, 640, 768):                      # stream-K on G persistent workgroups
                    tune[:] = [64, 64, -G]
                    res.append((timeit(fn), 64, 64, -G))
            for bm, bn in TILES:
                if bn > 32 and ncols <= 16 or bn > 64 and ncols <= 64:
                    continue
                if bm == 128 and bn == 128 and Cin % 32 == 0 and (Cs == 0 or Cx % 32 == 0) and tag == "fwd":
                    continue                      # 128x128 exists at chunk depth 16 only
                if bm == 128 and bn == 128 and tag == "bwd" and Cout % 32 == 0:
                    continue
                for S in (1, 2, 3, 4, 6, 8, 12, 16):
                    if S > 1 and (K // 32 // S) * 32 < 128:
                        break
                    if S > 1 and s == 2 and tag == "bwd":
                        break
                    tune[:] = [bm, bn, S]
                    res.append((timeit(fn), bm, bn, S))
            tune[:] = [0, 0, 0]
            t_auto = min(t_auto, timeit(fn))          # again AFTER the sweep: the first measurement of a layer (fresh buffers, clocks ramping) reads 1-5 us high
            if int(ws.view(torch.int32)[lib.e2e_conv_streamk_error_index()]) != 0:
                print(f"{name} {tag}: STREAM-K TIME-OUT FLAG RAISED", flush=True)
            res.sort()
            best = res[0]
            line = " ".join((f"sk{-S}:{t:.1f}" if S < 0 else f"{bm}x{bn}/{S}:{t:.1f}") for t, bm, bn, S in res[:6])
            line += "  || stream-K: " + " ".join(f"sk{-S}:{t:.1f}" for t, bm, bn, S in sorted(res, key=lambda r: r[3]) if S < 0)
            print(f"{name:10s} {tag} {gf:5.2f} GF  auto {t_auto:7.1f} us ({gf / t_auto * 1e3:5.1f} TF/s)  best {best[0]:7.1f} us ({gf / best[0] * 1e3:5.1f} TF/s)  | {line}", flush=True)


if __name__ == "__main__":
    main()
