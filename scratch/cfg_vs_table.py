"""The decomposition the built-in cost model picks for every layer of tools/gemm_tune.py (host-only query, runs without a GPU) next to that
layer's measured table (profiles/r04_gemm_tune_final.txt): is the pick among the six best listed, and at what time?"""
import ctypes, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
from e2ehip import _lib as L
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("gt", os.path.join(ROOT, "tools", "gemm_tune.py")); gt = importlib.util.module_from_spec(spec); spec.loader.exec_module(gt)
lib = L.load()
table = {}
for l in open(os.path.join(ROOT, "profiles", "r04_gemm_tune_final.txt")):
    m = re.match(r"(\S+)\s+(fwd|bwd)\s+[\d.]+ GF\s+auto\s+([\d.]+) us.*?\|\s*(.*?)\s*\|\|", l)
    if m:
        table[(m.group(1), m.group(2))] = (float(m.group(3)), dict((k, float(v)) for k, v in (x.split(":") for x in m.group(4).split())))
out = (ctypes.c_int * 3)()
for name, Cx, Cs, up, H, W, Cout, k, s, p, pm in gt.LAYERS:
    Cin = Cx + Cs
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    pp = p if pm == 1 else 0
    for kind, rows, cols, K, cin_eff in (("fwd", 2 * Ho * Wo, Cout, k * k * Cin, Cin), ("bwd", 2 * (H + 2 * pp) * (W + 2 * pp), Cin, k * k * Cout, Cout)):
        if kind == "bwd" and s == 2:
            continue
        cb = 32 if (cin_eff % 32 == 0 and (Cs == 0 or Cx % 32 == 0 or kind == "bwd")) else 16
        lib.e2e_conv_gemm_choice(rows, cols, K, cb, 1, out)
        key = f"{out[0]}x{out[1]}/{out[2]}" if out[2] >= 0 else f"sk{-out[2]}"
        auto_old, top = table.get((name, kind), (None, {}))
        print(f"{name:10s} {kind} rows {rows:7d} cols {cols:4d} K {K:5d} cb {cb}: picks {key:10s} table: {top.get(key, 'not among the six best')}  (old auto {auto_old}, best {min(top.values()) if top else None})")
