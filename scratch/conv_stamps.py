import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")]
from e2ehip import nn_ops
DEV = "cuda:0"
def run(B, C, H, W, Co):
    x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, C, 3, 3, device=DEV) * 0.05
    st = torch.zeros(64 * 256, dtype=torch.int64, device=DEV)
    with torch.no_grad():
        for _ in range(3): nn_ops.conv2d(x, w, None, 1, 1, "zeros", "relu", None, None, None, 1, None)
        torch.cuda.synchronize()
        os.environ["E2E_CONV_STAMPS"] = hex(st.data_ptr())
        nn_ops.conv2d(x, w, None, 1, 1, "zeros", "relu", None, None, None, 1, None)
        torch.cuda.synchronize()
        os.environ.pop("E2E_CONV_STAMPS")
    s = st.cpu().numpy().reshape(64, 256)
    print(f"== B{B} C{C} {H}x{W} Co{Co}  tiles={(B*H*W+63)//64*((Co+63)//64)}")
    for wg in (0, 1, 17, 40):
        n = int(s[wg, 255]); t = s[wg, :n].astype(np.int64)
        t0 = t[0]; body = t[1:-1].reshape(-1, 4)   # top, after load issue, after mfma issue, after store
        d_load = body[:, 1] - body[:, 0]; d_mfma = body[:, 2] - body[:, 1]; d_store = body[:, 3] - body[:, 2]
        d_bar = np.append(body[1:, 0], t[-1]) - body[:, 3]
        print(f" wg{wg}: total {t[-1]-t0} ticks over {len(body)} chunks | per chunk median: load-issue {np.median(d_load):.0f}  reads+mfma {np.median(d_mfma):.0f}  vmwait+store {np.median(d_store):.0f}  barrier {np.median(d_bar):.0f}")
        print("    first 6 chunks:", [tuple(int(v) for v in (a, b, c, d)) for a, b, c, d in zip(d_load[:6], d_mfma[:6], d_store[:6], d_bar[:6])])
run(2, 64, 128, 64, 64)
run(2, 64, 120, 160, 64)
run(2, 256, 30, 40, 256)
