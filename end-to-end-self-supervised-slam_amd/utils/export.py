"""Map export (the reference shows maps through plotly / open3d, utils/advanced_vis.py).
`save_ply` writes the fused map as a binary little-endian PLY that open3d / MeshLab / CloudCompare read."""
import numpy as np
import torch


def save_ply(path, points, colors=None, normals=None):
    """points (N,3) float; colors (N,3) in 0..255 or 0..1; normals (N,3).  Tensors may live on the GPU."""
    def host(t):
        return None if t is None else (t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t))
    p, c, n = host(points), host(colors), host(normals)
    if p.ndim != 2 or p.shape[1] != 3:
        raise ValueError(f"points: expected (N,3), got {p.shape}")
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if n is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    if c is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    rec = np.empty(p.shape[0], dtype=fields)
    rec["x"], rec["y"], rec["z"] = p[:, 0], p[:, 1], p[:, 2]
    if n is not None:
        rec["nx"], rec["ny"], rec["nz"] = n[:, 0], n[:, 1], n[:, 2]
    if c is not None:
        c = c if c.max() > 1.0 else c * 255.0
        c8 = np.clip(np.rint(c), 0, 255).astype(np.uint8)
        rec["red"], rec["green"], rec["blue"] = c8[:, 0], c8[:, 1], c8[:, 2]
    names = {"<f4": "float", "u1": "uchar"}
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {p.shape[0]}"]
    header += [f"property {names[t]} {k}" for k, t in fields] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())
    return path


def load_ply(path):
    """Reader for the files save_ply writes (round-trip tests)."""
    with open(path, "rb") as f:
        fields, n = [], 0
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property"):
                _, t, k = line.split()
                fields.append((k, "<f4" if t == "float" else "u1"))
            elif line == "end_header":
                break
        return np.frombuffer(f.read(), dtype=fields, count=n)
