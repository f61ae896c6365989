"""CPU-side checks of the C-ABI boundary: the library loads without a GPU and exports every symbol
include/e2eslam.h declares; the ctypes table covers the same set; CPU tensors are refused."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "e2eslam.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(e2e_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    import e2ehip
    from e2ehip import _lib
    lib = e2ehip.load()
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/e2eslam.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    assert lib.e2e_version() >= 100


def test_no_cpu_fallback():
    from e2ehip import E2EError, ops
    with pytest.raises(E2EError):
        ops.ssim(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))
    with pytest.raises(E2EError):
        ops.grid_sample(torch.rand(1, 3, 8, 8), torch.zeros(1, 8, 8, 2))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "end-to-end-self-supervised-slam_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product code imports the oracle: {bad}"
