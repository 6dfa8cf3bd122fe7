"""CPU-side checks of the drop-in boundary: the shared object loads, exports every symbol that
include/rec_pangu_hip.h declares (and nothing undeclared), the ctypes table covers them all, argument
validation works without a GPU, and no product module reaches into oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "rec_pangu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_exactly_the_header():
    from rec_pangu_amd import hip
    assert os.path.exists(hip.LIB_PATH), "build with __graft_entry__.build() / make -C rec_pangu_amd/csrc"
    declared = _declared()
    assert declared, "no declarations parsed"
    out = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (rp_[a-z0-9_]+)", out)))
    assert exported == declared
    assert sorted(hip.EXPORTED_SYMBOLS) == declared, "ctypes signature table out of sync with the header"
    lib = hip.lib()
    for name in declared:
        assert getattr(lib, name) is not None


def test_argument_validation_needs_no_gpu():
    from rec_pangu_amd import hip
    lib = hip.lib()
    from rec_pangu_amd import hip as _hip
    assert lib.rp_version() == _hip.ABI_VERSION  # (the bindings refuse a library of another version)
    n = ctypes.c_size_t(0)
    assert lib.rp_linear_wgrad_workspace_bytes(65536, 64, 1677, ctypes.byref(n)) == 0 and n.value > 0
    assert lib.rp_loss_partials(65536) >= 1
    # null pointers / bad sizes are refused with an error code and a message, never a crash
    rc = lib.rp_linear_fwd(None, 0, None, 0, None, None, 0, 1, 1, 1, 0, None, 0, None)
    assert rc == -1 and b"null" in lib.rp_last_error()
    rc = lib.rp_embed_gather_fwd(None, None, None, None, 0, None, 0, 1, 1, None, 0, None, None, None, None, None)
    assert rc == -1
    rc = lib.rp_adam_step(None, None, None, None, None, 0, 0.0, 0.0, 0.0, 0.0, 1, 0, None, None, None)
    assert rc == -1


def test_missing_library_fails_loudly(monkeypatch):
    from rec_pangu_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/librecpangu_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        hip.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rec_pangu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports oracle/"
                assert "ref_ops" not in src, f"{f} references the oracle"


def test_cin_pair_gather_lists_cover_every_pair_once_per_member():
    """Host logic of rp_cin_pair_bwd_x (rec_pangu_amd.hip.cin_pair_lists): for every field h the lists of all pair-tile
    halves together name each pair containing h exactly once per membership — the diagonal pair (h,h) twice — with
    the right partner and the right local row; and the forward's pair order is the row-major upper triangle."""
    import torch
    from rec_pangu_amd import hip
    for H in (1, 2, 5, 26, 32):
        lstart, lent = hip.cin_pair_lists(H, torch.device("cpu"))
        lstart, lent = lstart.tolist(), lent.tolist()
        pairs = [(h, m) for h in range(H) for m in range(h, H)]
        ntile = (len(pairs) + 127) // 128
        assert len(lstart) == 2 * ntile * H + 1
        seen = {h: [] for h in range(H)}
        for tile in range(ntile):
            for half in range(2):
                lo = tile * 128 + half * 64
                for h in range(H):
                    li = (2 * tile + half) * H + h
                    for e in lent[lstart[li]:lstart[li + 1]]:
                        pl, m = e & 255, e >> 8
                        assert 0 <= pl < 64 and lo + pl < len(pairs)
                        assert sorted(pairs[lo + pl]) == sorted((h, m)), (H, tile, half, h, pl, m)
                        seen[h].append(lo + pl)
        for h in range(H):
            want = sorted([p for p, (a, b) in enumerate(pairs) if a == h or b == h] +
                          [p for p, (a, b) in enumerate(pairs) if a == h and b == h])
            assert sorted(seen[h]) == want, (H, h)
    # symmetric pair weights: row-major upper triangle, off-diagonal entries summed
    W = torch.arange(2 * 3 * 3, dtype=torch.float32).view(2, 3, 3)
    ws = hip._cin_pair_ws(W)
    ref = torch.stack([torch.stack([W[o, 0, 0], W[o, 0, 1] + W[o, 1, 0], W[o, 0, 2] + W[o, 2, 0], W[o, 1, 1],
                                    W[o, 1, 2] + W[o, 2, 1], W[o, 2, 2]]) for o in range(2)])
    assert torch.equal(ws, ref)
