"""GPU parity tests: every C-ABI entry point (through rec_pangu_amd.hip = ctypes over
include/rec_pangu_hip.h) against the CPU oracle (oracle/ref_ops.py) on the same seeded inputs.

Bars (BASELINE.json north_star): index work bit-exact; floating point within 1e-4 of the fp32
reference on logits/loss — block-level checks here are held tighter (1e-5 relative to the operand
scale) because the kernels are fp32 FMA chains like ATen's.
"""
import pytest
import torch

from conftest import require_gpu
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    require_gpu()
    from rec_pangu_amd import hip as h
    h.lib()
    return h


DEV = "cuda"
# the public Criteo-Kaggle cardinalities
CRITEO = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652,
          2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def _tables(rows, D, gen):
    arena = torch.randn(sum(rows), D, generator=gen)
    base, off = [], 0
    for r in rows:
        base.append(off)
        off += r
    return arena, base


def _gather_case(hip, rows, D, ND, B, seed, pad_to=32, want_fm=True):
    g = torch.Generator().manual_seed(seed)
    F = len(rows)
    arena, base = _tables(rows, D, g)
    idx = [torch.randint(0, r, (B,), generator=g) for r in rows]
    dense = [torch.rand(B, generator=g) for _ in range(ND)]
    d = F * D + ND
    ldx = (d + pad_to - 1) // pad_to * pad_to
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    x, fm, ssum, keys = hip.embed_gather_fwd(
        arena.to(DEV), torch.tensor(base, device=DEV), torch.tensor(rows, device=DEV), [i.to(DEV) for i in idx],
        [t.to(DEV) for t in dense], ldx, want_fm, want_fm, True, err)
    torch.cuda.synchronize()
    # oracle
    emb = torch.stack([arena[base[f] + idx[f]] for f in range(F)], dim=1)  # == R.embedding_all on per-table views
    assert int(err.item()) == 0
    x = x.cpu()
    assert torch.equal(x[:, :F * D], emb.flatten(1)), "gathered rows must be bit-exact copies"
    if ND:
        assert torch.equal(x[:, F * D:d], torch.stack(dense, dim=1)), "dense columns must be exact copies"
    assert torch.count_nonzero(x[:, d:]) == 0, "padding columns must be zero"
    exp_keys = torch.cat([base[f] + idx[f] for f in range(F)]).to(torch.int32)
    assert torch.equal(keys.cpu(), exp_keys), "arena row keys (index work) must be bit-exact"
    if want_fm:
        ref_fm = R.fm_second_order(emb)
        scale = (emb.abs().sum(dim=(1, 2)) ** 2).clamp(min=1.0).unsqueeze(1) * 1e-6
        assert ((fm.cpu() - ref_fm).abs() <= scale + 1e-5).all(), (fm.cpu() - ref_fm).abs().max()
        torch.testing.assert_close(ssum.cpu(), emb.sum(dim=1), rtol=1e-5, atol=1e-5)
    return arena, base, idx, x, keys


@pytest.mark.parametrize("rows,D,ND,B", [
    ([8, 4, 51, 12, 3], 8, 3, 24),               # the golden-fixture shape
    ([1000] * 26, 64, 13, 1000),                 # Criteo field/dim shape, ragged batch
    ([3, 4, 10, 100000, 27], 64, 13, 4099),      # tiny hot tables next to a big one
    ([17, 5], 40, 9, 333),                       # D=40 (MMOE default): 10 of 16 lanes active
    ([50, 7, 9], 1, 2, 777),                     # D=1: the LR_Layer scalar path
    ([50, 7], 6, 0, 65),                         # D not a multiple of 4 -> scalar lanes
    ([9], 256, 1, 130),                          # one wave per row
    ([9, 5], 320, 0, 70),                        # D > 256: lanes loop over chunks
])
def test_embed_gather_fwd(hip, rows, D, ND, B):
    _gather_case(hip, rows, D, ND, B, seed=D * 7 + B)


def test_embed_gather_flags_out_of_range(hip):
    g = torch.Generator().manual_seed(0)
    rows, D, B = [5, 6], 8, 40
    arena, base = _tables(rows, D, g)
    idx = [torch.randint(0, r, (B,), generator=g) for r in rows]
    idx[1][7] = 6  # == row_count -> out of range
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.embed_gather_fwd(arena.to(DEV), torch.tensor(base, device=DEV), torch.tensor(rows, device=DEV),
                         [i.to(DEV) for i in idx], [], 16, False, False, False, err)
    assert int(err.item()) == 1
    idx[1][7] = -1
    err.zero_()
    hip.embed_gather_fwd(arena.to(DEV), torch.tensor(base, device=DEV), torch.tensor(rows, device=DEV),
                         [i.to(DEV) for i in idx], [], 16, False, False, False, err)
    assert int(err.item()) == 1


# (n, hi): one pair / one wave / tile borders (4096-pair tiles, 1024-pair wave spans) / 1, 2, 3 and 4 digit passes of the
# 9-bit radix (hi = 2^9, 2^9+1, 2^18+1, 2^27+1) / two keys only (every lane of a wave on one counter) / the DeepFM batch
SORT_CASES = [(1, 5), (63, 3), (64, 2), (65, 600), (1000, 7), (4095, 512), (4096, 513), (4097, (1 << 18) + 1),
              (5121, 2), (100001, 1 << 20), (262144 + 1023, (1 << 27) + 1), (1703936, 33762603)]


@pytest.mark.parametrize("n,hi", SORT_CASES)
def test_sort_pairs(hip, n, hi):
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(0, hi, (n,), generator=g, dtype=torch.int32)
    ko, po = hip.sort_pairs(keys.to(DEV), end_bit=max(1, (hi - 1).bit_length()))
    ref_k, ref_p = torch.sort(keys, stable=True)
    assert torch.equal(ko.cpu(), ref_k)
    assert torch.equal(po.cpu(), ref_p.to(torch.int32)), "sort must be stable (ascending position among equal rows)"


def test_sort_pairs_field_ordered_keys_and_full_width(hip):
    """the keys the gather backward sorts are 26 runs of one table's range each (field-major positions): most tiles hold a
    handful of digits in the upper passes.  And all 32 bits = signed order (negative keys first)."""
    g = torch.Generator().manual_seed(5)
    rows = [r // 16 + 1 for r in CRITEO]
    base = 0
    parts = []
    for r in rows:
        parts.append(base + torch.randint(0, r, (8192,), generator=g))
        base += r
    keys = torch.cat(parts).to(torch.int32)
    ko, po = hip.sort_pairs(keys.to(DEV), end_bit=(base - 1).bit_length())
    ref_k, ref_p = torch.sort(keys, stable=True)
    assert torch.equal(ko.cpu(), ref_k) and torch.equal(po.cpu(), ref_p.to(torch.int32))
    # end_bit below the keys' width: only the low bits order the pairs (rocPRIM's begin/end-bit meaning), stably
    ko, po = hip.sort_pairs(keys.to(DEV), end_bit=11)
    ref_p = torch.sort(keys & 2047, stable=True)[1]
    assert torch.equal(po.cpu(), ref_p.to(torch.int32)) and torch.equal(ko.cpu(), keys[ref_p])
    signed = torch.randint(-2 ** 31, 2 ** 31 - 1, (70001,), generator=g, dtype=torch.int64).to(torch.int32)
    ko, po = hip.sort_pairs(signed.to(DEV), end_bit=32)
    ref_k, ref_p = torch.sort(signed, stable=True)
    assert torch.equal(ko.cpu(), ref_k) and torch.equal(po.cpu(), ref_p.to(torch.int32))
    # persistent output buffers (the captured step's), twice: nothing is carried between calls
    out = (torch.empty_like(ko), torch.empty_like(po))
    for _ in range(2):
        hip.sort_pairs(signed.to(DEV), end_bit=32, out=out)
        assert torch.equal(out[0].cpu(), ref_k) and torch.equal(out[1].cpu(), ref_p.to(torch.int32))


@pytest.mark.parametrize("rows,B", [
    ([r // 16 + 1 for r in CRITEO], 8192),                 # the Criteo field structure: 1, 2 and 3 passes per field
    (list(CRITEO), 65536),                                 # the headline shape: 1.7 M pairs, full cardinalities
    ([1, 2, 513, 262145, 7, 300000], 4096 + 77),           # ragged tiles; tables right at the 9- and 18-bit borders; one row
    ([40, 150_000_000, 9, 600], 5000),                     # a table of more than 2^27 rows: FOUR passes
    ([5], 3), ([1000], 1), ([3, 3], 70001),
])
def test_sort_pairs_fields_equals_the_plain_sort(hip, rows, B):
    """rp_sort_pairs_fields_i32 (round 6): the pair list of one lookup sorted field segment by field segment, every field by
    the bits of ITS table — identical integers to the stable sort of the whole list by arena row (torch.sort, and
    rp_sort_pairs_i32), into fresh buffers and into persistent ones twice (nothing is carried between calls)."""
    g = torch.Generator().manual_seed(B + len(rows))
    base, parts = 0, []
    for r in rows:
        p = base + torch.randint(0, r, (B,), generator=g)
        p[0], p[-1] = base + r - 1, base  # the last and the first row of every table are there
        parts.append(p)
        base += r
    keys = torch.cat(parts).to(torch.int32)
    ref_k, ref_p = torch.sort(keys, stable=True)
    ko, po = hip.sort_pairs_fields(keys.to(DEV), B, rows)
    assert torch.equal(ko.cpu(), ref_k) and torch.equal(po.cpu(), ref_p.to(torch.int32))
    k2, p2 = hip.sort_pairs(keys.to(DEV), end_bit=max(1, (base - 1).bit_length()))
    assert torch.equal(ko, k2) and torch.equal(po, p2)
    out = (torch.empty_like(ko), torch.empty_like(po))
    ws = hip.sort_fields_workspace(B, len(rows), DEV)
    kd = keys.to(DEV)
    for _ in range(2):
        hip.sort_pairs_fields(kd, B, rows, out=out, workspace=ws)
        assert torch.equal(out[0].cpu(), ref_k) and torch.equal(out[1].cpu(), ref_p.to(torch.int32))


def test_sort_pairs_rocprim_path():
    """RP_SORT=rocprim (read once per process) keeps rocPRIM's radix sort selectable: same results"""
    import subprocess, sys, os
    code = (
        "import torch\n"
        "from rec_pangu_amd import hip\n"
        "g = torch.Generator().manual_seed(3)\n"
        "for n, hi in [(1000, 7), (100001, 1 << 20), (1703936, 33762603)]:\n"
        "    keys = torch.randint(0, hi, (n,), generator=g, dtype=torch.int32)\n"
        "    ko, po = hip.sort_pairs(keys.cuda(), end_bit=(hi - 1).bit_length())\n"
        "    rk, rp = torch.sort(keys, stable=True)\n"
        "    assert torch.equal(ko.cpu(), rk) and torch.equal(po.cpu(), rp.to(torch.int32))\n"
        "print('ok')\n")
    env = dict(os.environ, RP_SORT="rocprim")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("rows,D,B,with_fm,with_dx", [
    ([8, 4, 51, 12, 3], 8, 24, True, True),
    ([3, 4, 10, 5000, 27], 64, 4099, True, True),     # runs of >1000 equal rows: pieces chained across workgroups
    ([3, 4, 10, 5000, 27], 64, 4099, False, True),    # DCN-like: no FM term
    ([17, 5], 40, 333, True, False),                  # FM-only model: dx absent
    ([50, 7, 9], 1, 777, False, True),                # LR tables
])
def test_embed_grad_reduce_vs_autograd(hip, rows, D, B, with_fm, with_dx):
    g = torch.Generator().manual_seed(B + D)
    F = len(rows)
    arena, base = _tables(rows, D, g)
    idx = [torch.randint(0, r, (B,), generator=g) for r in rows]
    ldx = (F * D + 3 + 31) // 32 * 32
    dx = torch.randn(B, ldx, generator=g)
    gfm = torch.randn(B, 1, generator=g)
    # oracle: autograd through gather (+FM) on CPU
    w = arena.clone().requires_grad_(True)
    emb = torch.stack([w[base[f] + idx[f]] for f in range(F)], dim=1)
    obj = 0
    if with_dx:
        obj = obj + (emb.flatten(1) * dx[:, :F * D]).sum()
    if with_fm:
        obj = obj + (R.fm_second_order(emb) * gfm).sum()
    obj.backward()
    ref = w.grad
    keys = torch.cat([base[f] + idx[f] for f in range(F)]).to(torch.int32).to(DEV)
    sk, sp = hip.sort_pairs(keys, end_bit=max(1, (sum(rows) - 1).bit_length()))
    G = torch.zeros_like(arena, device=DEV)
    ssum = emb.detach().sum(dim=1).to(DEV)
    hip.embed_grad_reduce(sk, sp, B, D, dx.to(DEV) if with_dx else None, gfm.to(DEV) if with_fm else None,
                          ssum if with_fm else None, arena.to(DEV), G, accumulate=False)
    tol = 1e-5 * max(1.0, float(ref.abs().max()))
    assert (G.cpu() - ref).abs().max() <= 20 * tol, (G.cpu() - ref).abs().max()
    untouched = torch.ones(sum(rows), dtype=torch.bool)
    untouched[keys.cpu().long()] = False
    assert torch.count_nonzero(G.cpu()[untouched]) == 0, "rows nobody looked up must stay exactly zero"
    # accumulate mode adds a second copy; zero_rows restores the all-zero invariant
    hip.embed_grad_reduce(sk, sp, B, D, dx.to(DEV) if with_dx else None, gfm.to(DEV) if with_fm else None,
                          ssum if with_fm else None, arena.to(DEV), G, accumulate=True)
    assert (G.cpu() - 2 * ref).abs().max() <= 40 * tol
    hip.zero_rows(sk, D, G)
    assert torch.count_nonzero(G) == 0


@pytest.mark.parametrize("M,N,K,lda_pad,act", [
    (24, 16, 43, 64, "relu"),
    (1000, 64, 1677, 1696, "relu"),     # DeepFM first layer, padded A rows, unaligned W rows
    (777, 64, 64, 64, "relu"),
    (513, 1, 64, 64, "none"),           # the 64 -> 1 head
    (300, 1677, 64, 64, "none"),        # dgrad shape: wide N, short K
    (129, 100, 37, 37, "none"),         # nothing aligned
    (256, 512, 649, 672, "none"),       # MMOE expert GEMM shape
])
@pytest.mark.parametrize("mode", ["bf16x6", "fp32", "bf16x3", "bf16"])
def test_linear_fwd(hip, M, N, K, lda_pad, act, mode):
    """every matrix-core mode against an fp64 product; the tolerance is the mode's contract (include/rec_pangu_hip.h):
    bf16x6 is held to the same bound as the exact f32 MFMA."""
    hip.set_matmul_precision(mode)
    try:
        assert hip.get_matmul_precision() == mode
        _linear_fwd_case(hip, M, N, K, lda_pad, act, *{"fp32": (1e-5, 2e-5), "bf16x6": (1e-5, 2e-5),
                                                       "bf16x3": (1e-4, 2e-4), "bf16": (5e-2, 5e-2)}[mode])
    finally:
        hip.set_matmul_precision("auto")


def _linear_fwd_case(hip, M, N, K, lda_pad, act, rtol, atol):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.zeros(M, lda_pad)
    a[:, :K] = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = a[:, :K].double() @ w.double().t() + b.double()
    if act == "relu":
        ref = ref.relu()
    out = hip.linear_fwd(a.to(DEV), w.to(DEV), b.to(DEV), hip.ACT_RELU if act == "relu" else hip.ACT_NONE, K=K)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=rtol, atol=atol)
    # transpose-detecting mask epilogue (asymmetric aux)
    aux = torch.randn(M, N, generator=g)
    out2 = hip.linear_fwd(a.to(DEV), w.to(DEV), None, hip.ACT_MASK, aux=aux.to(DEV), K=K)
    ref2 = (a[:, :K].double() @ w.double().t()) * (aux > 0)
    torch.testing.assert_close(out2.cpu().double(), ref2, rtol=rtol, atol=atol)


@pytest.mark.parametrize("M,N,K,ldx_pad", [
    (24, 16, 43, 64), (4099, 64, 1677, 1696), (65536, 64, 64, 64), (1000, 1, 64, 64), (513, 100, 37, 37),
    (2048, 512, 649, 672),
])
@pytest.mark.parametrize("mode", ["bf16x6", "fp32", "bf16x3", "bf16"])
def test_linear_wgrad(hip, M, N, K, ldx_pad, mode):
    hip.set_matmul_precision(mode)
    try:
        _linear_wgrad_case(hip, M, N, K, ldx_pad, {"fp32": 1.0, "bf16x6": 1.0, "bf16x3": 20.0, "bf16": 3000.0}[mode])
    finally:
        hip.set_matmul_precision("auto")


def _linear_wgrad_case(hip, M, N, K, ldx_pad, slack):
    g = torch.Generator().manual_seed(M + N + K + 1)
    x = torch.zeros(M, ldx_pad)
    x[:, :K] = torch.randn(M, K, generator=g)
    dy = torch.randn(M, N, generator=g)
    dw, db = hip.linear_wgrad(dy.to(DEV), x.to(DEV), K)
    ref_w = dy.double().t() @ x[:, :K].double()
    ref_b = dy.double().sum(dim=0)
    tol = 2e-6 * M ** 0.5 * 4 * slack  # fp32 accumulation over M terms (x the mode's product error)
    torch.testing.assert_close(dw.cpu().double(), ref_w, rtol=1e-4 * slack, atol=tol * 10)
    torch.testing.assert_close(db.cpu().double(), ref_b, rtol=1e-4, atol=tol * 10)
    # accumulate=True adds on top
    dw2, db2 = hip.linear_wgrad(dy.to(DEV), x.to(DEV), K, dw=dw.clone(), db=db.clone(), accumulate=True)
    torch.testing.assert_close(dw2.cpu().double(), 2 * ref_w, rtol=1e-4 * slack, atol=tol * 20)
    torch.testing.assert_close(db2.cpu().double(), 2 * ref_b, rtol=1e-4, atol=tol * 20)


def test_transpose_and_relu_bwd(hip):
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 1677, generator=g)
    assert torch.equal(hip.transpose(w.to(DEV)).cpu(), w.t().contiguous())
    w = torch.randn(37, 5, generator=g)
    assert torch.equal(hip.transpose(w.to(DEV)).cpu(), w.t().contiguous())
    dy, y = torch.randn(1001, 64, generator=g), torch.randn(1001, 64, generator=g).relu()
    assert torch.equal(hip.relu_bwd(dy.to(DEV), y.to(DEV)).cpu(), dy * (y > 0))


def test_transpose_copy_counters_and_tail_parts(hip):
    """the small launches the captured step merges or splits (round 4): rp_transpose_copy = rp_transpose + rp_copy_rows of
    one matrix; rp_counters_add = several rp_counter_add; rp_mlp_tail_bwd_parts(1) then (2) = rp_mlp_tail_bwd"""
    g = torch.Generator().manual_seed(5)
    for R, Cc, rows_out in ((64, 1677, 1728), (37, 5, 0), (64, 64, 64)):
        w = torch.randn(R, Cc, generator=g).to(DEV)
        ld = (Cc + 3) // 4 * 4
        wt, w16 = hip.transpose_copy(w, rows_out, ld)
        ref_t = hip.transpose(w, rows_out=rows_out)
        assert wt.shape == ref_t.shape and torch.equal(wt, ref_t)
        assert torch.equal(w16, w) and w16.stride(0) == ld
    cs = [torch.full((1,), v, dtype=torch.int32, device=DEV) for v in (3, 10, -2)]
    hip.counters_add(cs, 2)
    hip.counters_add(cs[:1], 1)
    assert [int(c.item()) for c in cs] == [6, 12, 0]
    M, L = 1000, 2
    Ws = [torch.randn(64, 64, generator=g).to(DEV) / 8 for _ in range(L)]
    acts = [torch.randn(M, 64, generator=g).relu().to(DEV) for _ in range(L + 1)]
    dz, w_out = torch.randn(M, generator=g).to(DEV), torch.randn(1, 64, generator=g).to(DEV)
    a = hip.mlp_tail_bwd(dz, Ws, acts, w_out)   # (outside a launch plan: one call, both parts)
    import ctypes as C
    dhin = torch.empty(M, 64, device=DEV)
    grads = torch.empty(L * 4096 + L * 64 + 65, device=DEV)
    nbytes = C.c_size_t(0)
    assert hip.lib().rp_mlp_tail_bwd_workspace_bytes(M, L, C.byref(nbytes)) == 0
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=DEV)
    for parts in (1, 2):
        rc = hip.lib().rp_mlp_tail_bwd_parts(dz.data_ptr(), L, hip._ptr_array(Ws), hip._i64_array([64] * L), hip._ptr_array(acts), 64,
                                             w_out.data_ptr(), dhin.data_ptr(), 64, grads.data_ptr(), M, ws.data_ptr(), nbytes.value,
                                             parts, hip._stream())
        assert rc == 0
    assert torch.equal(dhin, a[0])
    assert torch.equal(grads[:4096].view(64, 64), a[1][0]) and torch.equal(grads[L * 4160:L * 4160 + 64].view(1, 64), a[3])


@pytest.mark.parametrize("B,n_add,apply_sigmoid,p_eps,weight", [
    (24, 2, True, 0.0, 1.0), (65536, 3, True, 0.0, 1.0), (1000, 1, False, 1e-6, 0.5), (300001, 1, True, 0.0, 1.0)])
def test_sigmoid_bce(hip, B, n_add, apply_sigmoid, p_eps, weight):
    g = torch.Generator().manual_seed(B)
    adds = [torch.randn(B, 1, generator=g) * 2 for _ in range(n_add)]
    if not apply_sigmoid:
        adds = [torch.rand(B, 1, generator=g) * 0.98 + 0.01]
    adds[0][0] = 40.0 if apply_sigmoid else adds[0][0]  # saturated logit: log clamp path
    y = (torch.rand(B, generator=g) < 0.3).float()
    zs = [a.clone().requires_grad_(True) for a in adds]
    z = sum(zs)
    p = torch.sigmoid(z) if apply_sigmoid else z
    ref_loss = weight * torch.nn.functional.binary_cross_entropy(p.squeeze(-1) + p_eps, y)
    ref_loss.backward()
    pred, loss = hip.sigmoid_bce_fwd([a.to(DEV) for a in adds], y.to(DEV), apply_sigmoid, p_eps, weight)
    torch.testing.assert_close(pred.cpu(), p.detach(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(loss.cpu(), ref_loss.detach(), rtol=2e-6, atol=1e-7)
    dz = hip.sigmoid_bce_bwd(pred, y.to(DEV), torch.ones((), device=DEV), apply_sigmoid, p_eps, weight)
    torch.testing.assert_close(dz.cpu(), zs[0].grad, rtol=1e-4, atol=1e-9)
    pred_only, none = hip.sigmoid_bce_fwd([a.to(DEV) for a in adds], None, apply_sigmoid)
    assert none is None and torch.equal(pred_only, pred)


def test_adam_matches_reference_order(hip):
    g = torch.Generator().manual_seed(9)
    shapes = [(1000, 64), (64,), (7, 13), (1,), (33762 * 64 + 3,)]
    p = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) * (0.0 if i == 2 else 1.0) for s in shapes] for i in range(3)]
    ref = [t.clone() for t in p]
    opt = torch.optim.Adam([torch.nn.Parameter(t) for t in ref], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    dp = [t.to(DEV) for t in p]
    dm = [torch.zeros_like(t) for t in dp]
    dv = [torch.zeros_like(t) for t in dp]
    for step in range(1, 4):
        for q, gr in zip(opt.param_groups[0]["params"], grads[step - 1]):
            q.grad = gr.clone()
        opt.step()
        dg = [t.to(DEV) for t in grads[step - 1]]
        hip.adam_step(dp, dg, dm, dv, 1e-2, 0.9, 0.999, 1e-8, step, zero_grad=(step == 2))
        if step == 2:
            assert all(torch.count_nonzero(t) == 0 for t in dg), "fused zero_grad must clear g"
    for q, d in zip(opt.param_groups[0]["params"], dp):
        torch.testing.assert_close(d.cpu(), q.detach(), rtol=1e-5, atol=1e-6)


def test_full_size_gather_properties(hip):
    """BASELINE config 1 shape (B=65536, 26 fields, D=64) on a 1/16-vocabulary arena: size-independent
    properties instead of a CPU oracle — every gathered row is a bit-exact copy of its arena row and the
    backward is linear: column sums of the gradient arena equal column sums of dx."""
    g = torch.Generator().manual_seed(1)
    card = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306,
            10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
    rows = [max(2, c // 16) + 1 for c in card]
    F, D, B, ND = 26, 64, 65536, 13
    arena = torch.randn(sum(rows), D, device=DEV)
    base = [0]
    for r in rows[:-1]:
        base.append(base[-1] + r)
    idx = [torch.randint(0, r, (B,), generator=g).to(DEV) for r in rows]
    dense = [torch.rand(B, generator=g).to(DEV) for _ in range(ND)]
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    n0 = hip.launch_count()
    x, fm, ssum, keys = hip.embed_gather_fwd(arena, torch.tensor(base, device=DEV), torch.tensor(rows, device=DEV),
                                             idx, dense, 1696, True, True, True, err)
    assert hip.launch_count() == n0 + 1
    emb = x[:, :F * D].view(B, F, D)
    for f in (0, 2, 8, 25):
        assert torch.equal(emb[:, f], arena[base[f] + idx[f]])
    torch.testing.assert_close(fm, 0.5 * ((emb.sum(1) ** 2) - (emb ** 2).sum(1)).sum(-1, keepdim=True), rtol=1e-4,
                               atol=1e-2)
    dx = torch.randn(B, 1696, device=DEV)
    sk, sp = hip.sort_pairs(keys, end_bit=(sum(rows) - 1).bit_length())
    assert bool((sk[1:] >= sk[:-1]).all())
    G = torch.zeros_like(arena)
    hip.embed_grad_reduce(sk, sp, B, D, dx, None, None, None, G, accumulate=False)
    lhs = G.double().sum(dim=0)
    rhs = dx[:, :F * D].double().view(B, F, D).sum(dim=(0, 1))
    torch.testing.assert_close(lhs, rhs, rtol=2e-4, atol=5e-3)  # fp32 row sums of up to 20k addends per hot row


@pytest.mark.parametrize("world,rows,b", [(1, [5, 1000, 37], 300), (2, [3, 20000, 64, 7], 513), (8, [4, 100000, 900, 2, 50], 1000),
                                          (8, [1, 1], 64)])
def test_route_kernels_vs_torch(world, rows, b):
    """rp_shard_keys + rp_sort_pairs_i32 + rp_route_build against the torch formulation the CPU (gloo) path uses
    (sharded._Route): slots, per-pair slots, rows to request and per-owner counts are identical integers — including
    owners nobody asks (tiny tables, world 8) and an out-of-range id (flagged, row 0)."""
    from rec_pangu_amd import hip
    g = torch.Generator().manual_seed(world * 1000 + b)
    F = len(rows)
    idx = [torch.randint(0, r, (b,), generator=g) for r in rows]
    idx[0][3] = rows[0] + 5  # bad id
    base = torch.tensor([sum(rows[:f]) for f in range(F)], dtype=torch.int64)
    cnt = torch.tensor(rows, dtype=torch.int64)
    total = sum(rows)
    lbits = max(1, int((total + world - 1) // world).bit_length())
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    comp = hip.shard_keys(base.to(DEV), cnt.to(DEV), [t.to(DEV) for t in idx], world, lbits, err)
    assert int(err.item()) == 1
    # torch reference (bad id -> row 0 of its table)
    ids = torch.stack(idx)
    ids = torch.where((ids < 0) | (ids >= cnt[:, None]), torch.zeros_like(ids), ids)
    keys = (ids + base[:, None]).reshape(-1)
    comp_ref = ((keys % world) << lbits) | torch.div(keys, world, rounding_mode="floor")
    assert torch.equal(comp.cpu().long(), comp_ref)
    nbits = lbits + max(1, (world - 1).bit_length())
    sk, sp = hip.sort_pairs(comp, end_bit=nbits)
    slot_sorted, slot_of_pair, uniq_rows, counts = hip.route_build(sk, sp, world, lbits)
    rk, rp_ = torch.sort(comp_ref, stable=True)
    uniq, inverse = torch.unique_consecutive(rk, return_inverse=True)
    assert torch.equal(sk.cpu().long(), rk) and torch.equal(sp.cpu().long(), rp_)
    assert torch.equal(slot_sorted.cpu().long(), inverse)
    ref_sop = torch.empty_like(keys)
    ref_sop[rp_] = inverse
    assert torch.equal(slot_of_pair.cpu(), ref_sop)
    counts = counts.cpu()
    assert int(counts[world]) == uniq.numel()
    assert torch.equal(counts[:world], torch.bincount(uniq >> lbits, minlength=world))
    assert torch.equal(uniq_rows[:uniq.numel()].cpu(), uniq & ((1 << lbits) - 1))
    # rp_route_field_major (round 6): the sorted (slot, position) list moved into FIELD-major order = a stable sort of the
    # list by field: field f's b requests at [f b, (f + 1) b), inside a field ascending slots with the positions of a run in
    # their sorted order — identical integers
    if world > 1:
        slot_fm, pos_fm = hip.route_field_major(sk, sp, slot_sorted, b, world, lbits)
        field = torch.div(rp_, b, rounding_mode="floor")
        _, order = torch.sort(field, stable=True)
        assert torch.equal(pos_fm.cpu().long(), rp_[order]) and torch.equal(slot_fm.cpu().long(), inverse[order])
        assert torch.equal(torch.div(pos_fm.cpu().long(), b, rounding_mode="floor"), torch.arange(F).repeat_interleave(b))


@pytest.mark.parametrize("O,H", [(128, 26), (128, 6), (7, 32), (1, 1), (100, 13)])
def test_cin_pair_pieces_vs_torch(hip, O, H):
    """rp_cin_pair_pieces (round 6): the pair kernels' weights — triangular fold + three bf16 pieces, both layouts from one
    launch — against the torch formulation of rounds 3-5, bit for bit (same fp32 adds, round-to-nearest-even conversions)."""
    g = torch.Generator().manual_seed(O * 100 + H)
    W = (torch.randn(O, H, H, generator=g) * 0.3).to(DEV)
    wsp, wst = hip.cin_pair_pieces(W, both=True)
    ref_p, ref_t = hip.cin_pair_pieces_torch(W), hip.cin_pair_pieces_torch(W, transposed=True)
    assert wsp.shape == ref_p.shape and wst.shape == ref_t.shape
    assert torch.equal(wsp.view(torch.int16), ref_p.view(torch.int16)) and torch.equal(wst.view(torch.int16), ref_t.view(torch.int16))
    assert torch.equal(hip.cin_pair_pieces(W).view(torch.int16), ref_p.view(torch.int16))
    assert torch.equal(hip.cin_pair_pieces(W, transposed=True).view(torch.int16), ref_t.view(torch.int16))


@pytest.mark.parametrize("O,H,M,with_b", [(128, 26, 128, True), (8, 6, 8, True), (16, 32, 5, False), (1, 1, 1, True)])
def test_cin_head_params_vs_float64(hip, O, H, M, with_b):
    """rp_cin_head_params_fwd / _bwd, rp_add_scalars, rp_sum_all (round 6: the weight-space arithmetic of the CIN's collapsed last
    layer, interaction.py:157-171 behind sum-pooling and fc) against float64:  V^T = (c . W_L)^T zero padded to 32 columns,
    vb = c . b_L;  dW_L = c^T (x) dV,  db_L = D sg c,  dc = W_L . dV + D sg b_L."""
    g = torch.Generator().manual_seed(O + H + M)
    WL, bL, c = torch.randn(O, H * M, generator=g), torch.randn(O, generator=g), torch.randn(O, generator=g)
    dV, gvec, D = torch.randn(H, M, generator=g), torch.randn(5000, generator=g), 64
    b_dev = bL.to(DEV) if with_b else None
    vt, vb = hip.cin_head_params_fwd(WL.to(DEV), b_dev, c.to(DEV), H, M)
    V = (c.double() @ WL.double()).view(H, M)
    ref_vt = torch.zeros(M, 32, dtype=torch.float64)
    ref_vt[:, :H] = V.t()
    torch.testing.assert_close(vt.cpu().double(), ref_vt, rtol=0, atol=1e-5 * max(1.0, float(V.abs().max())))
    assert abs(float(vb) - (float(c.double() @ bL.double()) if with_b else 0.0)) <= 1e-5 * O
    sg = hip.sum_all(gvec.to(DEV))
    assert abs(float(sg) - float(gvec.double().sum())) <= 1e-4 * float(gvec.abs().sum()) / 100
    out = torch.ones(300, device=DEV)
    hip.add_scalars(out, vb, 2.0, sg)
    assert torch.allclose(out.cpu(), torch.full((300,), 1.0) + 2.0 * float(vb) + float(sg), rtol=1e-6, atol=1e-6)
    dWL, dbL, dc = hip.cin_head_params_bwd(WL.to(DEV), b_dev, c.to(DEV), dV.to(DEV), sg, D, H, M)
    s = float(sg)
    torch.testing.assert_close(dWL.cpu().double(), c.double()[:, None] * dV.double().reshape(1, -1), rtol=1e-6, atol=1e-6)
    ref_dc = WL.double() @ dV.double().reshape(-1) + (D * s * bL.double() if with_b else 0.0)
    torch.testing.assert_close(dc.cpu().double(), ref_dc, rtol=1e-5, atol=1e-4 * max(1.0, float(ref_dc.abs().max())))
    if with_b:
        torch.testing.assert_close(dbL.cpu().double(), D * s * c.double(), rtol=1e-5, atol=1e-5 * abs(D * s))
    else:
        assert dbL is None


@pytest.mark.parametrize("shape,p", [((4096, 64), 0.1), ((1000, 37), 0.5), ((65536, 256), 0.2), ((3, 5), 0.3)])
def test_dropout_kernels(shape, p):
    """rp_dropout_fwd/bwd (nn.Dropout of layers/deep.py:66-68 and the towers): y = x * keep / (1 - p) with a saved
    mask; exact arithmetic given the mask, the mask a pure function of (seed, offset), keep rate = 1 - p within 5 sigma,
    different offsets -> different masks, backward = dy * keep / (1 - p); strided (padded) inputs give the same mask."""
    from rec_pangu_amd import hip
    M, N = shape
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, N, generator=g).to(DEV) + 3.0  # (no exact zeros)
    y, mask = hip.dropout_fwd(x, p, seed=1234, offset=8)
    y2, mask2 = hip.dropout_fwd(x, p, seed=1234, offset=8)
    assert torch.equal(mask, mask2) and torch.equal(y, y2)
    scale = 1.0 / (1.0 - p)
    assert torch.equal(y, torch.where(mask.bool(), x * torch.tensor(scale, dtype=torch.float32), torch.zeros_like(x)))
    keep = float(mask.float().mean())
    sigma = (p * (1 - p) / (M * N)) ** 0.5
    assert abs(keep - (1 - p)) <= 5 * sigma + 1e-9, (keep, 1 - p, sigma)
    _, mask3 = hip.dropout_fwd(x, p, seed=1234, offset=12)
    if M * N > 100:
        assert not torch.equal(mask, mask3)
    xp = torch.zeros(M, N + 7, device=DEV)
    xp[:, :N] = x
    _, mask4 = hip.dropout_fwd(xp[:, :N], p, seed=1234, offset=8)  # unaligned rows: scalar path, same mask
    assert torch.equal(mask, mask4)
    dy = torch.randn(M, N, generator=g).to(DEV)
    dx = hip.dropout_bwd(dy, mask, p)
    assert torch.equal(dx, torch.where(mask.bool(), dy * torch.tensor(scale, dtype=torch.float32), torch.zeros_like(dy)))
    # columns are not correlated with rows (a Philox counter bug would show up as a periodic mask)
    if M >= 1000:
        col = mask.float().mean(0)
        assert float((col - (1 - p)).abs().max()) <= 6 * (p * (1 - p) / M) ** 0.5


def test_dropout_follows_torch_seed_and_trains_without_aten():
    """The autograd wrapper takes (seed, offset) from torch's device generator: torch.manual_seed reproduces a training
    step bit for bit, consecutive calls differ; an xDeepFM / MMOE train step (default dropout 0.1 / 0.2) runs with
    torch's own dropout disabled (the ATen kernels are not on the path)."""
    import torch.nn.functional as TF
    from rec_pangu_amd import functional as Fh
    from test_host_models import CASES, build
    from conftest import load_golden
    x = torch.randn(512, 64, device=DEV, requires_grad=True)
    torch.manual_seed(5)
    a = Fh.dropout(x, 0.3)
    b = Fh.dropout(x, 0.3)
    torch.manual_seed(5)
    a2 = Fh.dropout(x, 0.3)
    assert torch.equal(a, a2) and not torch.equal(a, b)
    a.sum().backward()
    assert torch.equal(x.grad != 0, a != 0)

    def boom(*args, **kw):
        raise AssertionError("torch's dropout was called on the HIP path")

    saved = (TF.dropout, torch.dropout, torch.nn.functional.dropout)
    TF.dropout = boom
    torch.dropout = boom
    try:
        for name in ("xdeepfm", "mmoe_eval", "autoint_h2"):
            g = load_golden(f"model_{name}.npz")
            model = build(name).to(DEV)
            model.train()
            batch = {k: v.to(DEV) for k, v in g["batch"].items()}
            torch.manual_seed(9)
            o1 = model(batch)
            o1["loss"].backward()
            g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            model.zero_grad()
            torch.manual_seed(9)
            o2 = model(batch)
            o2["loss"].backward()
            assert torch.equal(o1["loss"], o2["loss"])
            for k, p in model.named_parameters():
                if p.grad is not None:
                    assert torch.equal(p.grad, g1[k]), k
            assert torch.isfinite(o1["loss"])
    finally:
        TF.dropout, torch.dropout = saved[0], saved[1]


def test_embed_grad_reduce_is_deterministic_and_ordered(hip):
    """The segmented reduce has one writer per row and a fixed summation order (ascending sorted position = ascending
    sample index): (a) at the full Criteo shape (26 fields x 65536 samples, tiny tables -> runs of > 20000 equal rows
    crossing hundreds of workgroups) two launches give bit-identical gradient arenas, also from a non-zeroed arena in
    accumulate=0 mode; (b) on a small case the result equals a sequential left-to-right fp32 sum in sample order,
    bit for bit."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rows = [c // 16 + 1 for c in bench.CRITEO_CARD]
    F, D, B = len(rows), 64, 65536
    g = torch.Generator(device=DEV).manual_seed(3)
    base = torch.tensor([0] + list(torch.tensor(rows).cumsum(0)[:-1]), device=DEV)
    idx = torch.stack([torch.randint(0, r, (B,), generator=g, device=DEV) for r in rows])
    keys = (idx + base[:, None]).reshape(-1).to(torch.int32)
    ldx = (F * D + 13 + 63) // 64 * 64
    dx = torch.randn(B, ldx, generator=g, device=DEV)
    gfm = torch.randn(B, 1, generator=g, device=DEV)
    arena = torch.randn(sum(rows), D, generator=g, device=DEV)
    ssum = torch.randn(B, D, generator=g, device=DEV)
    sk, sp = hip.sort_pairs(keys, end_bit=max(1, (sum(rows) - 1).bit_length()))
    outs = []
    for trial in range(3):
        G = torch.zeros_like(arena) if trial < 2 else torch.full_like(arena, 7.0)  # accumulate=0 overwrites touched rows
        hip.embed_grad_reduce(sk, sp, B, D, dx, gfm, ssum, arena, G, accumulate=False)
        outs.append(G)
    assert torch.equal(outs[0], outs[1]), "two launches of the gradient reduce differ"
    touched = torch.zeros(sum(rows), dtype=torch.bool, device=DEV)
    touched[sk.long()] = True
    assert torch.equal(outs[2][touched], outs[0][touched]) and bool((outs[2][~touched] == 7.0).all())
    col = dx[:, :F * D].reshape(B, F, D).sum(dim=(0, 1)).double()
    got = outs[0].double().sum(0) - (gfm.double() * ssum.double()).sum(0) * F + \
        (arena.double() * torch.zeros(sum(rows), 1, device=DEV, dtype=torch.float64).index_add_(
            0, keys.long(), gfm.double().repeat(F, 1))).sum(0)
    assert float((got - col).abs().max()) <= 1e-3 * float(col.abs().max() + 1)  # column sums of G == column sums of dX
    # (b) exact order on a small case: no FM term, one table of 3 rows, 700 samples
    B2, D2 = 700, 8
    idx2 = torch.randint(0, 3, (B2,), generator=torch.Generator().manual_seed(1))
    dx2 = torch.randn(B2, D2, generator=torch.Generator().manual_seed(2))
    ref = torch.zeros(3, D2)
    for b in range(B2):  # sequential fp32 sum in sample order
        ref[idx2[b]] += dx2[b]
    sk2, sp2 = hip.sort_pairs(idx2.to(torch.int32).to(DEV), end_bit=2)
    G2 = torch.zeros(3, D2, device=DEV)
    hip.embed_grad_reduce(sk2, sp2, B2, D2, dx2.to(DEV), None, None, None, G2, accumulate=False)
    # the kernel sums 8 positions in a register chain, then pieces in segment order: a fixed order, but not the
    # purely sequential one — equal to it within fp32 reassociation error, and identical across launches
    torch.testing.assert_close(G2.cpu(), ref, rtol=1e-5, atol=1e-5)
    G3 = torch.zeros(3, D2, device=DEV)
    hip.embed_grad_reduce(sk2, sp2, B2, D2, dx2.to(DEV), None, None, None, G3, accumulate=False)
    assert torch.equal(G2, G3)


@pytest.mark.parametrize("M,N,K", [(65536, 1024, 1677), (1000, 300, 100), (4096 + 77, 256 + 32, 64), (513, 1677, 1024), (256, 256, 32)])
@pytest.mark.parametrize("np_,mode", [(2, "bf16x3"), (1, "bf16")])
def test_linear_fwd_pieces_bit_identical_to_the_in_kernel_split(hip, M, N, K, np_, mode):
    """rp_linear_fwd_pieces (csrc/gemm_pieces.hip: operands pre-split into bf16 pieces by rp_pieces_pack, k-tiles staged by
    LDS-DMA into an XOR-swizzled image) against rp_linear_fwd on the fp32 operands in the same product mode: the pieces are
    the same values and every accumulator sees the same MFMA sequence, so the outputs are EQUAL — ragged M / N (clamped
    rows, guarded stores), K tails that are zero padding in the piece layout, the mask epilogue, and the layout itself
    (hi | lo per 32 values, against a torch split)."""
    hip.set_matmul_precision(mode)
    try:
        g = torch.Generator(device=DEV).manual_seed(M + N + K + np_)
        ldx = (K + 3) // 4 * 4
        a = torch.zeros(M, ldx, device=DEV)
        a[:, :K] = torch.randn(M, K, generator=g, device=DEV)
        w = torch.zeros(N, ldx, device=DEV)
        w[:, :K] = torch.randn(N, K, generator=g, device=DEV) / K ** 0.5
        b = torch.randn(N, generator=g, device=DEV)
        ap, wp = hip.pieces_pack(a, np_, K=K), hip.pieces_pack(w, np_, K=K)
        assert ap.shape[1] == hip.pieces_ld(K, np_) and ap.shape[1] % 64 == 0
        # the layout, against torch: hi = RN(x), lo = RN(x - hi)
        hi = a[:, :K].to(torch.bfloat16)
        tiles = ap.view(M, -1, 64)
        if np_ == 2:
            lo = (a[:, :K] - hi.float()).to(torch.bfloat16)
            kp = tiles.shape[1] * 32
            want = torch.zeros(M, kp, 2, dtype=torch.bfloat16, device=DEV)
            want[:, :K, 0], want[:, :K, 1] = hi, lo
            want = want.view(M, kp // 32, 32, 2).permute(0, 1, 3, 2).reshape(M, -1, 64)
        else:
            want = torch.zeros(M, tiles.shape[1] * 64, dtype=torch.bfloat16, device=DEV)
            want[:, :K] = hi
            want = want.view(M, -1, 64)
        assert torch.equal(tiles, want)
        ref = hip.linear_fwd(a, w[:, :K], b, hip.ACT_RELU, K=K)
        n0 = hip.launch_count()
        out = hip.linear_fwd_pieces(ap, wp, b, K, np_, act=hip.ACT_RELU)
        assert hip.launch_count() == n0 + 1
        assert torch.equal(out, ref)
        aux = torch.randn(M, N, generator=g, device=DEV)
        ref2 = hip.linear_fwd(a, w[:, :K], None, hip.ACT_MASK, aux=aux, K=K)
        out2 = torch.full((M, N), float("nan"), device=DEV)
        hip.linear_fwd_pieces(ap, wp, None, K, np_, act=hip.ACT_MASK, aux=aux, out=out2)
        assert torch.equal(out2, ref2)
    finally:
        hip.set_matmul_precision("auto")


@pytest.mark.parametrize("M,N,K,lda_pad", [(65536, 1024, 1677, 1728), (65536, 384, 205, 208), (65536, 256, 192, 192),
                                           (131072, 512, 649, 704)])
@pytest.mark.parametrize("mode", ["bf16x3", "auto", "bf16"])
def test_linear_fwd_wide_kernel(hip, M, N, K, lda_pad, mode):
    """The 256 x 128 / 64 x 64-wave kernel that serves wide, matrix-core-bound layers in the two-piece modes (XCD-aware
    tile order, double-buffered LDS, register ring three k-tiles deep): full tiles, a partial last column block
    (N = 384), a K tail that is not a multiple of 32 (1677, 205, 649), a K with exactly six k-tiles (192: the shortest
    pipelined case); bias + ReLU and the mask epilogue, against an fp64 product on the device."""
    hip.set_matmul_precision(mode)
    try:
        g = torch.Generator(device=DEV).manual_seed(M + N + K)
        a = torch.zeros(M, lda_pad, device=DEV)
        a[:, :K] = torch.randn(M, K, generator=g, device=DEV)
        w = torch.randn(N, K, generator=g, device=DEV) / K ** 0.5
        wp = torch.zeros(N, (K + 3) // 4 * 4, device=DEV)
        wp[:, :K] = w
        b = torch.randn(N, generator=g, device=DEV)
        rtol, atol = {"bf16x3": (1e-4, 2e-4), "auto": (1e-4, 2e-4), "bf16": (5e-2, 5e-2)}[mode]
        rows = torch.randint(0, M, (4096,), generator=g, device=DEV)  # fp64 reference on a sample of rows (+ the ends)
        rows[:256] = torch.arange(256, device=DEV)
        rows[-256:] = torch.arange(M - 256, M, device=DEV)
        ref = a[rows, :K].double() @ w.double().t() + b.double()
        n0 = hip.launch_count()
        out = hip.linear_fwd(a, wp[:, :K], b, hip.ACT_RELU, K=K)
        assert hip.launch_count() == n0 + 1
        torch.testing.assert_close(out[rows].double(), ref.relu(), rtol=rtol, atol=atol)
        aux = torch.randn(M, N, generator=g, device=DEV)
        out2 = hip.linear_fwd(a, wp[:, :K], None, hip.ACT_MASK, aux=aux, K=K)
        ref2 = (a[rows, :K].double() @ w.double().t()) * (aux[rows] > 0)
        torch.testing.assert_close(out2[rows].double(), ref2, rtol=rtol, atol=atol)
        # every output element was written (no tile skipped by the XCD-aware workgroup -> tile mapping)
        out3 = torch.full((M, N), float("nan"), device=DEV)
        hip.linear_fwd(a, wp[:, :K], b, hip.ACT_NONE, K=K, out=out3)
        assert not bool(torch.isnan(out3).any())
    finally:
        hip.set_matmul_precision("auto")


@pytest.mark.parametrize("rows,B,with_fm,with_dx,accumulate,owners", [
    ([8, 4, 51, 12, 3], 24, True, False, False, 1),
    ([3, 4, 10, 5000, 27], 4099, True, False, False, 1),   # tiles that span field borders; runs of > 1000 equal rows
    ([3, 4, 10, 5000, 27], 4099, False, True, True, 1),    # extra per-pair gradient (another consumer of x), accumulate mode
    ([50000, 7], 65536, True, True, False, 1),             # full batch: mostly unique rows + one hot table
    # the row-sharded path's order: rows renumbered owner-major (r -> (r % G) * ceil(R / G) + r // G), so the sorted
    # positions are field-major only inside an owner and a tile at an owner border holds fields {.., F-1, 0, ..}
    ([8, 4, 51, 12, 3], 24, True, False, False, 8),        # every owner inside ONE tile
    ([3, 4, 10, 5000, 27], 4099, True, True, False, 2),
    ([300, 40, 1000, 5000, 27, 90], 2048, True, False, False, 8),
])
def test_embed_grad_gemm_vs_unfused(hip, rows, B, with_fm, with_dx, accumulate, owners):
    """rp_embed_grad_gemm (the segmented reduce with the consuming Linear's dgrad formed inside, on the matrix core) against
    the unfused pair rp_linear_fwd (dX = dH . W) + rp_embed_grad_reduce in the fp32-faithful mode, and against an fp64
    reference: the same gradient arena within fp32 rounding; bit-identical between two launches."""
    hip.set_matmul_precision("bf16x6")
    try:
        D, H = 64, 64
        g = torch.Generator().manual_seed(B + len(rows))
        F = len(rows)
        arena, base = _tables(rows, D, g)
        idx = [torch.randint(0, r, (B,), generator=g) for r in rows]
        if owners > 1:  # renumber the arena rows owner-major (what the slots of a row-sharded lookup look like)
            R = sum(rows)
            per = (R + owners - 1) // owners
            renum = (torch.arange(R) % owners) * per + torch.arange(R) // owners
            big = torch.zeros(owners * per, D)
            big[renum] = arena
            arena = big
        K = F * D + 5
        ldx = (K + 63) // 64 * 64
        W1 = torch.randn(H, K, generator=g) / K ** 0.5
        dh = torch.randn(B, H, generator=g) * 1e-3
        dx_extra = torch.randn(B, ldx, generator=g) * 1e-3 if with_dx else None
        gfm = torch.randn(B, 1, generator=g) * 1e-3 if with_fm else None
        ssum = torch.randn(B, D, generator=g) if with_fm else None
        row_of = [(base[f] + idx[f]) if owners == 1 else renum[base[f] + idx[f]] for f in range(F)]
        keys = torch.cat(row_of).to(torch.int32).to(DEV)
        sk, sp = hip.sort_pairs(keys, end_bit=max(1, (arena.shape[0] - 1).bit_length()))
        dev = lambda t_: None if t_ is None else t_.to(DEV)
        wt = hip.transpose(dev(W1), rows_out=ldx)                      # [ldx, 64]
        # unfused: dX = dH . W1 (+ extra), then the reduce
        dxf = hip.linear_fwd(dev(dh), wt, None, hip.ACT_NONE)          # [B, ldx]
        if with_dx:
            dxf = dxf + dev(dx_extra)
        NR = arena.shape[0]
        G0 = torch.full((NR, D), 0.5, device=DEV) if accumulate else torch.zeros(NR, D, device=DEV)
        G1 = G0.clone()
        G2 = G0.clone()
        hip.embed_grad_reduce(sk, sp, B, D, dxf, dev(gfm), dev(ssum), dev(arena), G0, accumulate=accumulate)
        assert hip.embed_grad_gemm_fits(D, H, dev(dh), wt)
        hip.embed_grad_gemm(sk, sp, B, D, dev(dh), wt, dev(dx_extra), dev(gfm), dev(ssum), dev(arena), G1, accumulate=accumulate)
        hip.embed_grad_gemm(sk, sp, B, D, dev(dh), wt, dev(dx_extra), dev(gfm), dev(ssum), dev(arena), G2, accumulate=accumulate)
        assert torch.equal(G1, G2), "two launches differ"
        # fp64 reference
        dX = dh.double() @ W1.double()[:, :F * D]
        if with_dx:
            dX = dX + dx_extra.double()[:, :F * D]
        ref = torch.full((NR, D), 0.5 if accumulate else 0.0, dtype=torch.float64)
        for f in range(F):
            rows_f = row_of[f].long()
            contrib = dX[:, f * D:(f + 1) * D]
            if with_fm:
                contrib = contrib + gfm.double() * (ssum.double() - arena.double()[rows_f])
            ref.index_add_(0, rows_f, contrib)
        scale = float((ref - (0.5 if accumulate else 0.0)).abs().max())
        assert float((G1.cpu().double() - ref).abs().max()) <= 2e-5 * max(scale, 1e-6)
        assert float((G1 - G0).abs().max()) <= 2e-5 * max(scale, 1e-6)
    finally:
        hip.set_matmul_precision("auto")


@pytest.mark.parametrize("M,L", [(65536, 2), (1000, 1), (4099, 3), (128, 2)])
def test_mlp_tail_fused_vs_fp64(hip, M, L):
    """rp_mlp_tail_fwd / rp_mlp_tail_bwd (the 64 -> 64 -> .. -> 1 tail of deep.py:62-72 as one launch each way) against an
    fp64 autograd reference: logits, every saved hidden activation, the gradient w.r.t. the tail's input (masked by its
    ReLU), all weight / bias gradients; two backward launches are bit-identical (fixed-order partial sums)."""
    g = torch.Generator().manual_seed(M + L)
    pre = torch.randn(M, 64, generator=g)
    hin = pre.relu()                                   # the tail's input is a ReLU output
    Ws = [torch.randn(64, 64, generator=g) / 8 for _ in range(L)]
    bs = [torch.randn(64, generator=g) * 0.1 for _ in range(L)]
    w_out = torch.randn(1, 64, generator=g) / 8
    b_out = torch.randn(1, generator=g)
    dz = torch.randn(M, generator=g) / M
    # fp64 reference
    p64 = pre.double().requires_grad_(True)
    W64 = [w.double().requires_grad_(True) for w in Ws]
    b64 = [b.double().requires_grad_(True) for b in bs]
    wo64, bo64 = w_out.double().requires_grad_(True), b_out.double().requires_grad_(True)
    a = p64.relu()
    acts = []
    for w, b in zip(W64, b64):
        a = (a @ w.t() + b).relu()
        acts.append(a)
    logit = a @ wo64.t() + bo64
    (logit.reshape(-1) * dz.double()).sum().backward()
    dev = lambda t_: t_.to(DEV)
    out, hs = hip.mlp_tail_fwd(dev(hin), [dev(w) for w in Ws], [dev(b) for b in bs], dev(w_out), dev(b_out))
    torch.testing.assert_close(out.cpu().double(), logit.detach(), rtol=1e-5, atol=2e-5)
    for got, ref in zip(hs, acts):
        torch.testing.assert_close(got.cpu().double(), ref.detach(), rtol=1e-5, atol=2e-5)
    res = [hip.mlp_tail_bwd(dev(dz), [dev(w) for w in Ws], [dev(hin)] + hs, dev(w_out)) for _ in range(2)]
    dhin, dWs, dbs, dwo, dbo = res[0]
    for x, y in zip((res[0][0], *res[0][1], *res[0][2], res[0][3], res[0][4]), (res[1][0], *res[1][1], *res[1][2], res[1][3], res[1][4])):
        assert torch.equal(x, y), "two backward launches differ"

    def close(got, ref, what):
        scale = float(ref.abs().max())
        err = float((got.cpu().double() - ref).abs().max())
        assert err <= 2e-5 * max(scale, 1e-12), f"{what}: {err} vs scale {scale}"

    close(dhin, p64.grad, "d hin (masked)")
    for l in range(L):
        close(dWs[l], W64[l].grad, f"dW{l}")
        close(dbs[l], b64[l].grad, f"db{l}")
    close(dwo, wo64.grad, "dw_out")
    close(dbo, bo64.grad, "db_out")


@pytest.mark.parametrize("M,L,n_add", [(65536, 2, 1), (1000, 1, 0), (4099, 3, 2), (128, 2, 3)])
def test_mlp_tail_with_the_loss_head_inside_vs_separate_launches(hip, M, L, n_add):
    """rp_mlp_tail_fwd_bce / rp_mlp_tail_bwd_bce (deepfm.py:61-66 folded into the MLP tail's two launches) against
    mlp_tail64 + sigmoid_bce on the same inputs: the predictions and EVERY gradient bit-identical (the same arithmetic per
    row), the loss within fp32 rounding of its sum (another order) and of an fp64 evaluation of the reference's formula."""
    from rec_pangu_amd import functional as Fh
    g = torch.Generator().manual_seed(7 * M + L)
    dev = lambda t_: t_.to(DEV)
    hin0 = dev(torch.randn(M, 64, generator=g).relu())
    Ws0 = [dev(torch.randn(64, 64, generator=g) / 8) for _ in range(L)]
    bs0 = [dev(torch.randn(64, generator=g) * 0.1) for _ in range(L)]
    wo0, bo0 = dev(torch.randn(1, 64, generator=g) / 8), dev(torch.randn(1, generator=g))
    adds0 = [dev(torch.randn(M, 1, generator=g)) for _ in range(n_add)]
    label = dev((torch.rand(M, generator=g) < 0.3).float())
    seed = dev(torch.tensor(0.75))  # (a loss gradient that is not 1: it is read on the device)
    res = {}
    for mode in ("separate", "fused"):
        hin = hin0.clone().requires_grad_(True)
        Ws = [w.clone().requires_grad_(True) for w in Ws0]
        bs = [b.clone().requires_grad_(True) for b in bs0]
        wo, bo = wo0.clone().requires_grad_(True), bo0.clone().requires_grad_(True)
        adds = [a.clone().requires_grad_(True) for a in adds0]
        link = Fh.ReluLink()
        if mode == "separate":
            logit = Fh.mlp_tail64(hin, link, list(zip(Ws, bs)), (wo, bo))
            pred, loss = Fh.sigmoid_bce(adds + [logit], label)
        else:
            pred, loss = Fh.mlp_tail64_bce(hin, link, list(zip(Ws, bs)), (wo, bo), adds, label)
        loss.backward(gradient=seed)
        res[mode] = (pred.detach(), loss.detach(), [t.grad for t in (hin, *Ws, *bs, wo, bo, *adds)], link.dx)
    a, b = res["separate"], res["fused"]
    assert torch.equal(a[0], b[0]), "pred differs"
    for i, (x, y) in enumerate(zip(a[2], b[2])):
        assert x is not None and y is not None and torch.equal(x, y), f"gradient {i} differs"
    assert torch.equal(a[3], b[3]), "the masked input gradient handed to the producing layer differs"
    p64 = a[0].double().reshape(-1).cpu()
    y64 = label.double().cpu()
    ref = -(y64 * p64.log().clamp_min(-100) + (1 - y64) * (1 - p64).log().clamp_min(-100)).mean()
    assert abs(float(b[1]) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert abs(float(a[1]) - float(b[1])) <= 2e-6 * max(1.0, abs(float(ref)))


@pytest.mark.parametrize("rows,ND,B,biased", [
    ([8, 4, 51, 12, 3], 5, 24, True),          # one partial workgroup
    ([3, 4, 10, 5000, 27], 13, 4099, True),    # 32 full workgroups + a 3-sample tail, the Criteo dense count
    ([50, 7], 0, 1024, False),                 # no dense columns, no bias, only full workgroups
    ([9] * 26, 13, 640, True),                 # Criteo field count
    ([5] * 32, 16, 300, True),                 # the limits: 32 fields, 16 dense columns
])
def test_embed_gather_linear_vs_unfused(hip, rows, ND, B, biased):
    """rp_embed_gather_linear_fwd (lookup + concat + FM + the 64-wide Linear + ReLU in one launch) against the unfused
    rp_embed_gather_fwd + rp_linear_fwd in the fp32-faithful mode: x / keys bit-identical to the plain gather, fm / ssum
    within fp32 rounding, h1 within fp32 rounding of an fp64 reference; an out-of-range index raises the same error flag."""
    hip.set_matmul_precision("bf16x6")
    try:
        D, H = 64, 64
        g = torch.Generator().manual_seed(B + len(rows))
        F = len(rows)
        arena, base = _tables(rows, D, g)
        idx = [torch.randint(0, r, (B,), generator=g) for r in rows]
        dense = [torch.rand(B, generator=g) for _ in range(ND)]
        K = F * D + ND
        ldx = (K + 63) // 64 * 64
        W = torch.randn(H, K, generator=g) / K ** 0.5
        bias = torch.randn(H, generator=g) * 0.1 if biased else None
        dev = lambda t_: None if t_ is None else t_.to(DEV)
        rb = torch.tensor(base, dtype=torch.int64, device=DEV)
        rc = torch.tensor(rows, dtype=torch.int64, device=DEV)
        err = torch.zeros(2, dtype=torch.int32, device=DEV)
        didx, dd = [dev(t_) for t_ in idx], [dev(t_) for t_ in dense]
        x0, fm0, s0, k0 = hip.embed_gather_fwd(dev(arena), rb, rc, didx, dd, ldx, True, True, True, err)
        Wd = dev(W)
        if Wd.stride(0) % 4:
            Wp = torch.zeros(H, (K + 3) // 4 * 4, device=DEV)
            Wp[:, :K] = Wd
            Wd = Wp[:, :K]
        assert hip.embed_gather_linear_fits(D, F, ND, H, ldx, Wd)
        x1, h1, fm1, s1, k1 = hip.embed_gather_linear_fwd(dev(arena), rb, rc, didx, dd, ldx, Wd, dev(bias), True, True, True, err)
        x2, h2, _, _, _ = hip.embed_gather_linear_fwd(dev(arena), rb, rc, didx, dd, ldx, Wd, dev(bias), True, True, True, err)
        assert int(err[0]) == 0
        assert torch.equal(x1[:, :K], x0[:, :K]) and torch.equal(k1, k0), "gathered rows / keys must be bit-exact"
        torch.testing.assert_close(s1, s0, rtol=1e-5, atol=1e-5)
        fscale = (x0[:, :F * D].abs().sum(dim=1, keepdim=True) ** 2).clamp(min=1.0) * 1e-6
        assert bool(((fm1 - fm0).abs() <= fscale + 1e-5).all())
        assert torch.equal(h1, h2), "two launches differ"
        xr = torch.cat([arena[base[f] + idx[f]] for f in range(F)] + [d_[:, None] for d_ in dense], dim=1).double()
        pre = xr @ W.double().T + (bias.double() if biased else 0.0)
        ref = pre.clamp_min(0)
        scale = float(pre.abs().max())
        # a pre-activation within rounding of zero may land on either side: compare where |pre| is clear of it
        clear = pre.abs() > 1e-5 * scale
        assert float(((h1.cpu().double() - ref) * clear).abs().max()) <= 2e-5 * scale
        assert float((h1.cpu().double() - ref).abs().max()) <= 4e-5 * scale
        # same composed on the device
        h0 = hip.linear_fwd(x0, Wd, dev(bias), hip.ACT_RELU, K=K)
        assert float((h1 - h0).abs().max()) <= 4e-5 * scale
        # an index past its table: flagged, the sample's row reads as zeros like the plain gather
        bad = [t_.clone() for t_ in didx]
        bad[1][B // 2] = rows[1] + 3
        hip.embed_gather_linear_fwd(dev(arena), rb, rc, bad, dd, ldx, Wd, dev(bias), True, False, False, err)
        assert int(err[0]) != 0
    finally:
        hip.set_matmul_precision("auto")


@pytest.mark.parametrize("M,F,ND", [(65536, 26, 13), (4096, 6, 0), (1000, 2, 16), (333, 4, 5)])
@pytest.mark.parametrize("mode", ["auto", "bf16x6", "bf16x3"])
def test_linear_wgrad_gather_equals_wgrad_on_the_stored_activation(M, F, ND, mode):
    """rp_linear_wgrad_gather (the first layer's weight gradient gathering the embedding rows itself, so that the forward
    need not store them) against rp_linear_wgrad on the materialised [M, F*64 + ND] activation: the same staging, the same
    split plan, the same summation order — bit-identical, in every bf16 matrix-core mode."""
    from rec_pangu_amd import hip
    prev = hip.get_matmul_precision()
    hip.set_matmul_precision(mode)
    try:
        _wgrad_gather_case(hip, M, F, ND)
    finally:
        hip.set_matmul_precision(prev)


def _wgrad_gather_case(hip, M, F, ND):
    g = torch.Generator().manual_seed(M + F)
    R, D, K, Kg = 50000, 64, F * 64 + ND, F * 64
    arena = torch.randn(R, D, generator=g).to(DEV)
    keys = torch.randint(0, R, (F * M,), generator=g).to(torch.int32).to(DEV)
    xd = torch.zeros(M, 64)
    xd[:, :ND] = torch.rand(M, ND, generator=g)
    xd = xd.to(DEV)
    dy = (torch.randn(M, 64, generator=g) * (torch.rand(M, 64, generator=g) < 0.5)).to(DEV)
    assert hip.linear_wgrad_gather_fits(M, 64, K, Kg) == (F % 2 == 0)
    ld = (K + 63) // 64 * 64
    x = torch.zeros(M, ld, device=DEV)
    x[:, :Kg] = arena[keys.long().view(F, M).t().reshape(-1)].view(M, Kg)
    x[:, Kg:K] = xd[:, :ND]
    dw_ref, db_ref = hip.linear_wgrad(dy, x, K)
    dw, db = hip.linear_wgrad_gather(dy, arena, keys, Kg, xd if ND else None, K)
    assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)


@pytest.mark.parametrize("rows,B,with_fm,accumulate", [
    ([3, 40, 500, 7, 100, 20000, 11], 3000, True, False),    # tiny tables 3, 7, 11, 40, 100 rows; B not a multiple of 64
    ([25, 4, 1000, 28, 16, 106, 5], 8192, True, True),       # accumulate into a non-zero arena
    ([10, 19, 3000, 4], 1100, False, False),                 # no FM term
])
def test_embed_grad_tiny_tables_vs_sorted_path_and_fp64(hip, rows, B, with_fm, accumulate):
    """rp_embed_grad_tiny (sample-major one-hot GEMMs for the tables of a few rows) + rp_embed_grad_gemm(skip_fields) against
    rp_embed_grad_gemm over every field and against an fp64 reference: the same gradient arena within fp32 rounding,
    rows of the tiny tables nobody looked up left untouched (the deferred optimizer keeps waiting gradients there),
    bit-identical between two launches."""
    D, H = 64, 64
    g = torch.Generator().manual_seed(B + len(rows))
    F = len(rows)
    arena, base = _tables(rows, D, g)
    idx = [torch.randint(0, r, (B,), generator=g) for r in rows]
    idx[0][:] = idx[0].clamp(max=rows[0] - 2) if rows[0] > 2 else idx[0]  # (the last row of table 0 is never looked up)
    K = F * D + 5
    ldx = (K + 63) // 64 * 64
    W1 = torch.randn(H, K, generator=g) / K ** 0.5
    dh = torch.randn(B, H, generator=g) * 1e-3
    gfm = torch.randn(B, 1, generator=g) * 1e-3 if with_fm else None
    ssum = torch.randn(B, D, generator=g) if with_fm else None
    row_of = [base[f] + idx[f] for f in range(F)]
    keys = torch.cat(row_of).to(torch.int32).to(DEV)
    sk, sp = hip.sort_pairs(keys, end_bit=max(1, (arena.shape[0] - 1).bit_length()))
    dev = lambda t_: None if t_ is None else t_.to(DEV)  # noqa: E731
    wt = hip.transpose(dev(W1), rows_out=ldx)
    NR = arena.shape[0]
    init = 0.5 if accumulate else 0.0
    tiny = [(f, int(base[f]), rows[f]) for f in range(F) if rows[f] <= 254]
    assert sum(t[2] for t in tiny) <= 224 and len(tiny) >= 3
    skip = sum(1 << t[0] for t in tiny)
    Gs = []
    never = int(base[0]) + rows[0] - 1 if rows[0] > 2 else None
    for _ in range(2):
        G = torch.full((NR, D), init, device=DEV)
        if never is not None:
            G[never] = 7.25  # (what a waiting gradient of an earlier step would look like)
        hip.embed_grad_tiny(keys, B, tiny, dev(dh), wt, dev(gfm), dev(ssum), dev(arena), G, accumulate)
        hip.embed_grad_gemm(sk, sp, B, D, dev(dh), wt, None, dev(gfm), dev(ssum), dev(arena), G, accumulate, skip_fields=skip)
        Gs.append(G)
    assert torch.equal(Gs[0], Gs[1]), "two launches differ"
    Gall = torch.full((NR, D), init, device=DEV)
    hip.embed_grad_gemm(sk, sp, B, D, dev(dh), wt, None, dev(gfm), dev(ssum), dev(arena), Gall, accumulate)
    dX = dh.double() @ W1.double()[:, :F * D]
    ref = torch.full((NR, D), init, dtype=torch.float64)
    for f in range(F):
        contrib = dX[:, f * D:(f + 1) * D]
        if with_fm:
            contrib = contrib + gfm.double() * (ssum.double() - arena.double()[row_of[f].long()])
        ref.index_add_(0, row_of[f].long(), contrib)
    scale = float((ref - init).abs().max())
    got = Gs[0].cpu().double()
    if never is not None:
        assert torch.equal(Gs[0][never].cpu(), torch.full((D,), 7.25)), "a tiny-table row nobody looked up must not be written"
        got[never] = init
        Gs[0][never] = init
    assert float((got - ref).abs().max()) <= 2e-5 * max(scale, 1e-6), "against fp64"
    assert float((Gs[0] - Gall).abs().max()) <= 2e-5 * max(scale, 1e-6), "against the row-sorted kernel over every field"


@pytest.mark.parametrize("rows,B,with_fm,accumulate,with_tiny", [
    ([8, 4, 51, 12, 3], 24, True, False, False),               # one partial tile per field, every run inside one group
    ([3, 4, 10, 5000, 27], 4099, True, False, False),          # runs of > 1000 equal rows: pieces across groups, tiles and chunks
    ([3, 4, 10, 5000, 27], 4099, False, False, True),          # no FM term; the tiny tables through rp_embed_grad_tiny
    ([300, 40, 1000, 5000, 27, 90], 2048, True, True, True),   # accumulate into a non-zero arena + tiny tables
    ([50000, 7, 2000000], 65536, True, False, False),          # full batch: mostly unique rows, a hot table, 512 tiles per field
    ([50000, 13, 5, 700, 90], 16384 + 77, True, False, True),  # a ragged last tile
])
@pytest.mark.parametrize("impl", ["seg", "ss"])
def test_embed_grad_seg_vs_fp64_and_grad_gemm(hip, rows, B, with_fm, accumulate, with_tiny, impl):
    """rp_embed_grad_seg (segment sums first, then ONE matrix pass per run piece that yields the table rows' gradient AND the
    embedding columns of the first layer's weight gradient; the forward stores no activation) against an fp64 restatement
    of the reference's ops (embedding.py:61-63 backward + interaction.py:38-44 backward + deep.py:62-72 dgrad / wgrad) and
    against rp_embed_grad_gemm; bit-identical between two launches; both tile sizes when the library was started with
    RP_SEG_ROWS (the default here)."""
    D, H = 64, 64
    seg_fn = hip.embed_grad_seg if impl == "seg" else hip.embed_grad_ss  # (round 6: the two-launch form, same contract)
    g = torch.Generator().manual_seed(B + len(rows))
    F = len(rows)
    arena, base = _tables(rows, D, g)
    idx = [torch.randint(0, r, (B,), generator=g) for r in rows]
    ND = 5
    K = F * D + ND
    ldx = (K + 63) // 64 * 64
    W1 = torch.randn(H, K, generator=g) / K ** 0.5
    dh = torch.randn(B, H, generator=g) * 1e-3 * (torch.rand(B, H, generator=g) < 0.6)
    gfm = torch.randn(B, 1, generator=g) * 1e-3 if with_fm else None
    ssum = torch.randn(B, D, generator=g) if with_fm else None
    row_of = [base[f] + idx[f] for f in range(F)]
    keys = torch.cat(row_of).to(torch.int32).to(DEV)
    sk, sp = hip.sort_pairs(keys, end_bit=max(1, (arena.shape[0] - 1).bit_length()))
    dev = lambda t_: None if t_ is None else t_.to(DEV)  # noqa: E731
    Wd = dev(W1)
    wt = hip.transpose(Wd, rows_out=ldx)
    NR = arena.shape[0]
    init = 0.5 if accumulate else 0.0
    tiny = [(f, int(base[f]), rows[f]) for f in range(F) if rows[f] <= 254] if with_tiny else []
    skip = sum(1 << t[0] for t in tiny)
    outs = []
    for _ in range(2):
        G = torch.full((NR, D), init, device=DEV)
        dw = torch.full((H, K), float("nan"), device=DEV)
        if tiny:
            hip.embed_grad_tiny(keys, B, tiny, dev(dh), wt, dev(gfm), dev(ssum), dev(arena), G, accumulate, dw=dw)
        seg_fn(sk, sp, B, D, dev(dh), Wd, dev(gfm), dev(ssum), dev(arena), G, accumulate, skip_fields=skip,
                           field_rows=rows, dw=dw)
        outs.append((G, dw))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1][:, :F * D], outs[1][1][:, :F * D]), "two launches differ"
    G, dw = outs[0]
    assert bool(torch.isnan(dw[:, F * D:]).all()), "the dense columns are not this launch's"
    # fp64 restatement
    xr = torch.cat([arena[row_of[f].long()] for f in range(F)], dim=1).double()
    dX = dh.double() @ W1.double()[:, :F * D]
    ref = torch.full((NR, D), init, dtype=torch.float64)
    for f in range(F):
        contrib = dX[:, f * D:(f + 1) * D]
        if with_fm:
            contrib = contrib + gfm.double() * (ssum.double() - arena.double()[row_of[f].long()])
        ref.index_add_(0, row_of[f].long(), contrib)
    dw_ref = dh.double().T @ xr
    scale = float((ref - init).abs().max())
    assert float((G.cpu().double() - ref).abs().max()) <= 2e-5 * max(scale, 1e-6), "table gradient against fp64"
    wscale = float(dw_ref.abs().max())
    assert float((dw[:, :F * D].cpu().double() - dw_ref).abs().max()) <= 2e-5 * max(wscale, 1e-6), "weight gradient against fp64"
    # the pair-form kernel over every field
    Gall = torch.full((NR, D), init, device=DEV)
    hip.embed_grad_gemm(sk, sp, B, D, dev(dh), wt, None, dev(gfm), dev(ssum), dev(arena), Gall, accumulate)
    assert float((G - Gall).abs().max()) <= 2e-5 * max(scale, 1e-6)
    # no weight gradient asked for: the same rows, nothing else written
    G2 = torch.full((NR, D), init, device=DEV)
    if tiny:
        hip.embed_grad_tiny(keys, B, tiny, dev(dh), wt, dev(gfm), dev(ssum), dev(arena), G2, accumulate)
    seg_fn(sk, sp, B, D, dev(dh), Wd, dev(gfm), dev(ssum), dev(arena), G2, accumulate, skip_fields=skip)
    assert torch.equal(G2, G)
    if impl == "ss":
        # the unique-row lists made ahead (rp_embed_grad_ss_mark: with the sort, from the sorted keys alone) — index work, exact
        # against a host restatement — and the launch fed with them: the same bits as with the lists it makes itself
        marks = hip.embed_grad_ss_mark(sk, B, skip)
        kept = [f for f in range(F) if not (skip >> f) & 1]
        nbk = (B + 255) // 256
        skc = sk.cpu().long()
        offs = marks[2].cpu().long()
        run = 0
        for r_, f in enumerate(kept):
            ks = skc[f * B:(f + 1) * B]
            start = torch.ones(B, dtype=torch.bool)
            start[1:] = ks[1:] != ks[:-1]
            xs = torch.nonzero(start).flatten()
            assert torch.equal(marks[0].cpu().long()[r_ * B:r_ * B + xs.numel()], xs), f
            assert torch.equal(marks[1].cpu().long()[r_ * B:r_ * B + xs.numel()], ks[xs]), f
            per_blk = torch.zeros(nbk, dtype=torch.int64).index_add_(0, xs // 256, torch.ones_like(xs))
            assert torch.equal(offs[r_ * nbk:(r_ + 1) * nbk], run + torch.cumsum(per_blk, 0) - per_blk), f
            run += xs.numel()
        assert int(offs[len(kept) * nbk]) == run
        G3 = torch.full((NR, D), init, device=DEV)
        dw3 = torch.full((H, K), float("nan"), device=DEV)
        if tiny:
            hip.embed_grad_tiny(keys, B, tiny, dev(dh), wt, dev(gfm), dev(ssum), dev(arena), G3, accumulate, dw=dw3)
        seg_fn(sk, sp, B, D, dev(dh), Wd, dev(gfm), dev(ssum), dev(arena), G3, accumulate, skip_fields=skip, field_rows=rows, dw=dw3,
               marks=marks)
        assert torch.equal(G3, G) and torch.equal(dw3[:, :F * D], dw[:, :F * D])


@pytest.mark.parametrize("rows,B,with_fm,accumulate,smp,with_tiny", [
    ([8, 400000, 51, 90000], 24, True, False, [1, 3], False),                  # one ragged unit per field, no duplicates at all
    ([3, 2000000, 10, 5000, 70000], 4099, True, False, [1, 4], False),         # ragged last unit, a few duplicate pairs
    ([3, 2000000, 10, 5000, 70000], 4099, False, True, [1, 4], True),          # no FM term, accumulate, tiny tables beside
    ([300, 40, 1000, 5000, 27, 90], 2048, True, True, [0, 2, 3], True),        # EVERY pair of the launch's fields a duplicate
    ([50000, 7, 2000000, 300000], 65536, True, False, [0, 2, 3], False),       # full batch: 1024 units per field, hot + big
    ([50000, 13, 5, 700, 9000000, 90], 16384 + 77, True, False, [0, 4], True),  # keys beyond 2^23 rows, ragged
    ([40, 2000000, 10, 5000], 4099, True, False, [0, 1], False),               # runs of ~100 duplicates: summed by a whole workgroup
    ([1, 300000, 77], 2048 + 5, True, True, [0, 1], False),                    # ONE run over the whole batch + a few pairs
])
def test_embed_grad_smp_vs_fp64_and_seg(hip, rows, B, with_fm, accumulate, smp, with_tiny):
    """rp_embed_grad_smp_mark + rp_embed_grad_smp (round 6: the big tables' share of the first layer's backward, SAMPLE-major:
    a pair that is alone in its run writes its table row's gradient directly, the pairs of longer runs go through a side
    buffer and rp_embed_grad_reduce_rows) composed with rp_embed_grad_seg (the other fields) and rp_embed_grad_tiny, against an
    fp64 restatement of the reference's ops (embedding.py:61-63 backward + interaction.py:38-44 backward + deep.py:62-72
    dgrad / wgrad) and against rp_embed_grad_seg over every field; the marks against a host restatement (index work:
    exact); bit-identical between two launches."""
    D, H = 64, 64
    g = torch.Generator().manual_seed(B + len(rows) + 6)
    F = len(rows)
    arena, base = _tables(rows, D, g)
    idx = [torch.randint(0, r, (B,), generator=g) for r in rows]
    ND = 5
    K = F * D + ND
    ldx = (K + 63) // 64 * 64
    W1 = torch.randn(H, K, generator=g) / K ** 0.5
    dh = torch.randn(B, H, generator=g) * 1e-3 * (torch.rand(B, H, generator=g) < 0.6)
    gfm = torch.randn(B, 1, generator=g) * 1e-3 if with_fm else None
    ssum = torch.randn(B, D, generator=g) if with_fm else None
    row_of = [base[f] + idx[f] for f in range(F)]
    keys = torch.cat(row_of).to(torch.int32).to(DEV)
    sk, sp = hip.sort_pairs(keys, end_bit=max(1, (arena.shape[0] - 1).bit_length()))
    dev = lambda t_: None if t_ is None else t_.to(DEV)  # noqa: E731
    Wd = dev(W1)
    wt = hip.transpose(Wd, rows_out=ldx)
    NR = arena.shape[0]
    init = 0.5 if accumulate else 0.0
    tiny = [(f, int(base[f]), rows[f]) for f in range(F) if rows[f] <= 254 and f not in smp] if with_tiny else []
    skip = sum(1 << t[0] for t in tiny) | sum(1 << f for f in smp)
    # ---- the marks: index work, exact against a host restatement
    smp_t = [(f, int(base[f]), rows[f]) for f in smp]
    marks = hip.embed_grad_smp_mark(sk, sp, B, smp)
    skc, spc = sk.cpu().long(), sp.cpu().long()
    dupq_ref = torch.full((len(smp) * B,), -1, dtype=torch.int64)
    dupk_ref = torch.full((len(smp) * B,), -1, dtype=torch.int64)
    c0 = 0  # the duplicates are numbered in sorted order, field by field; the key list is compact, -1 behind them
    for fi, f in enumerate(smp):
        ks, ps = skc[f * B:(f + 1) * B], spc[f * B:(f + 1) * B]
        assert bool(((ps >= f * B) & (ps < (f + 1) * B)).all())
        dup = torch.zeros(B, dtype=torch.bool)
        dup[1:] |= ks[1:] == ks[:-1]
        dup[:-1] |= ks[:-1] == ks[1:]
        c = c0 + torch.cumsum(dup.long(), 0) - 1
        dupq_ref[fi * B + (ps - f * B)] = torch.where(dup, c, torch.full_like(c, -1))
        dupk_ref[c[dup]] = ks[dup]
        c0 += int(dup.sum())
    assert torch.equal(marks[0].cpu().long(), dupq_ref) and torch.equal(marks[1].cpu().long(), dupk_ref)
    outs = []
    for _ in range(2):
        G = torch.full((NR, D), init, device=DEV)
        dw = torch.full((H, K), float("nan"), device=DEV)
        if tiny:
            hip.embed_grad_tiny(keys, B, tiny, dev(dh), wt, dev(gfm), dev(ssum), dev(arena), G, accumulate, dw=dw)
        hip.embed_grad_smp(keys, marks, B, F, smp_t, dev(dh), Wd, dev(gfm), dev(ssum), dev(arena), G, accumulate, dw=dw)
        if skip != (1 << F) - 1:
            hip.embed_grad_seg(sk, sp, B, D, dev(dh), Wd, dev(gfm), dev(ssum), dev(arena), G, accumulate, skip_fields=skip,
                               field_rows=rows, dw=dw)
        outs.append((G, dw))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1][:, :F * D], outs[1][1][:, :F * D]), "two launches differ"
    G, dw = outs[0]
    xr = torch.cat([arena[row_of[f].long()] for f in range(F)], dim=1).double()
    dX = dh.double() @ W1.double()[:, :F * D]
    ref = torch.full((NR, D), init, dtype=torch.float64)
    for f in range(F):
        contrib = dX[:, f * D:(f + 1) * D]
        if with_fm:
            contrib = contrib + gfm.double() * (ssum.double() - arena.double()[row_of[f].long()])
        ref.index_add_(0, row_of[f].long(), contrib)
    dw_ref = dh.double().T @ xr
    scale = float((ref - init).abs().max())
    assert float((G.cpu().double() - ref).abs().max()) <= 2e-5 * max(scale, 1e-6), "table gradient against fp64"
    wscale = float(dw_ref.abs().max())
    assert float((dw[:, :F * D].cpu().double() - dw_ref).abs().max()) <= 2e-5 * max(wscale, 1e-6), "weight gradient against fp64"
    # rows nobody looked up are untouched
    touched = torch.zeros(NR, dtype=torch.bool)
    touched[torch.cat(row_of).long()] = True
    assert bool((G.cpu()[~touched] == init).all())
    # rp_embed_grad_seg over every field
    Gall = torch.full((NR, D), init, device=DEV)
    hip.embed_grad_seg(sk, sp, B, D, dev(dh), Wd, dev(gfm), dev(ssum), dev(arena), Gall, accumulate, skip_fields=0, field_rows=rows)
    assert float((G - Gall).abs().max()) <= 2e-5 * max(scale, 1e-6)
    # no weight gradient asked for: the same rows
    G2 = torch.full((NR, D), init, device=DEV)
    if tiny:
        hip.embed_grad_tiny(keys, B, tiny, dev(dh), wt, dev(gfm), dev(ssum), dev(arena), G2, accumulate)
    hip.embed_grad_smp(keys, marks, B, F, smp_t, dev(dh), Wd, dev(gfm), dev(ssum), dev(arena), G2, accumulate)
    if skip != (1 << F) - 1:
        hip.embed_grad_seg(sk, sp, B, D, dev(dh), Wd, dev(gfm), dev(ssum), dev(arena), G2, accumulate, skip_fields=skip)
    assert torch.equal(G2, G)


def test_embed_grad_reduce_rows_vs_host(hip):
    """rp_embed_grad_reduce_rows: sums of the rows of a key-sorted list per key, key -1 = no entry (never read: NaN rows
    there), runs across workgroups and a run of thousands of entries; against an fp64 index_add; accumulate on top."""
    g = torch.Generator().manual_seed(61)
    n, D, NR = 70000, 64, 5000
    keys = torch.sort(torch.randint(0, NR, (n,), generator=g)).values
    keys[20000:29000] = keys[20000]                     # one run of 9000 entries
    keys = torch.sort(keys).values
    # (holes cover WHOLE runs: the entries of one key stay contiguous — the list rp_embed_grad_smp_mark writes)
    hole = (torch.rand(NR, generator=g) < 0.4)[keys]
    rows = torch.randn(n, D, generator=g)
    rows[hole] = float("nan")
    k32 = torch.where(hole, torch.full_like(keys, -1), keys).to(torch.int32)
    for accumulate in (False, True):
        init = 0.25 if accumulate else 0.0
        G = torch.full((NR, D), init, device=DEV)
        hip.embed_grad_reduce_rows(k32.to(DEV), rows.to(DEV), G, accumulate)
        ref = torch.full((NR, D), init, dtype=torch.float64)
        ref.index_add_(0, keys[~hole], rows[~hole].double())
        live = torch.zeros(NR, dtype=torch.bool)
        live[keys[~hole]] = True
        got = G.cpu().double()
        assert float((got[live] - ref[live]).abs().max()) <= 1e-4
        assert bool((got[~live] == init).all())


@pytest.mark.parametrize("act_name", ["Tanh", "Sigmoid", "LeakyReLU"])
@pytest.mark.parametrize("M,N,K", [(4096, 64, 64), (3000, 200, 333), (8192, 256, 1677), (777, 1, 39)])
def test_linear_activation_epilogues_vs_torch(hip, act_name, M, N, K):
    """rp_linear_fwd with the Tanh / Sigmoid / LeakyReLU(0.01) epilogues (activation.py:37-59 hands these modules to an MLP by
    name) and their backward through the activation's OUTPUT (rp_act_bwd) against torch fp32 / fp64: every GEMM kernel
    family (short K: one elementwise launch behind it; the tiled and wide kernels: fused), forward within fp32 rounding of
    an fp64 reference, backward within 2e-6 of the analytic derivative; the standalone launches rp_act_fwd / rp_act_bwd too."""
    hip.set_matmul_precision("bf16x6")
    try:
        g = torch.Generator().manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g) * 0.1
        mod = getattr(torch.nn, act_name)()
        code = {"Tanh": hip.ACT_TANH, "Sigmoid": hip.ACT_SIGMOID, "LeakyReLU": hip.ACT_LEAKY}[act_name]
        xd = x.to(DEV)
        if xd.stride(0) % 4:
            xp = torch.zeros(M, (K + 3) // 4 * 4, device=DEV)
            xp[:, :K] = xd
            xd = xp[:, :K]
        Wd = torch.zeros(N, (K + 3) // 4 * 4, device=DEV)
        Wd[:, :K] = W.to(DEV)
        y = hip.linear_fwd(xd, Wd[:, :K], b.to(DEV), code, K=K)
        pre = x.double() @ W.double().T + b.double()
        ref = mod(pre)
        scale = max(1.0, float(pre.abs().max()))
        assert float((y.cpu().double() - ref).abs().max()) <= 2e-5 * scale
        dy = torch.randn(M, N, generator=g)
        p = pre.clone().requires_grad_(True)
        mod(p).backward(dy.double())
        dpre = hip.act_bwd(dy.to(DEV), y, code)
        # (the derivative is formed from the fp32 OUTPUT: a pre-activation within rounding of LeakyReLU's kink may land on
        #  either side — compare where it is clear of it)
        clear = pre.abs() > 1e-4 * scale
        assert float(((dpre.cpu().double() - p.grad) * clear).abs().max()) <= 5e-5 * max(1.0, float(dy.abs().max()))
        y2 = hip.act_fwd(hip.linear_fwd(xd, Wd[:, :K], b.to(DEV), hip.ACT_NONE, K=K), code)
        assert float((y2 - y).abs().max()) <= 1e-6 * scale
    finally:
        hip.set_matmul_precision("auto")
