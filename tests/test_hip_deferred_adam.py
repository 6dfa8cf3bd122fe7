"""Deferred execution of the lazy Adam's real step (FusedAdam(defer=True), csrc/adam.hip "DEFERRED execution").

The real step of a row waits in the gradient arena until the row is next needed; one launch per training step then
applies it together with the row's skipped zero-gradient steps.  Per row the same operations run on the same values in
the same order as in the immediate execution, so EVERYTHING observable must be bit-identical: the predictions of every
step (the forward reads caught-up rows), and parameters + optimizer state after a flush — with the serial replay
(which is itself bit-identical to the dense kernel) and with the closed-form replay."""
import copy

import pytest
import torch

from conftest import require_gpu

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    require_gpu()
    from rec_pangu_amd import hip
    hip.lib()


def _enc(n_dense, vocabs):
    enc = {f"I{i}": {"min": 0.0, "max": 1.0} for i in range(n_dense)}
    enc.update({f"C{i}": {"vocab_size": v} for i, v in enumerate(vocabs)})
    return enc


def _batches(enc, B, n, seed):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        b = {k: (torch.rand(B, generator=gen) if "min" in v else torch.randint(0, v["vocab_size"] + 1, (B,), generator=gen))
             for k, v in enc.items()}
        b["label"] = (torch.rand(B, generator=gen) < 0.3).float()
        out.append({k: v.to(DEV) for k, v in b.items()})
    return out


def _model(kind, enc):
    from rec_pangu_amd.models.ranking import DeepFM, xDeepFM
    torch.manual_seed(0)
    if kind == "deepfm64":      # the fused gather + Linear forward / fused gather backward of the headline model
        m = DeepFM(embedding_dim=64, hidden_units=[64, 64, 64], enc_dict=enc)
    elif kind == "deepfm8":
        m = DeepFM(embedding_dim=8, hidden_units=[16], enc_dict=enc)
    else:                       # a second EmbeddingLayer (the LR tables, D = 1: scalar lanes) with its own lazy state
        m = xDeepFM(embedding_dim=16, dnn_hidden_units=[16], cin_layer_units=[8, 8], enc_dict=enc)
    m = m.to(DEV)
    for mod in m.modules():
        if hasattr(mod, "check_indices"):
            mod.check_indices = "deferred"
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def _state(model, opt):
    sd = {k: v.clone() for k, v in model.state_dict().items()}       # (state_dict() flushes the lazy rows)
    od = opt.state_dict()["state"]
    for i, st in od.items():
        for k, v in st.items():
            if torch.is_tensor(v):
                sd[f"opt.{i}.{k}"] = v.clone()
    return sd


@pytest.mark.parametrize("kind,replay,steps", [("deepfm64", "closed", 330), ("deepfm64", "exact", 70), ("deepfm8", "closed", 300),
                                               ("xdeepfm", "exact", 40)])
def test_deferred_step_is_bit_identical_to_the_immediate_one(kind, replay, steps):
    from rec_pangu_amd.optim import make_adam
    enc = _enc(2, [50, 7, 3000, 20000])
    batches = _batches(enc, 256, 16, seed=11)
    runs = []
    for defer in (False, True):
        model = _model(kind, enc)
        opt = make_adam(model, 2e-3, replay=replay, defer=defer)
        assert opt.defer == defer
        preds, snaps = [], {}
        for i in range(steps):
            b = batches[(i * 7) % 16]
            model.prefetch(batches[((i + 1) * 7) % 16])
            out = model(b)
            out["loss"].backward()
            if i % 23 == 5:      # gradient accumulation: a second forward + backward before the step (rows of both)
                model(batches[(i + 3) % 16])["loss"].backward()
            if i % 29 == 9:      # an evaluation pass between backward and step: must not disturb the waiting gradients
                model.eval()
                with torch.no_grad():
                    preds.append(model(batches[(i + 5) % 16], is_training=False)["pred"].clone())
                model.train()
            if i == 31:          # a flush (state_dict) while the gradients of the step in progress are waiting
                snaps["mid"] = _state(model, opt)
            opt.step()
            model.zero_grad()
            preds.append(out["pred"].detach().clone())
            if i % 10 == 0:
                for g in opt.param_groups:   # a changing learning rate: the waiting step must use ITS step's scalars
                    g["lr"] = 2e-3 * (1.0 + 0.1 * ((i // 10) % 3))
        if defer:
            lz = model.embedding_layer._lazy
            assert lz.defer and int((lz.last < 0).sum()) > 0, "no real step is waiting: the deferred path did not run"
        snaps["end"] = _state(model, opt)
        if defer:
            lz = model.embedding_layer._lazy
            assert int((lz.last < 0).sum()) == 0 and int((model.embedding_layer.grad_arena != 0).sum()) == 0, \
                "a flush applies every waiting gradient and clears its row"
        runs.append((preds, snaps))
    (pa, sa), (pb, sb) = runs
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.equal(a, b), f"prediction {i} differs"
    for tag in ("mid", "end"):
        if tag in sa:
            for k in sa[tag]:
                assert torch.equal(sa[tag][k], sb[tag][k]), (tag, k)


def test_deferred_step_saves_the_second_table_launch_and_survives_a_resume():
    """(a) per training step the deferred mode issues ONE optimizer launch on the tables (the catch-up) where the
    immediate mode issues two (replay + step); (b) optimizer state saved from a deferred run resumes in either mode."""
    from rec_pangu_amd import hip
    from rec_pangu_amd.optim import make_adam
    enc = _enc(1, [500, 9, 4000])
    batches = _batches(enc, 128, 12, seed=3)
    counts = {}
    for defer in (False, True):
        model = _model("deepfm8", enc)
        opt = make_adam(model, 1e-3, replay="exact", defer=defer)
        for i in range(6):
            if i == 4:
                n0 = hip.launch_count()
            model(batches[i])["loss"].backward()
            opt.step()
            model.zero_grad()
        torch.cuda.synchronize()
        counts[defer] = hip.launch_count() - n0
        if defer:
            saved = (copy.deepcopy(model.state_dict()), copy.deepcopy(opt.state_dict()))
    assert counts[True] == counts[False] - 2, counts   # two steps measured: one launch less in each
    finals = []
    for defer in (False, True):
        model = _model("deepfm8", enc)
        opt = make_adam(model, 1e-3, replay="exact", defer=defer)
        model.load_state_dict(saved[0])
        opt.load_state_dict(copy.deepcopy(saved[1]))  # (load_state_dict adopts the given tensors: each run gets its own)
        for i in range(6, 12):
            model(batches[i])["loss"].backward()
            opt.step()
            model.zero_grad()
        finals.append(_state(model, opt))
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


def test_deferred_step_in_a_captured_graph():
    """the catch-up launch reads the step number from the device counter: a captured step replays it bit-identically"""
    from rec_pangu_amd.graph_step import GraphedTrainStep
    from rec_pangu_amd.models.layers.embedding import EmbeddingLayer
    from rec_pangu_amd.optim import make_adam
    enc = _enc(2, [500, 9, 4000])
    batches = _batches(enc, 256, 8, seed=9)
    finals = {}
    try:
        for mode in ("eager", "graph"):
            model = _model("deepfm64", enc)
            opt = make_adam(model, 2e-3, replay="closed", defer=True)
            gstep = GraphedTrainStep(model, opt) if mode == "graph" else None
            for i in range(300):
                cur, nb = batches[i % 8], batches[(i + 1) % 8]
                if gstep is not None:
                    gstep(cur, nb)
                else:
                    model.prefetch(nb)
                    model(cur)["loss"].backward()
                    opt.step()
                    model.zero_grad()
            finals[mode] = _state(model, opt)
            if gstep is not None:
                assert gstep.replays >= 290
                del gstep
    finally:
        EmbeddingLayer.unpin_sorts()
    for k in finals["eager"]:
        assert torch.equal(finals["eager"][k], finals["graph"][k]), k


def test_uncleared_catchup_rows_when_the_promised_backward_does_not_come():
    """Round 4: the catch-up launch in front of a training forward leaves the gradient rows it applies UNCLEARED when the
    backward that follows overwrites them anyway (rp_lazy_adam_catchup mark = 2, LazyAdamRows.replay) — 1 of its 8 row
    transfers.  The promise can be broken; then the rows are cleared by whoever notices first and the results stay
    bit-identical to the immediate execution:
      i % 5 == 1  a training forward whose backward never runs, followed by the optimizer step (a zero-gradient step of
                  every row; the stamped rows must read as zeros)
      i % 5 == 2  two training forwards before any backward, then ONE backward of the summed loss (the second lookup
                  arrives while the first one's rows are still uncleared; autograd runs the second lookup's backward first)
      i % 5 == 3  a training forward, an evaluation forward, then the backward
      otherwise   the plain step (the promise is kept: no clearing launch at all — asserted on the launch counter)"""
    from rec_pangu_amd import hip
    from rec_pangu_amd.optim import make_adam
    enc = _enc(2, [50, 7, 3000, 20000])
    batches = _batches(enc, 256, 16, seed=5)
    runs = []
    for defer in (False, True):
        model = _model("deepfm64", enc)
        opt = make_adam(model, 2e-3, replay="exact", defer=defer)
        preds, plain_launches = [], []
        for i in range(60):
            b, b2 = batches[(i * 7) % 16], batches[(i * 7 + 3) % 16]
            n0 = hip.launch_count()
            if i % 5 == 1:
                out = model(b)
            elif i % 5 == 2:
                out, out2 = model(b), model(b2)
                (out["loss"] + out2["loss"]).backward()
                preds.append(out2["pred"].detach().clone())
            elif i % 5 == 3:
                out = model(b)
                model.eval()
                with torch.no_grad():
                    preds.append(model(b2, is_training=False)["pred"].clone())
                model.train()
                out["loss"].backward()
            else:
                out = model(b)
                out["loss"].backward()
            opt.step()
            model.zero_grad()
            if i % 5 in (0, 4) and i >= 10:
                plain_launches.append(hip.launch_count() - n0)
            preds.append(out["pred"].detach().clone())
        if defer:
            lz = model.embedding_layer._lazy
            assert lz._noclear is None
            assert len(set(plain_launches)) == 1, plain_launches  # (no clearing launch sneaks into a plain step)
        runs.append((preds, _state(model, opt)))
        if defer:
            assert int((model.embedding_layer.grad_arena != 0).sum()) == 0, "after a flush no gradient row holds anything"
    (pa, sa), (pb, sb) = runs
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.equal(a, b), f"prediction {i} differs"
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_reading_table_gradients_under_the_deferred_step_raises():
    """under FusedAdam(defer=True) a table's gradient rows hold this step's gradients AND earlier steps'
    waiting ones.  Computing with a table's .grad between backward() and step() (gradient clipping) therefore RAISES instead
    of silently reading them; addresses and shapes stay readable, a knowing read goes through deferred_grad_reads(), and the
    immediate execution (defer=False) hands out plain dense gradients as the reference's loop would see them."""
    from rec_pangu_amd.optim import FusedAdam
    from rec_pangu_amd.models.layers.embedding import DeferredGradView, deferred_grad_reads
    enc = _enc(2, [300, 7, 2000])
    batches = _batches(enc, 128, 4, seed=5)
    for defer in (True, False):
        m = _model("deepfm8", enc)
        opt = FusedAdam(m.parameters(), lr=1e-2, fuse_zero_grad=True, lazy_tables=True, replay="closed", defer=defer)
        for b in batches[:3]:
            m(b)["loss"].backward()
            opt.step()
            m.zero_grad()
        m(batches[3])["loss"].backward()
        tables = [p for n, p in m.named_parameters() if "embedding_layer" in n]
        assert tables and all(p.grad is not None for p in tables)
        if defer:
            assert all(isinstance(p.grad, DeferredGradView) for p in tables)
            assert all(p.grad.shape == p.shape and p.grad.data_ptr() != 0 for p in tables)   # metadata stays readable
            with pytest.raises(RuntimeError, match="defer"):
                torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            with pytest.raises(RuntimeError, match="defer"):
                tables[0].grad.mul_(0.5)
            with deferred_grad_reads():
                assert float(sum(p.grad.abs().sum() for p in tables)) > 0.0
        else:
            assert not any(isinstance(p.grad, DeferredGradView) for p in tables)
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()   # the step itself is unaffected by the guard
        m.zero_grad()
