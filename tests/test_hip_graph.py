"""The captured training step (rec_pangu_amd/graph_step.py: forward + backward + FusedAdam step + zero_grad captured once,
device-resident step counters, static double-buffered inputs, next batch's sort inside the capture) against the eager loop
on the same batches: every prediction, loss, weight and optimizer moment bit-identical — replayed as a LAUNCH PLAN
(csrc/plan.hip; the sort on the plan's side stream) and as a hipGraph."""
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


@pytest.fixture(params=["plan", "hipgraph"])
def backend(request, monkeypatch):
    monkeypatch.setenv("RP_GRAPH_BACKEND", request.param)
    return request.param


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
        b["task1_label"] = (torch.rand(B, generator=gen) < 0.3).float()
        b["task2_label"] = (torch.rand(B, generator=gen) < 0.1).float()
        out.append({k: v.to(DEV) for k, v in b.items()})
    return out


def _build(kind, enc):
    from rec_pangu_amd.models.ranking import DCN, DeepFM
    torch.manual_seed(0)
    if kind == "deepfm64":   # D = 64, [64, 64, 64]: the fused gather + Linear forward and the fused gather backward
        model = DeepFM(embedding_dim=64, hidden_units=[64, 64, 64], enc_dict=enc)
    elif kind == "deepfm32tail":  # the fused MLP tail behind a GENERIC first layer: nobody joins the tail's deferred second
        model = DeepFM(embedding_dim=32, hidden_units=[64, 64, 64], enc_dict=enc)  # stage but the step itself
    elif kind == "xdeepfm_dropout":  # the reference's default: dropout 0.1 inside the MLP — ACTIVE in the captured step
        from rec_pangu_amd.models.ranking import xDeepFM
        model = xDeepFM(embedding_dim=16, dnn_hidden_units=[32, 16], cin_layer_units=[8, 8], enc_dict=enc)
    elif kind == "xdeepfm64":  # BASELINE config 3's structure (CIN [128, 128] at D = 64: pair-form first layer + collapsed last
        from rec_pangu_amd.models.ranking import xDeepFM  # layer) — round 6: library launches only, replays as a launch plan
        model = xDeepFM(embedding_dim=64, dnn_hidden_units=[64, 64, 64], cin_layer_units=[128, 128], enc_dict=enc)
    elif kind == "mmoe":  # the reference's defaults: BatchNorm1d + Dropout(0.2) towers, two tasks (round 5: a launch plan)
        from rec_pangu_amd.models.multi_task import MMOE
        model = MMOE(enc_dict=enc, embedding_dim=16, device=None)
    elif kind == "autoint":  # the fields as tokens beside the MLP (Fh.token_view), dropout 0.1, the LR layer's own tables
        from rec_pangu_amd.models.ranking import AutoInt
        model = AutoInt(embedding_dim=16, dnn_hidden_units=[32, 16], attention_layers=1, num_heads=2, attention_dim=4, enc_dict=enc)
    elif kind == "deepfm16":
        model = DeepFM(embedding_dim=16, hidden_units=[32, 16], enc_dict=enc)
    else:
        model = DCN(embedding_dim=16, hidden_units=[32, 16], enc_dict=enc)
    model = model.to(DEV)
    for m in model.modules():
        if hasattr(m, "check_indices"):
            m.check_indices = "deferred"
    return model


@pytest.mark.parametrize("kind,replay,steps,defer", [("deepfm64", "closed", 330, False), ("deepfm64", "exact", 60, False),
                                                     ("deepfm16", "closed", 300, False), ("dcn", "closed", 60, False),
                                                     ("deepfm32tail", "closed", 40, True), ("xdeepfm_dropout", "closed", 40, True),
                                                     ("deepfm64", "closed", 300, True), ("mmoe", "closed", 40, True),
                                                     ("autoint", "closed", 40, True), ("xdeepfm64", "closed", 40, True)])
def test_graphed_step_is_bit_identical_to_the_eager_loop(kind, replay, steps, defer, backend):
    """(330 / 300 steps cross step 256, where the closed-form replay takes over, and — with TABLE_CHUNK = 100 — several
    in-place extensions of the step tables; the learning rate changes twice on the way)"""
    from rec_pangu_amd import hip
    from rec_pangu_amd.graph_step import GraphedTrainStep
    from rec_pangu_amd.optim import FusedAdam, LazyAdamRows, StepTables
    enc = _enc(5, [3000, 17, 900, 4, 20000, 250])
    batches = _batches(enc, 384, steps + 1, seed=4)
    results = {}
    chunk, LazyAdamRows.TABLE_CHUNK = LazyAdamRows.TABLE_CHUNK, 100
    min_cap, StepTables.MIN_CAPACITY = StepTables.MIN_CAPACITY, 0  # (small tables: capacity doublings re-capture on the way)
    try:
        for mode in ("eager", "graph"):
            model = _build(kind, enc)
            opt = FusedAdam(model.parameters(), lr=1e-3, fuse_zero_grad=True, lazy_tables=True, replay=replay, defer=defer)
            gstep = GraphedTrainStep(model, opt) if mode == "graph" else None
            preds, losses = [], []
            for i in range(steps):
                if i in (40, 200):
                    for grp in opt.param_groups:
                        grp["lr"] *= 0.5
                if gstep is not None:
                    out = gstep(batches[i], batches[i + 1])
                else:
                    model.prefetch(batches[i + 1])
                    out = model(batches[i])
                    out["loss"].backward()
                    opt.step()
                    model.zero_grad()
                if i % 7 == 0 or i > steps - 4:
                    preds.append(out["pred" if "pred" in out else "task1_pred"].detach().clone())
                    losses.append(out["loss"].detach().clone())
            if gstep is not None:
                assert gstep.replays == steps - 2, "every step after the two eager ones must have been a graph replay"
                assert gstep.graphs[0] is not None and gstep.graphs[1] is not None
                if kind == "xdeepfm_dropout":
                    assert max(gstep._drop_calls) >= 1, "no dropout launch was captured: the model ran without active dropout"
                elif backend == "plan":
                    # DeepFM's step is library launches only: it must replay as a plan, the sort in the side section — and
                    # so are MMOE's and AutoInt's (BatchNorm statistics, loss sum, weight packing, token view: library
                    # launches) and DCN's since round 5 (the CrossNet's weight-space arithmetic is rp_crossnet_param_grads, its layer
                    # stack rp_multi_copy, the padding columns of dX_0 are written by the rows kernel)
                    assert gstep.backend_used == "plan" and gstep.plans[0].side >= 4, (gstep.backend_used, gstep.why_not_plan)
                else:
                    assert gstep.backend_used == "hipgraph"
            lz = model.embedding_layer._lazy
            assert lz.t == steps
            if mode == "graph":
                assert int(lz.tabs.t_dev.item()) == steps, "device and host step counters must agree"
            model.embedding_layer.raise_if_bad_index()
            sd = {k: v.clone() for k, v in model.state_dict().items()}  # (flushes the lazy rows: host counters must be right)
            osd = opt.state_dict()
            results[mode] = (preds, losses, sd, [{k: v.clone() for k, v in st.items() if torch.is_tensor(v)} for st in osd["state"].values()],
                             [g["_rp_step"] for g in osd["param_groups"]])
    finally:
        LazyAdamRows.TABLE_CHUNK = chunk
        StepTables.MIN_CAPACITY = min_cap
        from rec_pangu_amd.models.layers.embedding import EmbeddingLayer
        EmbeddingLayer.unpin_sorts()
    e, g = results["eager"], results["graph"]
    for a, b in zip(e[0], g[0]):
        assert torch.equal(a, b), "predictions differ"
    for a, b in zip(e[1], g[1]):
        assert torch.equal(a, b), "losses differ"
    for k in e[2]:
        assert torch.equal(e[2][k], g[2][k]), k
    for sa, sb in zip(e[3], g[3]):
        for k in sa:
            assert torch.equal(sa[k], sb[k]), f"optimizer state {k}"
    assert e[4] == g[4]
    assert hip.launch_count() > 0


def test_graphed_step_falls_back_to_eager_for_the_unannounced_and_the_last_batch(backend):
    """a batch that was not announced by the previous call is staged and sorted on the spot; a call without a next batch
    (end of an epoch) runs eagerly; the run continues on the graphs afterwards — all bit-identical to the eager loop"""
    from rec_pangu_amd.graph_step import GraphedTrainStep
    from rec_pangu_amd.optim import FusedAdam
    from rec_pangu_amd.models.layers.embedding import EmbeddingLayer
    enc = _enc(3, [500, 9, 4000])
    batches = _batches(enc, 256, 24, seed=9)
    order = [(i, i + 1) for i in range(8)] + [(8, None)] + [(12, 13), (13, 14), (20, 21), (14, 15), (15, None)]
    finals = {}
    try:
        for mode in ("eager", "graph"):
            model = _build("deepfm16", enc)
            opt = FusedAdam(model.parameters(), lr=2e-3, fuse_zero_grad=True, lazy_tables=True, replay="closed")
            gstep = GraphedTrainStep(model, opt) if mode == "graph" else None
            for cur, nxt in order:
                nb = batches[nxt] if nxt is not None else None
                if gstep is not None:
                    gstep(batches[cur], nb)
                else:
                    if nb is not None:
                        model.prefetch(nb)
                    model(batches[cur])["loss"].backward()
                    opt.step()
                    model.zero_grad()
            finals[mode] = {k: v.clone() for k, v in model.state_dict().items()}
            if gstep is not None:
                assert gstep.replays == len(order) - 2 - 2
    finally:
        EmbeddingLayer.unpin_sorts()
    for k in finals["eager"]:
        assert torch.equal(finals["eager"][k], finals["graph"][k]), k


def test_graphed_step_with_a_loader_that_refills_its_buffers_in_place():
    """a launch plan reads the caller's tensors when the replay RUNS, and the sort of a batch is made one call
    earlier from the same tensors.  A loader that recycles device buffers — here: the announced buffer is refilled in place
    with another batch between the call that announces it and the call that consumes it, every other step — must not make
    ids and sort disagree: the step sees the changed version counter and stages + re-sorts that batch.  Also: every input of
    the captured step has an argument site, and no recorded argument points inside an input."""
    from rec_pangu_amd.graph_step import GraphedTrainStep
    from rec_pangu_amd.optim import FusedAdam
    from rec_pangu_amd.models.layers.embedding import EmbeddingLayer
    enc = _enc(3, [500, 9, 4000])
    content = _batches(enc, 256, 16, seed=31)
    decoys = _batches(enc, 256, 16, seed=32)
    finals = {}
    try:
        for mode in ("eager", "graph"):
            model = _build("deepfm16", enc)
            opt = FusedAdam(model.parameters(), lr=2e-3, fuse_zero_grad=True, lazy_tables=True, replay="closed")
            if mode == "eager":
                for b in content[:-1]:
                    model(b)["loss"].backward()
                    opt.step()
                    model.zero_grad()
            else:
                gstep = GraphedTrainStep(model, opt, backend="plan")
                bufs = [{k: v.clone() for k, v in content[0].items()}, {k: torch.empty_like(v) for k, v in content[0].items()}]
                restaged = 0
                for i in range(len(content) - 1):
                    cur, nxt = bufs[i % 2], bufs[(i + 1) % 2]
                    late = i % 2 == 1
                    for k in nxt:  # the loader fills the buffer it announces ...
                        nxt[k].copy_((decoys if late else content)[i + 1][k])
                    before = gstep._staged_sig
                    gstep(cur, nxt)
                    if late:       # ... and, every other step, changes its mind afterwards: the same buffer, refilled in place
                        for k in nxt:
                            nxt[k].copy_(content[i + 1][k])
                        restaged += 1
                assert gstep.replays >= len(content) - 1 - 2 - 2 and restaged >= 6
                assert gstep.backend_used == "plan" and gstep.bind_report["interior"] == 0
                used = [k for k in gstep.bind_report["sites"]  # the step reads everything of its batch, the ids of the next one
                        if (k.startswith("cur:") and (k[4:] in enc or k == "cur:label")) or (k.startswith("next:C"))]
                assert used and all(gstep.bind_report["sites"][k] > 0 for k in used), gstep.bind_report
            finals[mode] = {k: v.clone() for k, v in model.state_dict().items()}
    finally:
        EmbeddingLayer.unpin_sorts()
    for k in finals["eager"]:
        assert torch.equal(finals["eager"][k], finals["graph"][k]), k


@pytest.mark.parametrize("kind", ["deepfm64", "deepfm16", "dcn"])
def test_catch_up_ahead_with_everything_that_can_come_between_two_steps(kind, monkeypatch):
    """RP_CATCHUP_AHEAD=1: a replayed step ends with the optimizer catch-up of the NEXT batch's rows (graph_step.py) — the
    rows are then stamped 'gradient coming' across the step boundary.  Everything a training loop does between two steps
    must leave the run bit-identical to the eager loop: an unannounced batch (the promise is dropped: the stamped rows must
    read as zero gradients), the last batch of an epoch (no next batch: an eager step), an evaluation forward of other rows,
    a state_dict() (flushes the lazy state), a change of the learning rate."""
    from rec_pangu_amd.graph_step import GraphedTrainStep
    from rec_pangu_amd.optim import FusedAdam
    from rec_pangu_amd.models.layers.embedding import EmbeddingLayer
    monkeypatch.setenv("RP_CATCHUP_AHEAD", "1")
    monkeypatch.setenv("RP_GRAPH_BACKEND", "plan")
    enc = _enc(3, [500, 9, 4000, 30, 12000])
    batches = _batches(enc, 320, 40, seed=11)
    # (current, announced next, what happens AFTER the step)
    order = [(i, i + 1, None) for i in range(6)] + [(6, 7, "eval"), (7, 8, None), (8, 9, "state_dict"), (9, 10, None),
             (10, 30, None),        # announces 30 ...
             (11, 12, None),        # ... but 11 arrives: the promise for 30's rows is dropped
             (12, 13, "lr"), (13, 14, None), (14, None, None),  # end of an epoch
             (20, 21, None), (21, 22, "eval"), (22, 23, None), (23, 24, None)]
    finals, preds = {}, {}
    try:
        for mode in ("eager", "graph"):
            model = _build(kind, enc)
            opt = FusedAdam(model.parameters(), lr=2e-3, fuse_zero_grad=True, lazy_tables=True, replay="closed", defer=True)
            gstep = GraphedTrainStep(model, opt) if mode == "graph" else None
            seen = []
            for cur, nxt, after in order:
                nb = batches[nxt] if nxt is not None else None
                if gstep is not None:
                    out = gstep(batches[cur], nb)
                else:
                    if nb is not None:
                        model.prefetch(nb)
                    out = model(batches[cur])
                    out["loss"].backward()
                    opt.step()
                    model.zero_grad()
                seen.append(out["loss"].detach().clone())
                if after == "eval":
                    model.eval()
                    with torch.no_grad():
                        seen.append(model(batches[35])["pred"].detach().clone())
                    model.train()
                elif after == "state_dict":
                    seen.append(torch.cat([v.detach().reshape(-1).float()[:64] for v in model.state_dict().values()]))
                elif after == "lr":
                    for grp in opt.param_groups:
                        grp["lr"] *= 0.5
            finals[mode] = {k: v.clone() for k, v in model.state_dict().items()}
            preds[mode] = seen
            if gstep is not None:
                assert gstep._ahead_used and gstep.backend_used == "plan", (gstep._ahead_used, gstep.backend_used, gstep.why_not_plan)
                assert gstep.replays >= len(order) - 2 - 3
    finally:
        EmbeddingLayer.unpin_sorts()
    for i, (a, b) in enumerate(zip(preds["eager"], preds["graph"])):
        assert torch.equal(a, b), f"observation {i} differs"
    for k in finals["eager"]:
        assert torch.equal(finals["eager"][k], finals["graph"][k]), k


def test_graphed_step_above_a_million_pairs(backend):
    """1.2 M (sample, field) pairs per batch: the size at which rocPRIM's onesweep sort faulted under unsynchronised
    replays (GraphedTrainStep.MAX_PAIRS_ROCPRIM).  The own radix sort has no memset nodes: 60 replays without a
    synchronisation in between end in the same bits as the eager loop."""
    from rec_pangu_amd.graph_step import GraphedTrainStep
    from rec_pangu_amd.optim import FusedAdam
    from rec_pangu_amd.models.layers.embedding import EmbeddingLayer
    from rec_pangu_amd.models.ranking import DeepFM
    enc = _enc(2, [3000000, 70000])
    batches = _batches(enc, 600000, 4, seed=4)
    finals = {}
    try:
        for mode in ("eager", "graph"):
            torch.manual_seed(0)
            model = DeepFM(embedding_dim=8, hidden_units=[16], enc_dict=enc).to(DEV)
            for m in model.modules():
                if hasattr(m, "check_indices"):
                    m.check_indices = "deferred"
            opt = FusedAdam(model.parameters(), lr=2e-3, fuse_zero_grad=True, lazy_tables=True, replay="closed")
            gstep = GraphedTrainStep(model, opt) if mode == "graph" else None
            for i in range(64):
                cur, nb = batches[i % 4], batches[(i + 1) % 4]
                if gstep is not None:
                    gstep(cur, nb)
                else:
                    model.prefetch(nb)
                    model(cur)["loss"].backward()
                    opt.step()
                    model.zero_grad()
            model.embedding_layer.raise_if_bad_index()
            finals[mode] = {k: v.clone() for k, v in model.state_dict().items()}
            if gstep is not None:
                assert gstep.replays >= 60
                del gstep
    finally:
        EmbeddingLayer.unpin_sorts()
    for k in finals["eager"]:
        assert torch.equal(finals["eager"][k], finals["graph"][k]), k


def test_graphed_step_refuses_what_it_cannot_capture():
    from rec_pangu_amd.graph_step import GraphedTrainStep
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.optim import FusedAdam
    enc = _enc(2, [50, 7])
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=8, hidden_units=[8], enc_dict=enc).to(DEV)  # check_indices is still "sync"
    opt = FusedAdam(model.parameters(), lr=1e-3, fuse_zero_grad=True, lazy_tables=True)
    with pytest.raises(RuntimeError, match="deferred"):
        GraphedTrainStep(model, opt)
    with pytest.raises(RuntimeError, match="FusedAdam"):
        GraphedTrainStep(model, torch.optim.Adam(model.parameters()))
    for m in model.modules():
        if hasattr(m, "check_indices"):
            m.check_indices = "deferred"
    # active dropout: the mask's (seed, offset) are launch arguments and would be frozen at capture
    from rec_pangu_amd import hip
    real = torch.cuda.is_current_stream_capturing
    torch.cuda.is_current_stream_capturing = lambda: True
    try:
        with pytest.raises(RuntimeError, match="dropout"):
            hip._dropout_seed_offset(torch.device(DEV))
    finally:
        torch.cuda.is_current_stream_capturing = real


@pytest.mark.parametrize("force_a2a,ahead", [(False, True), (True, True), (True, False)])
def test_graphed_step_with_row_sharded_tables_single_rank(force_a2a, ahead, backend, monkeypatch):
    """The captured step WITH its collectives: a DeepFM whose tables are row-sharded under a 1-rank RCCL group — route,
    exchange (the identity, or RCCL self-copies with RP_FORCE_A2A=1), owner-side gather / reduce, dense all-reduce, deferred
    lazy Adam with device-resident counters — against the eager loop on the same batches: every prediction and the final
    weights bit-identical.  Replayed as a hipGraph (round 4: all_to_all_single as graph nodes) and, round 6, as a LAUNCH PLAN
    in segments: the plan is cut at every collective (rp_plan_host_mark) and the replay issues them itself in between —
    which needs the whole sharded step to consist of library launches (no ATen fill / cast / cat / scale left in it)."""
    import copy
    import socket
    import torch.distributed as dist
    from rec_pangu_amd.graph_step import GraphedTrainStep
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.optim import make_adam
    from rec_pangu_amd.sharded import ShardedEmbeddingLayer, allreduce_dense_grads, shard_model_tables
    if force_a2a:
        monkeypatch.setenv("RP_FORCE_A2A", "1")
    if not ahead:  # (rounds 4-5: the route of a batch is built inside its own step — the path a step takes when the
        monkeypatch.setenv("RP_SHARD_AHEAD", "0")  # fixed-capacity buffers cannot be made)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    try:
        enc = _enc(3, [3000, 17, 900, 20000, 6])
        batches = _batches(enc, 512, 41, seed=8)
        torch.manual_seed(0)
        base = DeepFM(embedding_dim=64, hidden_units=[64, 64, 64], enc_dict=enc).to(DEV)
        results = {}
        for mode in ("eager", "graph"):
            model = shard_model_tables(copy.deepcopy(base), 1, 0)
            assert isinstance(model.embedding_layer, ShardedEmbeddingLayer)
            for m in model.modules():
                if hasattr(m, "check_indices"):
                    m.check_indices = "deferred"
            opt = make_adam(model, 1e-3)
            gstep = GraphedTrainStep(model, opt, post_backward=lambda: allreduce_dense_grads(model)) if mode == "graph" else None
            preds = []
            for i in range(40):
                if gstep is not None:
                    out = gstep(batches[i], batches[i + 1])
                else:
                    model.prefetch(batches[i + 1])
                    out = model(batches[i])
                    out["loss"].backward()
                    allreduce_dense_grads(model)
                    opt.step()
                    model.zero_grad()
                preds.append(out["pred"].detach().clone())
            model.embedding_layer.raise_if_bad_index()
            if gstep is not None:
                assert gstep.replays >= 36 and gstep.backend_used == backend, (gstep.replays, gstep.backend_used, gstep.why_not_plan)
                if backend == "plan":
                    # the three exchanges (ids, rows, row gradients) and the dense all-reduce are the replay's own calls
                    pl = gstep.plans[0]
                    assert sum(1 for fn in pl.host_calls if fn is not None) == (4 if force_a2a else 1), pl.host_calls
                    # (round 6) the next batch's route, id exchange and owner-side sort are segments of the ahead stream
                    if ahead:
                        assert sum(pl.seg_tags) == (2 if force_a2a else 1) and pl.ahead_stream is not None, pl.seg_tags
                    else:
                        assert sum(pl.seg_tags) == 0 and pl.ahead_stream is None, pl.seg_tags
            results[mode] = (preds, {k: v.clone() for k, v in model.state_dict().items()})
            del gstep
        for a, b in zip(results["eager"][0], results["graph"][0]):
            assert torch.equal(a, b), "predictions differ"
        for k in results["eager"][1]:
            assert torch.equal(results["eager"][1][k], results["graph"][1][k]), k
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()
