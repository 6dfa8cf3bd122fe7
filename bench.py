#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on its config: samples/s of a DeepFM train step
(forward + backward + the reference's dense Adam + zero_grad) on synthetic Criteo-shaped batches,
global batch 65536, 26 sparse fields (Criteo-Kaggle cardinalities, 33.76 M rows) x D=64 + 13 dense,
MLP [64,64,64], fp32, on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on stdout (rank 0).  Inputs are resident in HBM before the timed region.  Besides the
contract's fields it carries `roofline` (dominant kernel, HIP events on the launch stream inside the
timed region), `kernels` (every C-ABI entry point: calls/step, mean ms, algorithmic GB/s where defined)
and `cpu_baseline` (the CPU oracle port timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CRITEO_CARD = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306,
               10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 matrix-core peak (MI355X_MICROARCH.md; v_mfma_f32_32x32x16_bf16)


def criteo_enc_dict(scale=1):
    enc = {f"I{i + 1}": {"min": 0.0, "max": 1.0} for i in range(13)}
    enc.update({f"C{i + 1}": {"vocab_size": max(2, c // scale)} for i, c in enumerate(CRITEO_CARD)})
    return enc


def synth_batch(enc, B, seed, device, id_dist="uniform"):
    """One synthetic batch, generated on `device` (a CPU generator would take seconds per 65536 x 40 batch and the
    bench draws a DIFFERENT batch for every step: with a handful of recycled batches every embedding row would be
    revisited within a few steps, which hides the cost of the lazy optimizer's replay of skipped steps)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    b = {}
    for k, v in enc.items():
        if "min" in v:
            b[k] = torch.rand(B, generator=g, device=dev)
        elif id_dist == "uniform":  # uniform ids: worst case for caches, dedup and the lazy optimizer
            b[k] = torch.randint(0, v["vocab_size"] + 1, (B,), generator=g, device=dev)
        else:  # "zipf": bounded power law with exponent 1.05 (SURVEY.md 8d's optional skewed run), id 0 the hottest
            V1, s_ = float(v["vocab_size"] + 1), 1.05
            u = torch.rand(B, generator=g, device=dev, dtype=torch.float64)
            x = (1.0 + u * ((V1 + 1.0) ** (1.0 - s_) - 1.0)) ** (1.0 / (1.0 - s_))
            b[k] = (x.floor().long() - 1).clamp_(0, v["vocab_size"])
    b["label"] = (torch.rand(B, generator=g, device=dev) < 0.25).float()
    b["task1_label"] = b["label"]
    b["task2_label"] = (torch.rand(B, generator=g, device=dev) < 0.1).float()
    return b


def mmoe_enc_dict(scale=1):
    """BASELINE config 3: the multi-task example schema shape (16 sparse + 9 dense); cardinalities = the first 16
    Criteo ones (the reference's 100-row sample has no meaningful cardinalities of its own)."""
    enc = {f"I{i + 1}": {"min": 0.0, "max": 1.0} for i in range(9)}
    enc.update({f"C{i + 1}": {"vocab_size": max(2, c // scale)} for i, c in enumerate(CRITEO_CARD[:16])})
    return enc


def build_model(name, enc, hidden=(64, 64, 64)):
    """The BASELINE.json configs: 1 = DeepFM (headline), 2 = xDeepFM CIN [128,128], 3 = MMOE 4 experts, towers
    [256,128]; DCN / AutoInt with their reference defaults at D=64 for completeness."""
    from rec_pangu_amd.models.ranking import DeepFM, xDeepFM, DCN, AutoInt
    from rec_pangu_amd.models.multi_task import MMOE
    if name == "deepfm":
        return DeepFM(embedding_dim=64, hidden_units=list(hidden), enc_dict=enc)
    if name == "xdeepfm":
        return xDeepFM(embedding_dim=64, dnn_hidden_units=[64, 64, 64], cin_layer_units=[128, 128], enc_dict=enc)
    if name == "dcn":
        return DCN(embedding_dim=64, crossing_layers=3, enc_dict=enc)
    if name == "autoint":
        return AutoInt(embedding_dim=64, dnn_hidden_units=[64, 64, 64], attention_layers=1, num_heads=1,
                       attention_dim=8, enc_dict=enc)
    if name == "mmoe":
        return MMOE(num_task=2, n_expert=4, embedding_dim=40, mmoe_hidden_dim=128, hidden_dim=[256, 128], enc_dict=enc)
    raise ValueError(name)


def pmc_traffic(entry):
    """HBM bytes per launch of the kernel behind a C-ABI entry point, from the committed rocprofv3 PMC summary of
    this same workload (profiles/r*_pmc.json: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes).
    bench.py cannot run the profiler on itself, so this is a recorded figure; null when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None
    try:
        pmc = json.load(open(files[-1]))
    except Exception:
        return None
    hits = [v["hbm_bytes_per_launch"] for k, v in pmc.items()
            if k.startswith(entry + "_kernel") and "hbm_bytes_per_launch" in v]
    return hits[0] if len(hits) == 1 else None  # several template variants behind one entry point: ambiguous


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle port (oracle/ref_ops.py: the reference's algorithm in ATen fp32 ops + autograd, dense
    torch.optim.Adam as trainer.py:75) on this box's host cores.  Bounded sample: B=65536, vocabulary / 16
    (dense Adam then touches 2.1 M rows instead of 33.8 M), 1 warm-up + up to 2 timed steps."""
    from oracle import ref_ops as R  # checker/baseline only
    from rec_pangu_amd.models.ranking import DeepFM
    cores = min(os.cpu_count() or 1, 64)  # more threads than this only adds contention in ATen's scatter ops
    torch.set_num_threads(cores)
    enc = criteo_enc_dict(scale=16)
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=64, hidden_units=[64, 64, 64], enc_dict=enc)
    params = {k: torch.nn.Parameter(v.clone()) for k, v in model.state_dict().items()}
    del model
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    B = 65536
    batch = synth_batch(enc, B, 1, "cpu")

    def step():
        out = R.deepfm(params, enc, batch)
        out["loss"].backward()
        opt.step()
        opt.zero_grad()

    step()
    t0, n = time.perf_counter(), 0
    while n < 2 and (n == 0 or time.perf_counter() - t0 < seconds_budget):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(B / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} train step(s) of B={B} (fwd+bwd+dense Adam), vocabulary/16 = "
                      f"{sum(v['vocab_size'] + 1 for v in enc.values() if 'vocab_size' in v)} rows, "
                      f"{dt:.2f} s/step, torch CPU fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="samples per GPU per step (weak scaling)")
    ap.add_argument("--vocab-scale", type=int, default=1, help="divide every cardinality (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--id-dist", default="uniform", choices=["uniform", "zipf"],
                    help="sparse id distribution: uniform (default, worst case) or a bounded Zipf(1.05)")
    ap.add_argument("--mode", default="train", choices=["train", "forward"])
    ap.add_argument("--model", default="deepfm", choices=["deepfm", "xdeepfm", "dcn", "autoint", "mmoe"],
                    help="deepfm = BASELINE headline config; the others are BASELINE configs 2-3 / siblings")
    ap.add_argument("--hidden", default="64,64,64",
                    help="DeepFM hidden_units; the reference default 64,64,64 is HBM-bound (SURVEY D5), "
                         "1024,512,256 is the MFMA-bound variant BASELINE's MLP-utilisation target refers to")
    ap.add_argument("--sharded", action="store_true",
                    help="take the row-sharded all-to-all path even with one rank (validates the N>1 code on 1 GPU)")
    ap.add_argument("--optimizer", default="lazy", choices=["lazy", "dense"],
                    help="how the reference's dense Adam is executed on the embedding arena: 'lazy' = exact lazy "
                         "replay (bit-identical results, flushed inside the timed region), 'dense' = stream every row")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.sharded
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:  # single process, --sharded
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=dev)

    from rec_pangu_amd import hip
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.optim import make_adam
    hip.lib()

    enc = mmoe_enc_dict(args.vocab_scale) if args.model == "mmoe" else criteo_enc_dict(args.vocab_scale)
    B = args.batch
    if sharded:
        from rec_pangu_amd.sharded import shard_model_tables, allreduce_dense_grads  # row-sharded tables + RCCL
    torch.manual_seed(0)
    with torch.device(dev):
        hidden = tuple(int(h) for h in args.hidden.split(","))
        model = build_model(args.model, enc, hidden)
    if sharded:
        model = shard_model_tables(model, world, rank)
    for m in model.modules():
        if hasattr(m, "check_indices"):
            m.check_indices = "deferred"  # no per-step host sync; checked once after the run
    model.train()
    opt = make_adam(model, 1e-3, lazy_tables=(args.optimizer == "lazy"))
    n_params = sum(p.numel() for p in model.parameters())
    n_table_params = sum(p.numel() for m in model.modules() if hasattr(m, "table_parameters") for p in m.table_parameters())
    n_table_rows = model.embedding_layer.arena.shape[0]

    # WEAK scaling: every GPU works on its own `--batch` samples (65536, the configuration BASELINE.json quotes),
    # so the global batch is world x 65536 and per-GPU work is constant as N grows.  Each rank draws its own
    # batches; the tables are row-sharded and the lookup all-to-all serves the whole global batch.
    local_B = args.batch
    B = local_B * world
    # a distinct batch per step (capped at 512 batches = 8.8 GB of ids at Criteo shape)
    n_batches = min(args.steps + args.warmup, 512)
    batches = [synth_batch(enc, local_B, 100 + 100003 * rank + i, dev, args.id_dist) for i in range(n_batches)]

    def step(i):
        data = batches[i % len(batches)]
        if args.mode == "forward":
            with torch.no_grad():
                model(data, is_training=False)
            return
        out = model(data)
        out["loss"].backward()
        if sharded:
            allreduce_dense_grads(model)  # one flat bucket; embedding-row grads already travelled in backward
        opt.step()
        model.zero_grad()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Warm-up; its last few steps double as the per-kernel profiling pass (a HIP-event pair around EVERY launch —
    # that serialises the queue and costs ~45 % of the step, so it stays out of the timed region).
    n_prof = min(3, args.warmup)
    for i in range(args.warmup - n_prof):
        step(i)
    barrier()
    hip.enable_timing(True)
    for i in range(args.warmup - n_prof, args.warmup):
        step(i)
    barrier()
    prof = hip.timing_summary() if n_prof else None
    hip.enable_timing(False)
    # Timed region: events only around the launches the roofline objects report (dominant kernel, gather, GEMMs).
    if prof is not None:
        ours = {n: c * m for n, (c, m) in prof.items() if n not in ("lazy_adam_flush",)}
        watch = set(sorted(ours, key=ours.get, reverse=True)[:2]) | {"embed_gather_fwd", "linear_fwd", "linear_wgrad"}
        hip.enable_timing(True, only=watch)
    else:
        hip.enable_timing(True)
    ev_stride = 4 if (n_prof and args.steps >= 8) else 1  # events on every 4th timed step only
    t0 = time.perf_counter()
    for i in range(args.steps):
        hip.pause_timing(i % ev_stride != 0)
        step(args.warmup + i)
    hip.pause_timing(False)
    if args.mode == "train" and hasattr(opt, "flush"):
        opt.flush()  # lazy Adam: every row is brought to step K INSIDE the timed region (dense-equivalent state)
    barrier()
    dt = time.perf_counter() - t0
    timing = hip.timing_summary()
    hip.enable_timing(False)
    if prof is None:
        prof, n_prof = timing, args.steps
    for m in model.modules():
        if hasattr(m, "raise_if_bad_index"):
            m.raise_if_bad_index()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * args.steps / dt

    # ---- per-kernel numbers (algorithmic bytes from SURVEY.md §8d) --------------------------------
    F = sum(1 for v in enc.values() if "vocab_size" in v)
    ND = sum(1 for v in enc.values() if "min" in v)
    D = model.embedding_dim
    n_unique = int(sum(torch.unique(batches[0][f"C{i + 1}"]).numel() for i in range(F)))
    n_pairs = F * local_B
    row_b = D * 4
    alg_bytes = {
        # dX row per pair (+ the sum_f v row per pair unless the FM term is folded into the dgrad: DeepFM at D = 64),
        # table row read + gradient row written per unique row
        "embed_grad_reduce": ((1 if (args.model == "deepfm" and D == 64) else 2) * n_pairs + 2 * n_unique) * row_b,
        # p,m,v read+written, g read + cleared, per unique touched row
        "lazy_adam_rows_step": 8 * n_unique * row_b,
        # p,m,v read+written per unique row that skipped at least one step (upper bound: all of them)
        "lazy_adam_rows_replay": 6 * n_unique * row_b,
        # (key, position) pairs read + written once per radix pass (26 key bits -> 4 passes of <= 8 bits)
        "sort_pairs_i32": 4 * 2 * 8 * n_pairs,
        # table rows read + int64 ids read + [B, F*D+ND] fp32 output written
        "embed_gather_fwd": local_B * (F * (D * 4 + 8) + (F * D + ND) * 4),
        # dense Adam: read p,g,m,v + write p,m,v = 7 fp32 streams over every parameter
        "adam_step": 7 * 4 * (n_params - (n_table_params if args.optimizer == "lazy" else 0)),
    }
    if getattr(model, "lr_layer", None) is not None:
        # the LR_Layer's 1-wide tables go through the same entries once more per step: report the mean per launch
        def both(f):
            return (f(D) + f(1)) / 2
        alg_bytes["embed_gather_fwd"] = both(lambda d_: local_B * (F * (d_ * 4 + 8) + (F * d_ + ND) * 4))
        alg_bytes["embed_grad_reduce"] = both(lambda d_: (2 * n_pairs + 2 * n_unique) * d_ * 4)
        alg_bytes["lazy_adam_rows_step"] = both(lambda d_: 8 * n_unique * d_ * 4)
        alg_bytes["lazy_adam_rows_replay"] = both(lambda d_: 6 * n_unique * d_ * 4)
    d_in = F * D + ND
    alg_bytes["crossnet_fwd"] = local_B * d_in * 4                      # X_0 read once, only a logit leaves
    alg_bytes["crossnet_bwd_rows"] = 2 * local_B * d_in * 4             # X_0 read, dX_0 written
    if args.model == "mmoe":
        K_, E_, T_ = model.mmoe_hidden_dim, model.n_expert, model.num_task
        alg_bytes["mmoe_combine_fwd"] = local_B * 4 * (K_ * E_ + T_ * E_ + T_ * K_)
        alg_bytes["mmoe_combine_bwd"] = local_B * 4 * (2 * (K_ * E_ + T_ * E_) + T_ * K_)
    if args.model == "autoint":
        att = model.self_attention[0]
        npj = 4 if att.W_res is not None else 3
        # split form: QKVR in (+ residual rows when there is no W_res), out + row statistics back; backward: QKVR,
        # out, dout in, dQKVR out
        alg_bytes["attention_core_fwd"] = local_B * 4 * (F * (npj + 1) * att.output_dim + 2 * att.num_heads * F)
        alg_bytes["attention_core_bwd"] = local_B * 4 * (F * (2 * npj + 2) * att.output_dim + 2 * att.num_heads * F)
        alg_bytes["field_attention_fwd"] = local_B * 4 * F * (D + att.output_dim)
        alg_bytes["field_attention_bwd"] = local_B * 4 * F * (2 * D + att.output_dim)
    mfma_flops = {}
    if args.model == "xdeepfm":
        units, Mi, per = list(model.cin.cin_layer_units), F, 0
        for i, O_ in enumerate(units):
            per += 2 * F * Mi * (O_ if i + 1 < len(units) else 1) * D   # the last layer runs collapsed to O = 1
            Mi = O_
        # flop per launch averaged over the launches of a step (fwd: one chain; bwd_x: two chains; bwd_w: one)
        mfma_flops = {"cin_layer_fwd": local_B * per / len(units), "cin_layer_bwd_x": 2 * local_B * per / len(units),
                      "cin_layer_bwd_w": local_B * per / len(units)}
        # first layer on the bf16 matrix core (rp_cin_bs_*): 2*H*M*O*D flop per sample per pass (bwd_x: ONE pass with the
        # symmetrised weights), collapsed last layer (rp_cin_last_*): HBM-bound on X_{L-1}
        f1 = 2.0 * F * F * units[0] * D * local_B
        # ... or in the pair form (rp_cin_pair_*): ONE GEMM over the F(F+1)/2 pair products per pass, 2*O*npair*D flop/sample
        fp_ = 2.0 * units[0] * (F * (F + 1) // 2) * D * local_B
        bf16_mfma_flops = {"cin_bs_fwd": f1, "cin_bs_bwd_x": f1, "cin_bs_bwd_w": f1,
                           "cin_pair_fwd": fp_, "cin_pair_bwd_x": fp_, "cin_pair_bwd_w": fp_}
        if len(units) > 1:
            xl = local_B * units[-2] * D * 4
            alg_bytes["cin_last_fwd"] = xl + local_B * F * D * 4
            alg_bytes["cin_last_bwd_x"] = 2 * (xl + local_B * F * D * 4)
            alg_bytes["cin_last_bwd_v"] = xl + local_B * F * D * 4
    else:
        bf16_mfma_flops = {}
    mlp_flops = None
    if args.model == "deepfm" and hidden != (64, 64, 64):
        dims = [F * D + ND] + list(hidden) + [1]
        # forward + dgrad launches go through linear_fwd (2 flop per MAC), wgrad through linear_wgrad
        mlp_flops = {"linear_fwd": 2 * 2 * local_B * sum(a * b for a, b in zip(dims[:-1], dims[1:])),
                     "linear_wgrad": 2 * local_B * sum(a * b for a, b in zip(dims[:-1], dims[1:]))}
    if args.model == "deepfm" and hidden == (64, 64, 64):
        # mean algorithmic bytes per launch over the launches of one step (activations in + out, fp32);
        # forward 1677->64->64->64->1 plus the four dgrad launches on the transposed weights / four wgrad launches
        alg_bytes["linear_fwd"] = local_B * 4 * 2 * ((d_in + 64) + 2 * (64 + 64) + (64 + 1)) // 8
        alg_bytes["linear_wgrad"] = local_B * 4 * ((d_in + 64) + 2 * (64 + 64) + (64 + 1)) // 4
    kernels = {}
    for name, (calls, mean_ms) in sorted(prof.items()):  # the profiling pass (every launch bracketed by events)
        k = {"calls_per_step": round(calls / n_prof, 2), "mean_ms": round(mean_ms, 4)}
        if name in alg_bytes:
            k["algorithmic_GBps"] = round(alg_bytes[name] / (mean_ms * 1e-3) / 1e9, 1)
        kernels[name] = k
    # dominant = largest share of the step among the launches timed INSIDE the timed region
    total = {n: c * m for n, (c, m) in timing.items() if n in alg_bytes or n in mfma_flops or n in bf16_mfma_flops}
    dominant = max(total, key=total.get) if total else None
    roofline = None
    if dominant in bf16_mfma_flops:
        tf = bf16_mfma_flops[dominant] / (timing[dominant][1] * 1e-3) / 1e12
        roofline = {"kernel": dominant, "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF,
                    "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "traffic": None,
                    "matmul_precision": "bf16x6", "mfma_products_per_flop": 6,
                    "mfma_issue_frac": round(tf * 6 * ((32 * 32) / (F * F) if "cin_bs" in dominant else 1.0)
                                             / MFMA_BF16_PEAK_TF, 4),
                    "note": ("algorithmic flops 2*H*M*O*D per sample; the matrix core runs 32x32 tiles (H, M padded from "
                             f"{F}) with 6 bf16 products per flop: mfma_issue_frac counts those") if "cin_bs" in dominant
                    else ("pair form: algorithmic flops 2*O*(H(H+1)/2)*D per sample for this pass, 6 bf16 products per "
                          "flop on the matrix core (mfma_issue_frac)")}
    elif dominant in mfma_flops:
        tf = mfma_flops[dominant] / (timing[dominant][1] * 1e-3) / 1e12
        roofline = {"kernel": dominant, "bound": "mfma", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s",
                    "frac": round(tf / 157.3, 4), "traffic": None,
                    "note": "exact-fp32 MFMA peak; flops per launch = mean over this entry's launches in a step"}
    elif dominant in alg_bytes:
        a = alg_bytes[dominant] / (timing[dominant][1] * 1e-3) / 1e9
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(a / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic(dominant) if (args.model == "deepfm" and world == 1) else None,
                    "algorithmic_bytes_per_launch": alg_bytes[dominant]}
    if mlp_flops is not None:
        # MFMA-bound variant: the GEMM launches against the dense bf16 matrix-core peak (the GEMMs run on
        # v_mfma_f32_32x32x16_bf16 with split-bf16 operands: `achieved` counts ALGORITHMIC flops, the matrix core
        # issues `mfma_products_per_flop` bf16 products for each of them)
        n_timed_steps = len(range(0, args.steps, ev_stride))  # events were recorded on these steps only
        tot_ms = {n: timing[n][0] * timing[n][1] / n_timed_steps for n in mlp_flops if n in timing}
        if tot_ms:
            n = max(tot_ms, key=tot_ms.get)
            tf = mlp_flops[n] / (tot_ms[n] * 1e-3) / 1e12
            nprod = {"bf16x6": 6, "bf16x3": 3, "bf16": 1, "fp32": 1}[hip.get_matmul_precision()]
            peak = 157.3 if hip.get_matmul_precision() == "fp32" else MFMA_BF16_PEAK_TF
            roofline = {"kernel": n, "bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(tf / peak, 4), "traffic": None,
                        "matmul_precision": hip.get_matmul_precision(), "mfma_products_per_flop": nprod,
                        "mfma_issue_frac": round(tf * nprod / peak, 4),
                        "note": "flops = all launches of this entry per step; fp32 operands, fp32 accumulation"}
    gather = None
    if "embed_gather_fwd" in timing and prof["embed_gather_fwd"][0] == n_prof:  # exactly one gather launch per step
        a = alg_bytes["embed_gather_fwd"] / (timing["embed_gather_fwd"][1] * 1e-3) / 1e9
        gather = {"kernel": "embed_gather_fwd", "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
                  "traffic": pmc_traffic("embed_gather_fwd") if (args.model == "deepfm" and world == 1) else None,
                  "algorithmic_bytes_per_launch": alg_bytes["embed_gather_fwd"]}

    if rank == 0:
        res = {
            "metric": f"samples/sec {type(model).__name__} Criteo-shape bsz={local_B}/GPU (train step: fwd+bwd+dense Adam+zero_grad)"
                      if args.mode == "train" else f"samples/sec {type(model).__name__} Criteo-shape bsz={local_B}/GPU (forward only)",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{type(model).__name__} ({args.model}), {F} sparse fields (Criteo-Kaggle "
                                   f"cardinalities/{args.vocab_scale}, "
                                   f"{n_table_rows * world if world > 1 else n_table_rows} arena rows) x D={D} + {ND} dense, "
                                   f"batch {local_B} per GPU (global {B}), "
                                   + ("uniform ids" if args.id_dist == "uniform" else "bounded Zipf(1.05) ids")
                                   + (f", MLP {list(hidden)}" if args.model == "deepfm" else
                                      (", MLP [64,64,64]" if args.model != "mmoe" else ", 4 experts x 128, towers [256,128]"))
                                   + (", CIN [128,128]" if args.model == "xdeepfm" else ""),
                       "global_batch": B, "per_gpu_batch": local_B,
                       "optimizer": ("dense Adam, reference semantics, executed lazily (bit-identical; all rows flushed "
                                     "to the last step inside the timed region)" if args.optimizer == "lazy"
                                     else "dense Adam (reference semantics, every row streamed each step, fused zero_grad)"),
                       "unique_rows_per_batch": n_unique,
                       "parallelism": "single GPU" if not sharded else f"tables row-sharded x{world}, all-to-all lookup"},
            "roofline": roofline, "roofline_gather": gather, "kernels": kernels,
            "kernels_note": f"per-kernel table: HIP events around every launch during the last {n_prof} warm-up steps; "
                            f"roofline/roofline_gather durations: HIP events inside the timed region, every "
                            f"{ev_stride}th step",
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline()
    # the JSON line is the LAST thing on stdout: RCCL's init banner sits in the C stdio buffer until exit otherwise
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if sharded:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
