"""BaseModel — drop-in for rec_pangu/models/base_model.py:14-90 (ranking / multi-task base only).

Owns the arena-backed EmbeddingLayer; keeps the reference's two init schemes bit-for-bit (they
consume the global torch RNG in parameter-registration order, which is identical here), and adds the
shared tail of every ranking model: sum of logit branches -> sigmoid -> loss.
"""
import numpy as np
import torch
from torch import nn
from torch.nn.init import xavier_normal_, constant_

from .. import functional as Fh
from .layers.embedding import make_embedding_layer
from .utils import dense_columns


def build_loss(loss_fun):
    """The reference eval()s the `loss_fun` string (deepfm.py:31); same here, in a namespace that
    only exposes torch."""
    if isinstance(loss_fun, str):
        return eval(loss_fun, {"__builtins__": {}}, {"torch": torch, "nn": torch.nn})
    return loss_fun


def _is_plain_bce(loss) -> bool:
    return type(loss) is torch.nn.BCELoss and loss.weight is None and loss.reduction == "mean"


class BaseModel(nn.Module):
    def __init__(self, enc_dict: dict, embedding_dim: int) -> None:
        super().__init__()
        self.enc_dict = enc_dict
        self.embedding_dim = embedding_dim
        self.embedding_layer = make_embedding_layer(self.enc_dict, self.embedding_dim)

    # ---- init schemes (base_model.py:28-59) ---------------------------------------------------
    def _init_weights(self, module: nn.Module) -> None:
        if isinstance(module, nn.Embedding):
            xavier_normal_(module.weight.data)
        elif hasattr(module, "init_tables"):  # row-sharded layer: the same per-table draws, kept only for its own rows
            module.init_tables(xavier_normal_)
        elif isinstance(module, nn.Linear):
            xavier_normal_(module.weight.data)
            if module.bias is not None:
                constant_(module.bias.data, 0)

    def reset_parameters(self):
        """kaiming_normal_ on every >=2-D parameter (embedding tables included); 1-D left as built."""
        for weight in self.parameters():
            if len(weight.shape) == 1:
                continue
            shard = getattr(weight, "_rp_shard_init", None)
            if shard is not None and shard() is not None:
                # the local shard of row-sharded tables stands where the F table Parameters stand in parameters():
                # draw the F tables in that order (one temporary at a time), keep this rank's rows
                shard().init_tables(torch.nn.init.kaiming_normal_)
                continue
            torch.nn.init.kaiming_normal_(weight)

    def set_pretrained_weights(self, col_name: str, pretrained_dict: dict, trainable: bool = True) -> None:
        """base_model.py:61-90, including its quirk: the matrix has vocab_size rows (one short of the
        vocab_size+1 table it replaces), so the OOV id has no row afterwards."""
        assert col_name in self.enc_dict.keys(), "Pretrained Embedding Col: {} must be in the {}".format(
            col_name, self.enc_dict.keys())
        dim = len(list(pretrained_dict.values())[0])
        assert self.embedding_dim == dim, "Pretrained Embedding Dim:{} must be equal to Model Embedding Dim:{}".format(
            dim, self.embedding_dim)
        mat = np.random.rand(self.enc_dict[col_name]["vocab_size"], dim)
        for k, v in self.enc_dict[col_name].items():
            if k == "vocab_size":
                continue
            mat[v, :] = pretrained_dict.get(k, np.random.rand(dim))
        self.embedding_layer.set_weights(col_name=col_name,
                                         embedding_matrix=torch.nn.Parameter(torch.from_numpy(mat).float()),
                                         trainable=trainable)

    def zero_grad(self, set_to_none: bool = True) -> None:
        """nn.Module.zero_grad walks named_modules() / named_parameters() on every call (~0.1 ms of host time per step at
        26 tables + the MLP).  Same effect over a cached (module, name, parameter) list; a Parameter object that has been
        replaced since (set_weights, load_state_dict(assign=True)) is noticed by identity and the list rebuilt.
        (Submodules registered after the first call are picked up when any parameter changes — or call
        `model.__dict__.pop("_zg", None)`.)"""
        if not set_to_none:
            # the deferred table optimizer keeps gradient rows that still wait for their step in the gradient arena:
            # zeroing them in place would drop those updates — they are applied first (a flush)
            for m in self.modules():
                lz = getattr(m, "_lazy", None)
                if lz is not None and getattr(lz, "defer", False):
                    m.flush_lazy()
            return super().zero_grad(set_to_none=False)
        plan = self.__dict__.get("_zg")
        if plan is not None:
            for m, name, p in plan:
                if m._parameters.get(name) is not p:
                    plan = None
                    break
        if plan is None:
            plan = self.__dict__["_zg"] = [(m, name, p) for m in self.modules() for name, p in m._parameters.items()
                                           if p is not None]
        for _, _, p in plan:
            if p.grad is not None:
                p.grad = None

    # ---- shared pieces of the forward ---------------------------------------------------------
    @property
    def on_hip(self) -> bool:
        return self.embedding_layer.arena.is_cuda

    def prefetch(self, data) -> None:
        """Optional, HIP only: announce the batch of the NEXT step (the same dict of device tensors the next forward will
        get).  Its row sort starts on a side stream and overlaps the current step (EmbeddingLayer.prefetch_sort)."""
        fn = getattr(self.embedding_layer, "prefetch_sort", None)
        if fn is not None and self.on_hip:
            fn(data)

    def _dense_list(self, data):
        return [data[c] for c in dense_columns(self.enc_dict)]

    def _finish(self, logits, data, is_training, loss_fun):
        """y = sigmoid(sum(logits)); loss = loss_fun(y.squeeze(-1), label)  (e.g. deepfm.py:61-66)."""
        if logits[0].is_cuda:
            if is_training and _is_plain_bce(loss_fun):
                pred, loss = Fh.sigmoid_bce(logits, data["label"].float())
                return {"pred": pred, "loss": loss}
            pred = Fh.sigmoid_sum(logits)
        else:
            z = logits[0]
            for t in logits[1:]:
                z = z + t
            pred = torch.sigmoid(z)
        if is_training:
            return {"pred": pred, "loss": loss_fun(pred.squeeze(-1), data["label"])}
        return {"pred": pred}
