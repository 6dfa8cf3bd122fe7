"""torch.autograd.Function wrappers that put the HIP kernels (rec_pangu_amd/hip.py) on the autograd tape.

Everything here takes HIP-device fp32 tensors; nothing in this module computes on the CPU and
nothing falls back: a missing librecpangu_hip.so raises from hip.lib().
"""
import os
from typing import List, Sequence

import torch

from . import hip

ACT_NONE, ACT_RELU, ACT_MASK = hip.ACT_NONE, hip.ACT_RELU, hip.ACT_MASK
ACT_TANH, ACT_SIGMOID, ACT_LEAKY = hip.ACT_TANH, hip.ACT_SIGMOID, hip.ACT_LEAKY


def act_code(module):
    """the GEMM-epilogue code of an activation module (rec_pangu/models/layers/activation.py:37-59 builds them by name),
    or None: ReLU, Tanh, Sigmoid, LeakyReLU at its default slope"""
    import torch.nn as nn
    if isinstance(module, nn.ReLU):
        return ACT_RELU
    if isinstance(module, nn.Tanh):
        return ACT_TANH
    if isinstance(module, nn.Sigmoid):
        return ACT_SIGMOID
    if isinstance(module, nn.LeakyReLU) and abs(module.negative_slope - 0.01) < 1e-12:
        return ACT_LEAKY
    return None


def _unit_inner(t: torch.Tensor) -> torch.Tensor:
    """Kernels take row-major 2-D views with unit inner stride (any leading dimension)."""
    if t.dim() == 2 and (t.shape[1] <= 1 or t.stride(1) == 1) and (t.shape[0] <= 1 or t.stride(0) >= t.shape[1]):
        return t
    return t.contiguous()


# ----------------------------------------------------------------------------------------------
# K4  Linear (+bias, +ReLU)   — layers/deep.py:62-72
# ----------------------------------------------------------------------------------------------
def _rows16(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear keeps W as [N, K] with row stride K; when K is not a multiple of 4 (DeepFM: 1677) its rows are not
    16-byte aligned and the GEMM would have to fetch W with scalar loads.  A [N, ceil4(K)] staging copy (a few
    hundred KB, one small launch) restores dwordx4 loads; the extra columns are never read (K is passed on)."""
    K = w.shape[1]
    if K % 4 == 0 or K <= 64 or not w.is_cuda:
        return w
    # one staging copy per weight VALUE: the key is torch's version counter plus hip.weight_epoch(), which the fused
    # optimizer kernels bump (they write parameters through raw pointers, invisibly to torch's counter).  DeepFM's forward
    # asks twice per step (the "does the fused launch fit" check and the launch itself).
    key = (w.data_ptr(), w._version, hip.weight_epoch(), torch.cuda.is_current_stream_capturing())
    hit = getattr(w, "_rp_rows16", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():  # (a cached tensor with a grad_fn would keep the weight's AccumulateGrad node alive)
        if w.dtype is torch.float32:
            out = hip.copy_rows(w.detach(), (K + 3) // 4 * 4)  # a library launch: part of a recorded launch plan
        else:
            buf = torch.empty((w.shape[0], (K + 3) // 4 * 4), dtype=w.dtype, device=w.device)
            buf[:, :K].copy_(w)
            out = buf[:, :K]
    if not key[3]:  # (inside a stream capture the copy must stay a node of every graph that uses it)
        try:
            w._rp_rows16 = (key, out)
        except AttributeError:
            pass
    return out


class FMFold:
    """Link between the gather node and the first Linear that consumes its output x (DeepFM): the FM part of the
    embedding gradient, g_fm[b] * S[b, :] added to every field's slice of dX[b], is folded into the dgrad GEMM that
    produces dX (rp_linear_fwd_rowadd) instead of being applied per (sample, field) pair in the segmented reduce.
    `dfm` is recorded by _GradTap (tap_fm_grad) when the loss backward produces it, before any Linear runs."""
    __slots__ = ("ssum", "ncols", "dfm", "folded", "dgrad", "D", "fused", "extra")

    def __init__(self, ssum, ncols, D=0):
        self.ssum, self.ncols, self.dfm, self.folded = ssum, ncols, None, False
        # the gradient of x[:, :ncols] from a consumer that read the embedding block as [B, F, D] tokens (token_view: AutoInt's
        # attention): left here by its backward, added into the other consumers' dX by the gather's backward
        self.extra = None
        # (dH, W^T) of the first Linear when its dgrad is left to the gather backward (rp_embed_grad_gemm): dX is then
        # never materialised; the Linear returns a stride-0 zero in its place
        self.dgrad, self.D, self.fused = None, D, False


class _TokenView(torch.autograd.Function):
    """x [B, ldx] (embedding block | dense | padding) -> a contiguous [B, F, D] copy of its first F*D columns (rp_copy_rows), for
    a consumer that reads the fields as tokens (AutoInt, autoint.py:44-46).  The slice + unflatten + reshape it replaces made a
    strided view whose re-pack was an ATen clone, and in the backward a zero-filled [B, ldx] slice gradient plus the ATen sum
    of the two consumers' gradients of x.  Here the token gradient is parked in the gather's link and the gather's backward
    adds it into the other consumer's dX with one library launch (rp_add_rows) — so x needs another consumer (the MLP)."""

    @staticmethod
    def forward(ctx, x, F: int, D: int, link):
        x = _unit_inner(x)
        out = torch.empty((x.shape[0], F * D), dtype=torch.float32, device=x.device)
        hip.copy_rows_to(x[:, :F * D], out)
        ctx.link = link
        return out.view(x.shape[0], F, D)

    @staticmethod
    def backward(ctx, dtok):
        ctx.link.extra = dtok.reshape(dtok.shape[0], -1).contiguous()
        return None, None, None, None


def token_view(x, F: int, D: int, link):
    return _TokenView.apply(x, F, D, link)


class _StackRows(torch.autograd.Function):
    """cat(ws, dim=0) of 2-D matrices with one column count, as ONE library launch (rp_multi_copy into the row blocks of a fresh
    buffer); the gradients leave as the row blocks (contiguous views) of the incoming one."""

    @staticmethod
    def forward(ctx, *ws):
        rows = [w.shape[0] for w in ws]
        out = torch.empty((sum(rows), ws[0].shape[1]), dtype=torch.float32, device=ws[0].device)
        dst, r0 = [], 0
        for r in rows:
            dst.append(out[r0:r0 + r])
            r0 += r
        src = [w.detach().contiguous() for w in ws]
        if not hip.multi_copy(dst, src):
            torch._foreach_copy_(dst, src)
        ctx.rows = rows
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        outs, r0 = [], 0
        for r in ctx.rows:
            outs.append(g[r0:r0 + r])
            r0 += r
        return tuple(outs)


def stack_rows(ws):
    return _StackRows.apply(*ws)


class _GradTap(torch.autograd.Function):
    """Identity whose backward records the incoming gradient in an FMFold.  Applied to the FM output AFTER the MLP
    forward, so that in the backward pass (later nodes run first) it fires before the MLP's first Linear computes dX.
    (A tensor hook would not do: hooks of a non-leaf tensor run only when its grad_fn — the gather, which also waits
    for dX — is about to execute.)"""

    @staticmethod
    def forward(ctx, t, link):
        ctx.link = link
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        ctx.link.dfm = g
        return g, None


def tap_fm_grad(fm, link):
    return _GradTap.apply(fm, link)


class ReluLink:
    """Hand-off between a Linear+ReLU (producer) and the Linear that consumes its output: the consumer's dgrad GEMM
    applies the producer's ReLU mask in its epilogue (RP_ACT_MASK with aux = its own input, y > 0 <=> pre > 0) and
    leaves the tensor here; the producer's backward skips its relu_bwd pass when the gradient it receives IS that
    tensor.  Holding the reference also keeps autograd from accumulating another consumer's gradient into it in
    place; a summed gradient is a different tensor and takes the ordinary path (masking twice is harmless)."""
    __slots__ = ("dx",)

    def __init__(self):
        self.dx = None


class _LinearAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act: int, fm_link=None, in_link=None, out_link=None):
        x = _unit_inner(x)
        K = weight.shape[1]
        y = hip.linear_fwd(x, _rows16(weight), bias, act, K=K)
        ctx.fm_link, ctx.in_link, ctx.out_link = fm_link, in_link, out_link
        ctx.act, ctx.K, ctx.has_bias = act, K, bias is not None
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _unit_inner(dy)
        dpre = dy
        if ctx.act == ACT_RELU:
            lk = ctx.out_link
            masked = lk is not None and lk.dx is not None and lk.dx.data_ptr() == dy.data_ptr() \
                and lk.dx.shape == dy.shape and lk.dx.stride() == dy.stride()
            if lk is not None:
                lk.dx = None
            if not masked:
                dpre = hip.relu_bwd(dy, y)
        elif ctx.act in (ACT_TANH, ACT_SIGMOID, ACT_LEAKY):
            dpre = hip.act_bwd(dy, y, ctx.act)  # through the activation's output: tanh 1 - y^2, sigmoid y (1 - y), leaky: sign
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # [ldx, N]: the dgrad GEMM is the same NT kernel on W^T; zero rows beyond K make it write the zeros of
            # x's padding columns itself (a strided fill of those columns costs more than the whole GEMM)
            wt = hip.transpose(weight, rows_out=x.shape[1])
            lk = ctx.fm_link
            if lk is not None and lk.D == 64 and weight.shape[0] == 64 and hip.get_matmul_precision() != "fp32" \
                    and hip.embed_grad_gemm_fits(lk.D, weight.shape[0], dpre, wt):
                # the consumer of this gradient is the embedding gather: its backward forms the dX rows it needs from
                # (dH, W^T) on the matrix core, inside the segmented reduce.  What flows back through autograd is a
                # stride-0 zero of the right shape (other consumers of x add their gradients to it as usual).
                lk.dgrad = (dpre, wt)
                # (the library's own fill: an ATen one would keep a recorded step from replaying as a launch plan)
                dx = hip.zeros((1, 1), x.dtype, x.device).expand(x.shape[0], x.shape[1])
                if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                    dw, db = hip.linear_wgrad(dpre, x, ctx.K, want_bias=ctx.has_bias)
                return dx, dw, db, None, None, None, None
            dx = torch.empty_like(x)
            if lk is not None and lk.dfm is not None and lk.ssum is not None and hip.linear_fwd_rowadd(
                    dpre, wt, lk.dfm.reshape(-1).contiguous(), lk.ssum, lk.ncols, dx):
                lk.folded = True  # the gather backward now only applies the -g_fm * v part
            elif ctx.in_link is not None:
                hip.linear_fwd(dpre, wt, None, ACT_MASK, aux=x, out=dx)  # x = relu(...) of the producer: its mask
                ctx.in_link.dx = dx
            else:
                hip.linear_fwd(dpre, wt, None, ACT_NONE, out=dx)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = hip.linear_wgrad(dpre, x, ctx.K, want_bias=ctx.has_bias)
        return dx, dw, db, None, None, None, None


class _Activation(torch.autograd.Function):
    """an activation module as a launch of its own (rp_act_fwd / rp_act_bwd), for the places where it does not follow a Linear"""

    @staticmethod
    def forward(ctx, x, act: int):
        y = hip.act_fwd(_unit_inner(x), act)
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return hip.act_bwd(_unit_inner(dy), y, ctx.act), None


def activation(x, act: int):
    return _Activation.apply(x, act)


def linear_act(x, weight, bias=None, act: int = ACT_NONE, fm_link=None, in_link=None, out_link=None):
    """act(x[:, :K] @ weight^T + bias) for 2-D x; x may carry zero padding columns beyond K.
    in_link / out_link: ReluLink shared with the producing / consuming layer (see ReluLink)."""
    lead = None
    if x.dim() != 2:
        lead = x.shape[:-1]
        x = x.reshape(-1, x.shape[-1])
        in_link = None
    y = _LinearAct.apply(x, weight, bias, act, fm_link, in_link, out_link if lead is None else None)
    return y if lead is None else y.reshape(*lead, y.shape[-1])


class _MLPTail64(torch.autograd.Function):
    """hin [B,64] (output of a Linear+ReLU) -> [Linear 64x64 + ReLU] x L -> Linear 64 -> 1, ONE launch each way
    (rp_mlp_tail_fwd / rp_mlp_tail_bwd).  `in_link`: the ReluLink of the producing Linear+ReLU — the gradient that
    comes out is already masked by hin > 0, so that layer skips its relu_bwd pass."""

    @staticmethod
    def forward(ctx, hin, in_link, w_out, b_out, *wb):
        hin = _unit_inner(hin)
        L = len(wb) // 2
        Ws = [w.contiguous() for w in wb[:L]]
        bs = list(wb[L:])
        logit, hs = hip.mlp_tail_fwd(hin, Ws, bs, w_out.contiguous(), b_out)
        ctx.L, ctx.in_link, ctx.has_bias = L, in_link, [b is not None for b in bs] + [b_out is not None]
        ctx.save_for_backward(hin, w_out, *Ws, *hs)
        return logit

    @staticmethod
    def backward(ctx, dz):
        L = ctx.L
        hin, w_out, *rest = ctx.saved_tensors
        Ws, hs = rest[:L], rest[L:]
        dhin, dWs, dbs, dw_out, db_out = hip.mlp_tail_bwd(dz.reshape(-1).contiguous(), list(Ws), [hin] + list(hs),
                                                           w_out.contiguous())
        if ctx.in_link is not None:
            ctx.in_link.dx = dhin
        dbs = [d if hb else None for d, hb in zip(dbs, ctx.has_bias[:L])]
        return (dhin, None, dw_out.view_as(w_out), db_out if ctx.has_bias[L] else None, *dWs, *dbs)


class _MLPTail64BCE(torch.autograd.Function):
    """_MLPTail64 with the model's loss head inside both launches (rp_mlp_tail_fwd_bce / rp_mlp_tail_bwd_bce):
    forward(hin, in_link, label, p_eps, weight, n_add, w_out, b_out, *addends, *Ws, *bs) -> (pred [B,1], loss []).
    pred and every gradient are bit-identical to mlp_tail64 + sigmoid_bce (same arithmetic per row); the loss scalar is the
    same sum in another order.  A DeepFM step loses three launches of its critical path (sigmoid_bce_fwd, loss_finish,
    sigmoid_bce_bwd: 18 us of 0.87 ms at B = 65536)."""

    @staticmethod
    def forward(ctx, hin, in_link, label, p_eps: float, weight: float, n_add: int, w_out, b_out, *rest):
        ctx.set_materialize_grads(False)  # `pred` is normally not differentiated
        hin = _unit_inner(hin)
        addends, wb = rest[:n_add], rest[n_add:]
        L = len(wb) // 2
        Ws = [w.contiguous() for w in wb[:L]]
        bs = list(wb[L:])
        label = label.contiguous()
        pred, loss, hs = hip.mlp_tail_fwd_bce(hin, Ws, bs, w_out.contiguous(), b_out, [a.contiguous() for a in addends], label,
                                              p_eps, weight)
        ctx.cfg = (L, in_link, [b is not None for b in bs] + [b_out is not None], p_eps, weight, [tuple(a.shape) for a in addends])
        ctx.save_for_backward(hin, w_out, pred, label, *Ws, *hs)
        return pred, loss

    @staticmethod
    def backward(ctx, dpred, dloss):
        L, in_link, has_bias, p_eps, weight, shapes = ctx.cfg
        hin, w_out, pred, label, *rest = ctx.saved_tensors
        Ws, hs = rest[:L], rest[L:]
        n_add = len(shapes)
        if dloss is None and dpred is None:
            return (None,) * (8 + n_add + 2 * L)
        if dpred is not None or dloss is None:  # someone differentiated through `pred` itself (rare): the separate launches
            dz = None
            if dloss is not None:
                dz = hip.sigmoid_bce_bwd(pred, label, dloss, True, p_eps, weight)
            if dpred is not None:
                extra = dpred * (pred * (1 - pred))
                dz = extra if dz is None else dz + extra
            dz = dz.reshape(-1).contiguous()
            dhin, dWs, dbs, dw_out, db_out = hip.mlp_tail_bwd(dz, list(Ws), [hin] + list(hs), w_out.contiguous())
        else:
            dz = torch.empty((pred.shape[0],), dtype=torch.float32, device=pred.device) if n_add else None
            dhin, dWs, dbs, dw_out, db_out = hip.mlp_tail_bwd(None, list(Ws), [hin] + list(hs), w_out.contiguous(),
                                                               bce=(pred, label, dloss.contiguous(), p_eps, weight, dz))
        if in_link is not None:
            in_link.dx = dhin
        dbs = [d if hb else None for d, hb in zip(dbs, has_bias[:L])]
        dadds = [dz.reshape(sh) for sh in shapes]
        return (dhin, None, None, None, None, None, dw_out.view_as(w_out), db_out if has_bias[L] else None, *dadds, *dWs, *dbs)


def mlp_tail64_bce(hin, in_link, hidden, head, addends, label, p_eps: float = 0.0, weight: float = 1.0):
    """(pred [B,1], loss) = BCE(sigmoid(sum(addends) + tail(hin)), label): mlp_tail64 and sigmoid_bce as one launch each way"""
    Ws = [w for w, _ in hidden]
    bs = [b for _, b in hidden]
    return _MLPTail64BCE.apply(hin, in_link, label, float(p_eps), float(weight), len(addends), head[0], head[1], *addends, *Ws, *bs)


def mlp_tail64(hin, in_link, hidden, head):
    """hidden: [(weight [64,64], bias or None), ...] (1..3 layers, each followed by ReLU); head: (weight [1,64], bias)."""
    Ws = [w for w, _ in hidden]
    bs = [b for _, b in hidden]
    return _MLPTail64.apply(hin, in_link, head[0], head[1], *Ws, *bs)


# ----------------------------------------------------------------------------------------------
# K5  CrossNet (+ the fc that follows it in DCN)   — layers/interaction.py:119-141, ranking/dcn.py:64
# ----------------------------------------------------------------------------------------------
class _CrossNet(torch.autograd.Function):
    """forward(x0, wfc, bfc, L, w_0 .. w_{L-1}, b_0 .. b_{L-1}): the layers' parameters arrive one by one and are stacked by ONE
    library launch (rp_multi_copy) — torch.stack was two ATen cat launches per step —, their gradients leave as rows of the
    [L, d] results of rp_crossnet_param_grads.  A DCN step then holds library launches only (graph_step: launch plan)."""

    @staticmethod
    def forward(ctx, x0, wfc, bfc, L: int, *wb):
        x0 = _unit_inner(x0)
        d = wb[0].numel()
        stk = torch.empty((2 * L, d), dtype=torch.float32, device=x0.device)
        src = [t.detach().reshape(-1) for t in wb]
        if not hip.multi_copy([stk[i] for i in range(2 * L)], src):
            torch._foreach_copy_([stk[i] for i in range(2 * L)], src)
        W, Bv = stk[:L], stk[L:]
        wfc_c = None if wfc is None else wfc.contiguous()
        xout, logit, s = hip.crossnet_fwd(x0, d, W, Bv, wfc_c, bfc, want_x=wfc is None)
        ctx.d, ctx.L, ctx.fused_fc = d, L, wfc is not None
        ctx.shapes = [t.shape for t in wb]
        ctx.save_for_backward(x0, W, Bv, wfc_c, s)
        return logit if wfc is not None else xout

    @staticmethod
    def backward(ctx, g):
        """Streaming backward (rp_crossnet_bwd_rows): X_l = A_l X_0 + C_l, so the per-row kernel only emits dX_0 and
        2L+2 scalars per sample; the parameter gradients are one skinny wgrad GEMM V^T X_0 plus [L,d] arithmetic
        (rp_crossnet_param_grads)."""
        x0, W, Bv, wfc, s = ctx.saved_tensors
        g = g.contiguous()
        L, d = ctx.L, ctx.d
        fused = ctx.fused_fc
        dx0, V = hip.crossnet_bwd_rows(x0, d, W, wfc if fused else None, s, None if fused else g, g if fused else None)
        P, cs = hip.linear_wgrad(V, x0, d)                      # [2L+2, d], column sums of V
        colg = None
        if not fused:
            _, colg = hip.linear_wgrad(g, g, 1)                 # column sums of the incoming gradient
        dW, dB, dwfc = hip.crossnet_param_grads(P, cs, W, Bv, wfc.reshape(-1) if fused else None, colg)
        grads = [dW[i].view(ctx.shapes[i]) for i in range(L)] + [dB[i].view(ctx.shapes[L + i]) for i in range(L)]
        if fused:
            return (dx0, dwfc.view_as(wfc), cs[2 * L + 1:2 * L + 2], None) + tuple(grads)
        return (dx0, None, None, None) + tuple(grads)


def crossnet(x0, layer_w, layer_b, wfc=None, bfc=None):
    """x0 [B, >=d] -> X_L [B, d], or the fc logit [B,1] when (wfc [1,d], bfc [1]) are given.  layer_w / layer_b: the L
    layers' weight / bias parameters."""
    L = len(layer_w)
    return _CrossNet.apply(x0, wfc, bfc, L, *layer_w, *layer_b)


# ----------------------------------------------------------------------------------------------
# K6  xDeepFM CIN layer   — layers/interaction.py:164-168
# ----------------------------------------------------------------------------------------------
class CINLink:
    """Hand-off between the CIN's first layer and its collapsed last layer (both read X_0): the last layer's backward — which
    runs first — parks its gradient of X_0 here and the first layer's backward adds it into its own with one library launch
    (rp_add_rows), instead of autograd's ATen sum of the two [B, H D] gradients (a recorded step must hold none)."""
    __slots__ = ("dx0",)

    def __init__(self):
        self.dx0 = None


class _CINLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, xp, W, bias, H: int, M: int, D: int, want_out: bool, link=None):
        x0 = _unit_inner(x0)
        same = xp is None
        xp_t = x0 if same else _unit_inner(xp)
        W = W.contiguous()
        O = W.shape[0]
        # first layers (X_{k-1} = X_0, <= 32 fields) run on the bf16 matrix core (rp_cin_bs_*)
        bs = same and hip.get_matmul_precision() != "fp32" and hip.cin_bs_fits(H, M, D)
        wst = None
        if bs and hip.cin_pair_fits(H, O, D):  # one GEMM over the H(H+1)/2 pair products
            # (both layouts of the pair weights' bf16 pieces from one launch: the backward's is kept — the weights do not
            #  change between this forward and its backward)
            wsp, wst = hip.cin_pair_pieces(W.view(O, H, M), both=True)
            out, pooled = hip.cin_pair_fwd(x0, wsp, bias, H, O, D, want_out, True)
        elif bs:
            out, pooled = hip.cin_bs_fwd(x0, xp_t, hip.bf16_pieces(W.view(O, H, M)), bias, H, M, O, D, want_out, True)
        else:
            out, pooled = hip.cin_layer_fwd(x0, xp_t, W, bias, H, M, D, want_out, True)
        ctx.cfg = (H, M, D, same, bias is not None, want_out, bs)
        ctx.link = link
        ctx.save_for_backward(x0, None if same else xp_t, W, wst)
        if want_out:
            return out, pooled
        return pooled

    @staticmethod
    def backward(ctx, *grads):
        x0, xp, W, wst = ctx.saved_tensors
        H, M, D, same, has_bias, want_out, bs = ctx.cfg
        g_out, g_pool = (grads if want_out else (None, grads[0]))
        g_out = None if g_out is None else g_out.contiguous()
        g_pool = None if g_pool is None else _unit_inner(g_pool)
        extra = None
        if ctx.link is not None and ctx.link.dx0 is not None:
            extra, ctx.link.dx0 = ctx.link.dx0, None
        if bs:
            O = W.shape[0]
            W3 = W.view(O, H, M)
            # X_0 enters in both roles: one pass with W[o,h,m] + W[o,m,h] gives its whole gradient
            gp = None if g_pool is None else g_pool.contiguous()  # [B, O] packed (it arrives as a slice of the cat)
            if hip.cin_pair_fits(H, O, D) and (g_out is None or g_out.data_ptr() % 16 == 0):
                # (the last layer's gradient of X_0, parked in the link: this launch adds its own into it)
                dx0 = hip.cin_pair_bwd_x(x0, wst if wst is not None else hip.cin_pair_pieces(W3, transposed=True), g_out, gp,
                                         H, O, D, like=x0, into=extra)
                extra = None
            else:
                dx0 = hip.cin_bs_bwd_x(x0, hip.bf16_pieces(W3 + W3.transpose(1, 2)), g_out, gp, H, M, O, D, like=x0)
            if extra is not None:
                hip.add_rows_to(extra, dx0[:, :extra.shape[1]])
            if x0.stride(0) % 4 == 0 and x0.data_ptr() % 16 == 0:
                if O <= 128:  # symmetric pair form: products formed once, 2.9x fewer matrix-core passes
                    dW, db = hip.cin_pair_bwd_w(x0, g_out, gp, H, O, D, has_bias)
                else:
                    dW, db = hip.cin_bs_bwd_w(x0, x0, g_out, gp, H, M, O, D, has_bias)
                dW = dW.view_as(W)
            else:
                dW, db = hip.cin_layer_bwd_w(x0, x0, W, H, M, D, g_out, g_pool, has_bias)
            return dx0, None, dW, db, None, None, None, None, None
        dx0, dxp, dW, db = hip.cin_layer_bwd(x0, x0 if same else xp, W, H, M, D, g_out, g_pool, has_bias)
        if extra is not None:
            hip.add_rows_to(extra, dx0[:, :extra.shape[1]])
        return dx0, dxp, dW, db, None, None, None, None, None


class _CINLast(torch.autograd.Function):
    """The collapsed last CIN layer: p[b] = sum_d sum_{h,m} V[h,m] X_0[b,h,d] X_{L-1}[b,m,d]  (rp_cin_last_*)."""

    @staticmethod
    def forward(ctx, x0, xp, V, H: int, M: int, D: int):
        x0 = _unit_inner(x0)
        same = xp is None
        xp_t = x0 if same else _unit_inner(xp)
        vt = torch.zeros((M, 32), dtype=torch.float32, device=x0.device)
        vt[:, :H] = V.reshape(H, M).t()
        ctx.cfg = (H, M, D, same)
        ctx.save_for_backward(x0, None if same else xp_t, vt)
        return hip.cin_last_fwd(x0, xp_t, vt, H, M, D)

    @staticmethod
    def backward(ctx, gp):
        x0, xp, vt = ctx.saved_tensors
        H, M, D, same = ctx.cfg
        dx0, dxp, dV = hip.cin_last_bwd(x0, x0 if same else xp, vt, gp.reshape(-1).contiguous(), H, M, D)
        if same:  # first layer == last layer: both roles are X_0 (M == H, same row layout)
            dx0 = dx0 + dxp
            dxp = None
        return dx0, dxp, dV.reshape(1, H * M), None, None, None


def cin_last(x0, xp, V, H: int, M: int, D: int):
    """x0 [B, >=H*D], xp [B, M*D] or None (X_{L-1} = X_0), V [1, H*M] -> [B, 1]."""
    return _CINLast.apply(x0, xp, V, H, M, D)


class _CINChunked(torch.autograd.Function):
    """A CIN layer whose second factor X_{k-1} is NOT X_0 (a middle layer), fed by any number of maps, on the bf16 matrix
    core: the sum over chunks of <= 32 maps of rp_cin_bs layers (interaction.py:164-168; include/rec_pangu_hip.h K6).
    X_k[b,o,:] = sum_c sum_{h, m in c} W[o,h,m] X_0[b,h,:] X_{k-1}[b,m,:] + bias[o]."""
    CHUNK = 32

    @staticmethod
    def forward(ctx, x0, xp, W, bias, H: int, M: int, D: int):
        x0, xp = _unit_inner(x0), _unit_inner(xp)
        O = W.shape[0]
        W3 = W.reshape(O, H, M)
        out = pooled = None
        for c0 in range(0, M, _CINChunked.CHUNK):
            c1 = min(M, c0 + _CINChunked.CHUNK)
            o_c, p_c = hip.cin_bs_fwd(x0, xp[:, c0 * D:c1 * D], hip.bf16_pieces(W3[:, :, c0:c1]), bias if c0 == 0 else None,
                                      H, c1 - c0, O, D, True, True)
            if out is None:
                out, pooled = o_c, p_c
            else:
                hip.accumulate(out, o_c)
                hip.accumulate(pooled, p_c)
        ctx.cfg = (H, M, D, bias is not None)
        ctx.save_for_backward(x0, xp, W3)
        return out, pooled

    @staticmethod
    def backward(ctx, g_out, g_pool):
        x0, xp, W3 = ctx.saved_tensors
        H, M, D, has_bias = ctx.cfg
        O = W3.shape[0]
        g_out = None if g_out is None else g_out.contiguous()
        gp = None if g_pool is None else g_pool.contiguous()  # packed [B, O]
        dx0 = None
        dxp = torch.empty((x0.shape[0], M * D), dtype=torch.float32, device=x0.device)
        dW = torch.empty((O, H, M), dtype=torch.float32, device=x0.device)
        db = None
        for c0 in range(0, M, _CINChunked.CHUNK):
            c1 = min(M, c0 + _CINChunked.CHUNK)
            mc = c1 - c0
            xc, Wc = xp[:, c0 * D:c1 * D], W3[:, :, c0:c1]
            # X_0-role gradient: rows h, contraction over the chunk's maps; X_{k-1}-role: rows m, contraction over h
            d0 = hip.cin_bs_bwd_x(xc, hip.bf16_pieces(Wc), g_out, gp, H, mc, O, D, like=x0)
            dx0 = d0 if dx0 is None else hip.accumulate(dx0, d0)
            hip.cin_bs_bwd_x(x0, hip.bf16_pieces(Wc.transpose(1, 2)), g_out, gp, mc, H, O, D, like=None,
                             out=dxp[:, c0 * D:c1 * D])
            dWc, dbc = hip.cin_bs_bwd_w(x0, xc, g_out, gp, H, mc, O, D, has_bias and c0 == 0)
            dW[:, :, c0:c1] = dWc.view(O, H, mc)
            if dbc is not None:
                db = dbc
        return dx0, dxp, dW.view(O, H * M), db, None, None, None


def cin_middle_fits(H: int, D: int, x0, xp) -> bool:
    """can a middle CIN layer (any number of input maps) run on the chunked bf16 matrix-core form?"""
    return (hip.get_matmul_precision() != "fp32" and hip.cin_bs_fits(H, min(32, H), D) and x0.stride(0) % 4 == 0
            and x0.data_ptr() % 16 == 0 and xp.data_ptr() % 16 == 0)


def cin_middle(x0, xp, W, bias, H: int, M: int, D: int):
    """-> (X_k [B, O, D], pooled [B, O]) of a middle CIN layer on the bf16 matrix core (see _CINChunked)"""
    return _CINChunked.apply(x0, xp, W, bias, H, M, D)


def cin_layer(x0, xp, W, bias, H: int, M: int, D: int, want_out: bool = True, link=None):
    """(X_k [B, O, D], pooled [B, O]) or pooled alone; link: a CINLink shared with cin_head (the collapsed last layer's
    gradient of X_0 is added into this layer's inside its backward)"""
    return _CINLayer.apply(x0, xp, W, bias, H, M, D, want_out, link)


class _CINHead(torch.autograd.Function):
    """The CIN's collapsed LAST layer with its weight-space arithmetic inside (round 6; models/layers/interaction.py has the
    algebra):  logit[b] = sum_d sum_{h,m} V[h,m] X_0[b,h,d] X_{L-1}[b,m,d] + D (c . b_L) + fc.bias,  V = c . W_L.
    One launch for (V^T, c . b_L), rp_cin_last_fwd, one launch for the two scalars; the backward's dW_L / db_L / dc from one
    launch.  What this replaces were two matmuls, four elementwise launches and their autograd counterparts per step."""

    @staticmethod
    def forward(ctx, x0, xp, WL, bL, c, fcb, H: int, M: int, D: int, link):
        x0 = _unit_inner(x0)
        xp = _unit_inner(xp)
        WL = WL.contiguous()
        c = c.reshape(-1).contiguous()
        vt, vb = hip.cin_head_params_fwd(WL, bL, c, H, M)
        out = hip.cin_last_fwd(x0, xp, vt, H, M, D)
        hip.add_scalars(out, vb, float(D), fcb)
        ctx.cfg, ctx.link = (H, M, D, bL is not None, fcb is not None, c.shape), link
        ctx.save_for_backward(x0, xp, vt, WL, bL, c)
        return out

    @staticmethod
    def backward(ctx, g):
        x0, xp, vt, WL, bL, c = ctx.saved_tensors
        H, M, D, has_b, has_fcb, cshape = ctx.cfg
        g = g.reshape(-1).contiguous()
        dx0, dxp, dV = hip.cin_last_bwd(x0, xp, vt, g, H, M, D)
        sg = hip.sum_all(g)
        dWL, dbL, dc = hip.cin_head_params_bwd(WL, bL, c, dV, sg, D, H, M)
        if ctx.link is not None:
            ctx.link.dx0, dx0 = dx0, None  # (added into the first layer's gradient of X_0 by its backward: CINLink)
        return dx0, dxp, dWL, dbL, dc.view(1, -1), (sg if has_fcb else None), None, None, None, None


def cin_head(x0, xp, WL, bL, c, fcb, H: int, M: int, D: int, link=None):
    """x0 [B, >= H D], xp = X_{L-1} [B, M D], WL [O, H M], bL [O] or None, c [1, O] (fc.weight's slice for the last layer's
    pooling), fcb [1] or None -> [B, 1]"""
    return _CINHead.apply(x0, xp, WL, bL, c, fcb, H, M, D, link)


class _RowSplit(torch.autograd.Function):
    """w [1, n] -> (w[:, :k], w[:, k:]) as tensors of their own that alias w's memory (no launch); the backward writes the two
    gradients into the halves of one [1, n] buffer with one library launch.  (Slicing a parameter leaves autograd a slice node
    per half: a zero fill and a copy each in the backward — ATen launches a recorded step must not hold.)"""

    @staticmethod
    def forward(ctx, w, k: int):
        ctx.k, ctx.n = k, w.shape[1]
        d = w.detach()
        return d[:, :k], d[:, k:]

    @staticmethod
    def backward(ctx, g1, g2):
        k, n = ctx.k, ctx.n
        ref = g1 if g1 is not None else g2
        g = torch.empty((1, n), dtype=torch.float32, device=ref.device)
        parts = []
        for gi, (a, b) in ((g1, (0, k)), (g2, (k, n))):
            parts.append((g[:, a:b], gi if gi is not None else hip.zeros((1, b - a), torch.float32, ref.device)))
        if not hip.multi_copy([p[0] for p in parts], [p[1].contiguous() for p in parts]):
            for dst, src in parts:
                hip.copy_rows_to(src, dst)
        return g, None


def row_split(w, k: int):
    return _RowSplit.apply(w, k)


class _TokenAlias(torch.autograd.Function):
    """_TokenView without the copy: the first F*D columns of x as a tensor of its own that aliases x's memory, for consumers that
    take a row stride (the CIN kernels).  The gradient takes the same road: parked in the gather's link, added into the other
    consumer's dX by the gather's backward (rp_add_rows)."""

    @staticmethod
    def forward(ctx, x, n: int, link):
        ctx.link = link
        return _unit_inner(x).detach()[:, :n]

    @staticmethod
    def backward(ctx, d):
        ctx.link.extra = d if d.is_contiguous() else d.contiguous()
        return None, None, None


def token_alias(x, n: int, link):
    return _TokenAlias.apply(x, n, link)



# ----------------------------------------------------------------------------------------------
# K7  AutoInt field self-attention layer   — layers/attention.py:63-101
# ----------------------------------------------------------------------------------------------
class _FieldAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, T: int, Din: int, H: int, a: int, has_res: bool, scale: float):
        x = _unit_inner(x)
        W = W.contiguous()
        out = hip.field_attention_fwd(x, W, T, Din, H, a, has_res, scale)
        ctx.cfg = (T, Din, H, a, has_res, scale)
        ctx.save_for_backward(x, W)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, W = ctx.saved_tensors
        T, Din, H, a, has_res, scale = ctx.cfg
        dx, dW = hip.field_attention_bwd(x, W, T, Din, H, a, has_res, scale, gout.contiguous(),
                                         ctx.needs_input_grad[0])
        return dx, dW, None, None, None, None, None, None


class _AttentionCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkvr, xres, T: int, H: int, a: int, scale: float):
        qkvr = qkvr.contiguous()
        nproj = qkvr.shape[1] // (H * a)
        xr = None if nproj == 4 else _unit_inner(xres)
        out, stats = hip.attention_core_fwd(qkvr, nproj, xr, T, H, a, scale)
        ctx.cfg = (nproj, T, H, a, scale)
        ctx.save_for_backward(qkvr, out, stats)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkvr, out, stats = ctx.saved_tensors
        nproj, T, H, a, scale = ctx.cfg
        dqkvr, dxres = hip.attention_core_bwd(qkvr, nproj, out, dout.contiguous(), stats, T, H, a, scale)
        return dqkvr, dxres, None, None, None, None


def field_attention_split(X, W, T: int, Din: int, H: int, a: int, has_res: bool, scale: float = 0.0):
    """Same layer as field_attention, split: ONE matrix-core GEMM for all projections of all B*T tokens
    (rp_linear_fwd; its dgrad / wgrad give dX and the weight gradients), then the T x T attention core per sample
    (rp_attention_core_*).  X [B, T, Din] -> [B, T, H*a]."""
    B = X.shape[0]
    x2 = X.reshape(B * T, Din)
    qkvr = linear_act(x2, W, None, ACT_NONE)                       # [B*T, (3|4)*H*a]
    out = _AttentionCore.apply(qkvr, None if has_res else x2, T, H, a, scale)
    return out.view(B, T, H * a)


def field_attention(x, W, T: int, Din: int, H: int, a: int, has_res: bool, scale: float = 0.0):
    """x [B, >=T*Din] -> relu(attention + residual) [B, T, H*a];  W = cat(Wq, Wk, Wv[, Wres]) [(3|4)*H*a, Din]."""
    return _FieldAttention.apply(x, W, T, Din, H, a, has_res, scale)


# ----------------------------------------------------------------------------------------------
# K8  MMOE: one GEMM over [experts | gates] stored input-major ([h, N], as the reference keeps them),
#     then gate softmax + gate-weighted combine   — multi_task/mmoe.py:86-104
# ----------------------------------------------------------------------------------------------
class _LinearInputMajor(torch.autograd.Function):
    """z = x[:, :h] @ Wm + bias with Wm [h, N] (einsum 'ij,jk->ik' layout)."""

    @staticmethod
    def forward(ctx, x, Wm, bias):
        x = _unit_inner(x)
        Wm = Wm.contiguous()
        h = Wm.shape[0]
        z = hip.linear_fwd(x, hip.transpose(Wm), bias, ACT_NONE, K=h)
        ctx.h = h
        ctx.save_for_backward(x, Wm)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, Wm = ctx.saved_tensors
        dz = _unit_inner(dz)
        h, N = Wm.shape
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if x.shape[1] > h:  # zero rows appended to Wm: the GEMM writes the padding columns' zeros itself
                Wp = torch.empty((x.shape[1], Wm.shape[1]), dtype=Wm.dtype, device=Wm.device)
                Wp[:h].copy_(Wm)
                Wp[h:].zero_()
            else:
                Wp = Wm
            hip.linear_fwd(dz, Wp, None, ACT_NONE, out=dx)  # dz @ Wm^T
        if ctx.needs_input_grad[1]:
            dW, _ = hip.linear_wgrad(x[:, :h], dz, N, want_bias=False)  # roles swapped: x^T @ dz -> [h, N]
        if ctx.needs_input_grad[2]:
            # column sums of dz: the deterministic two-stage column-sum kernel (they rode on the wgrad kernel with K = 1 until
            # round 5: 0.137 ms at [65536, 520], 8x the time of streaming dz once)
            db = hip.batchnorm_colsum(dz)
        return dx, dW, db


class _MMOEProject(torch.autograd.Function):
    """z = x[:, :h] @ [experts | gate_1 | .. | gate_T] + [experts_bias | gate biases]  (multi_task/mmoe.py:86-104 as ONE GEMM)
    straight from the model's parameters: forward(x, pack, T, experts [h,K,E], experts_bias, *gates [h,E], *gate_biases [E]).
    `pack` [rows of x, N] is a persistent buffer of the model whose rows beyond h are zero (the dgrad's weight operand: it
    writes exact zeros into x's padding columns): the parameters are copied into their column blocks by rp_copy_rows /
    rp_multi_copy and their gradients leave as contiguous blocks the same way — torch.cat (x2), the padded copy and its
    zero fill, and autograd's strided-gradient clones were 6+ ATen launches per step."""

    @staticmethod
    def forward(ctx, x, pack, T: int, experts, experts_bias, *gw):
        x = _unit_inner(x)
        h, K, E = experts.shape
        gates, gbias = gw[:T], gw[T:]
        N = K * E + T * E
        hip.copy_rows_to(experts.detach().reshape(h, K * E), pack[:h, :K * E])
        for t in range(T):
            hip.copy_rows_to(gates[t].detach(), pack[:h, K * E + t * E:K * E + (t + 1) * E])
        bias = torch.empty((N,), dtype=torch.float32, device=x.device)
        dst = [bias[:K * E]] + [bias[K * E + t * E:K * E + (t + 1) * E] for t in range(T)]
        src = [experts_bias.detach().reshape(-1)] + [b.detach() for b in gbias]
        if not hip.multi_copy(dst, src):
            torch._foreach_copy_(dst, src)
        Wm = pack[:h]
        z = hip.linear_fwd(x, hip.transpose(Wm), bias, ACT_NONE, K=h)
        ctx.cfg = (h, K, E, T)
        ctx.save_for_backward(x, pack)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, pack = ctx.saved_tensors
        h, K, E, T = ctx.cfg
        dz = _unit_inner(dz)
        N = K * E + T * E
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            hip.linear_fwd(dz, pack, None, ACT_NONE, out=dx)  # dz @ Wm^T (the zero rows of pack give the padding columns of dx)
        need = ctx.needs_input_grad  # (x, pack, T, experts, experts_bias, gates.., gate biases..: the reference's gates are
        #                               plain tensors, not parameters — buffers here — and take no gradient)
        dexp, dgates = None, [None] * T
        if need[3] or any(need[5:5 + T]):
            dW, _ = hip.linear_wgrad(x[:, :h], dz, N, want_bias=False)  # roles swapped: x^T @ dz -> [h, N]
            if need[3]:
                dexp = hip.copy_rows_to(dW[:, :K * E], torch.empty((h, K * E), dtype=torch.float32, device=x.device)).view(h, K, E)
            for t in range(T):
                if need[5 + t]:
                    dgates[t] = hip.copy_rows_to(dW[:, K * E + t * E:K * E + (t + 1) * E],
                                                 torch.empty((h, E), dtype=torch.float32, device=x.device))
        dbe, dbias = None, [None] * T
        if need[4] or any(need[5 + T:5 + 2 * T]):
            db = hip.batchnorm_colsum(dz)
            dbe = db[:K * E].view(K, E) if need[4] else None
            dbias = [db[K * E + t * E:K * E + (t + 1) * E] if need[5 + T + t] else None for t in range(T)]
        return (dx, None, None, dexp, dbe, *dgates, *dbias)


def mmoe_project(x, pack, experts, experts_bias, gates, gate_biases):
    return _MMOEProject.apply(x, pack, len(gates), experts, experts_bias, *gates, *gate_biases)


def linear_input_major(x, Wm, bias):
    return _LinearInputMajor.apply(x, Wm, bias)


class _MMOECombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, K: int, E: int, T: int):
        z = _unit_inner(z)
        out, gate = hip.mmoe_combine_fwd(z, K, E, T)
        ctx.cfg = (K, E, T)
        ctx.save_for_backward(z, gate)
        ctx.set_materialize_grads(False)
        # T outputs, one per task (the rows of one [T, B, K] buffer): handed out as ONE tensor, the towers' slices came back
        # through autograd as T zero-filled [T, B, K] gradients plus their sum — 2T + 1 ATen launches over 67 MB each
        return tuple(out[t] for t in range(T))

    @staticmethod
    def backward(ctx, *douts):
        z, gate = ctx.saved_tensors
        K, E, T = ctx.cfg
        dout = torch.empty((T, z.shape[0], K), dtype=torch.float32, device=z.device)
        have = [t for t in range(T) if douts[t] is not None]
        for t in range(T):
            if douts[t] is None:
                dout[t].zero_()  # (a task whose mixture nobody used)
        if have and not hip.multi_copy([dout[t] for t in have], [douts[t].contiguous() for t in have]):
            torch._foreach_copy_([dout[t] for t in have], [douts[t] for t in have])
        dz = hip.mmoe_combine_bwd(z, K, E, T, gate, dout)
        if z.shape[1] != dz.shape[1]:
            full = torch.zeros_like(z)
            full[:, :dz.shape[1]] = dz
            dz = full
        return dz, None, None, None


def mmoe_combine(z, K: int, E: int, T: int):
    """z [B, K*E + T*E] (experts | gate logits) -> T gate-weighted expert mixtures [B, K] (the rows of one [T, B, K] buffer)."""
    return _MMOECombine.apply(z, K, E, T)


# ----------------------------------------------------------------------------------------------
# stand-alone FM pooling ([B,F,D] already materialised)   — layers/interaction.py:36-44
# ----------------------------------------------------------------------------------------------
class _FMPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, F: int, D: int, bi: bool):
        x2d = _unit_inner(x2d)
        out_sum, out_bi = hip.fm_pool_fwd(x2d, F, D, not bi, bi)
        ctx.cfg = (F, D, bi)
        ctx.save_for_backward(x2d)
        return out_bi if bi else out_sum

    @staticmethod
    def backward(ctx, g):
        (x2d,) = ctx.saved_tensors
        F, D, bi = ctx.cfg
        g = g.contiguous()
        return hip.fm_pool_bwd(x2d, F, D, None if bi else g, g if bi else None), None, None, None


def fm_pool(feature_emb, bi_interaction: bool = False):
    """feature_emb [B,F,D] (any row stride) -> product_sum_pooling [B,1] or Bi_interaction_pooling [B,D]."""
    B, F, D = feature_emb.shape
    return _FMPool.apply(feature_emb.reshape(B, F * D), F, D, bi_interaction)


# ----------------------------------------------------------------------------------------------
# K9  BatchNorm1d (MMOE towers)   — multi_task/mmoe.py:54
# ----------------------------------------------------------------------------------------------
class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float):
        x = _unit_inner(x)
        y, mean, var, rstd = hip.batchnorm_train_fwd(x, gamma, beta, eps)
        ctx.save_for_backward(x, mean, rstd, gamma)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)  # (the statistics' gradients were materialised as zero fills: 2 ATen launches per BN)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _gm, _gv):
        x, mean, rstd, gamma = ctx.saved_tensors
        if dy is None:
            return None, None, None, None
        dx, dgamma, dbeta = hip.batchnorm_train_bwd(x, _unit_inner(dy), mean, rstd, gamma)
        # (affine=False — Dice's BatchNorm — has neither: a gradient for a None input is an autograd error)
        return dx, (dgamma if gamma is not None else None), (dbeta if ctx.needs_input_grad[2] else None), None


class _BatchNormApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mean, rstd, gamma, beta):
        x = _unit_inner(x)
        ctx.save_for_backward(x, mean, rstd, gamma)
        return hip.batchnorm_apply(x, mean, rstd, gamma, beta)

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma = ctx.saved_tensors
        dy = _unit_inner(dy)
        dx = hip.batchnorm_apply_bwd(dy, rstd, gamma)
        dgamma = dbeta = None
        if gamma is not None and ctx.needs_input_grad[3]:
            dgamma = (dy * ((x - mean) * rstd)).sum(dim=0)  # eval-mode fine-tuning is off the hot path
        if ctx.needs_input_grad[4]:
            dbeta = dy.sum(dim=0)
        return dx, None, None, dgamma, dbeta


def batch_norm(x, bn: torch.nn.BatchNorm1d):
    """nn.BatchNorm1d semantics on the HIP kernels, including the running-statistics update in training mode."""
    use_batch = bn.training or not bn.track_running_stats or bn.running_mean is None
    if use_batch and type(bn) is not torch.nn.BatchNorm1d:
        return bn(x)  # sharded.SyncBatchNorm1d: global-batch statistics (torch ops + one all-reduce each way)
    if use_batch:
        y, mean, var = _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.eps)
        if bn.training and bn.track_running_stats and bn.running_mean is not None:
            # one launch (six ATen launches per BatchNorm until round 5: 24 per MMOE step); in place, like nn.BatchNorm1d
            hip.batchnorm_update_running(mean, var, bn, x.shape[0])
        return y
    rstd = torch.rsqrt(bn.running_var + bn.eps)
    return _BatchNormApply.apply(x, bn.running_mean, rstd, bn.weight, bn.bias)


# ----------------------------------------------------------------------------------------------
# Dice   — layers/activation.py:10-34:  p = sigmoid(bn(x)),  y = p x + (1 - p) alpha x
# ----------------------------------------------------------------------------------------------
class _DiceGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, xhat, alpha):
        x, xhat = _unit_inner(x), _unit_inner(xhat)
        alpha = alpha.contiguous()
        ctx.save_for_backward(x, xhat, alpha)
        return hip.dice_gate_fwd(x, xhat, alpha)

    @staticmethod
    def backward(ctx, dy):
        x, xhat, alpha = ctx.saved_tensors
        dxd, dxh, dal = hip.dice_gate_bwd(x, xhat, alpha, _unit_inner(dy))
        dalpha = hip.batchnorm_colsum(dal) if ctx.needs_input_grad[2] else None  # deterministic two-stage column sums
        return dxd, dxh, dalpha


def dice(x, bn: torch.nn.BatchNorm1d, alpha):
    """Dice on the HIP kernels for 2-D x [M, N]: batch_norm (rp_batchnorm_*, running statistics updated in training
    mode exactly like nn.BatchNorm1d) + the gate (rp_dice_gate_*)."""
    return _DiceGate.apply(x, batch_norm(x, bn), alpha)


# ----------------------------------------------------------------------------------------------
# Dropout (training mode, p > 0)   — layers/deep.py:66-68, the multi-task towers
# ----------------------------------------------------------------------------------------------
class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p: float):
        x = _unit_inner(x)
        y, mask = hip.dropout_fwd(x, p)
        ctx.p = p
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return hip.dropout_bwd(_unit_inner(dy), mask, ctx.p), None


def dropout(x, p: float):
    """y = x * keep / (1 - p) on the HIP kernel (rp_dropout_*); x is [..., N], any leading shape."""
    if x.dim() == 2:
        return _Dropout.apply(x, float(p))
    lead = x.shape[:-1]
    return _Dropout.apply(x.reshape(-1, x.shape[-1]), float(p)).reshape(*lead, x.shape[-1])


# ----------------------------------------------------------------------------------------------
# K10  sum of logits -> sigmoid -> BCE(mean)   — ranking/deepfm.py:61-63, multi_task/mmoe.py:127
# ----------------------------------------------------------------------------------------------
class _SigmoidBCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, label, apply_sigmoid: bool, p_eps: float, weight: float, *addends):
        ctx.set_materialize_grads(False)  # `pred` is normally not differentiated: no zero dpred, no dead arithmetic
        adds = [a.contiguous() for a in addends]
        label = label.contiguous()
        pred, loss = hip.sigmoid_bce_fwd(adds, label, apply_sigmoid, p_eps, weight)
        ctx.cfg = (apply_sigmoid, p_eps, weight, len(adds), [tuple(a.shape) for a in addends])
        ctx.save_for_backward(pred, label)
        return pred, loss

    @staticmethod
    def backward(ctx, dpred, dloss):
        pred, label = ctx.saved_tensors
        apply_sigmoid, p_eps, weight, n, shapes = ctx.cfg
        dz = None
        if dloss is not None:
            dz = hip.sigmoid_bce_bwd(pred, label, dloss, apply_sigmoid, p_eps, weight)
        if dpred is not None:  # someone differentiated through `pred` itself (rare): torch ops on device
            extra = dpred * (pred * (1 - pred) if apply_sigmoid else 1.0)
            dz = extra if dz is None else dz + extra
        outs = [None if dz is None else dz.reshape(s) for s in shapes]
        return (None, None, None, None, *outs)


def sigmoid_bce(addends: Sequence[torch.Tensor], label: torch.Tensor, apply_sigmoid=True, p_eps=0.0, weight=1.0):
    """pred [B,1] = sigmoid(sum(addends)); loss = weight * mean BCE(pred + p_eps, label)."""
    return _SigmoidBCE.apply(label, apply_sigmoid, p_eps, weight, *addends)


class _SigmoidBCEMulti(torch.autograd.Function):
    """The multi-task loss (multi_task/mmoe.py:127, towers.py): forward(T, apply_sigmoid, p_eps, weights, *labels, *logits) ->
    (pred_1 .. pred_T, loss) with loss = sum_i weights[i] * mean BCE(pred_i + p_eps, label_i): every task's launch adds its
    term to ONE device scalar, in task order — the fp32 additions of the reference's python sum without its ATen launches."""

    @staticmethod
    def forward(ctx, T: int, apply_sigmoid: bool, p_eps: float, weights, *rest):
        ctx.set_materialize_grads(False)
        labels = [t.contiguous() for t in rest[:T]]
        logits = [t.contiguous() for t in rest[T:]]
        preds, loss = [], None
        for i in range(T):
            pred, loss = hip.sigmoid_bce_fwd([logits[i]], labels[i], apply_sigmoid, p_eps, float(weights[i]), add_to=loss)
            preds.append(pred)
        ctx.cfg = (T, apply_sigmoid, p_eps, [float(w) for w in weights], [tuple(t.shape) for t in rest[T:]])
        ctx.save_for_backward(*preds, *labels)
        return (*preds, loss)

    @staticmethod
    def backward(ctx, *grads):
        T, apply_sigmoid, p_eps, weights, shapes = ctx.cfg
        saved = ctx.saved_tensors
        preds, labels = saved[:T], saved[T:]
        dloss = grads[T]
        outs = []
        for i in range(T):
            dz = None
            if dloss is not None:
                dz = hip.sigmoid_bce_bwd(preds[i], labels[i], dloss, apply_sigmoid, p_eps, weights[i])
            if grads[i] is not None:  # someone differentiated through a prediction itself (rare): torch ops on device
                extra = grads[i] * (preds[i] * (1 - preds[i]) if apply_sigmoid else 1.0)
                dz = extra if dz is None else dz + extra
            outs.append(None if dz is None else dz.reshape(shapes[i]))
        return (None, None, None, None) + (None,) * T + tuple(outs)


def sigmoid_bce_multi(logits: Sequence[torch.Tensor], labels: Sequence[torch.Tensor], weights, apply_sigmoid=True, p_eps=0.0):
    """([pred_i [B,1]], loss) with loss = sum_i weights[i] * mean BCE(sigmoid(logits[i]) + p_eps, labels[i])"""
    T = len(logits)
    out = _SigmoidBCEMulti.apply(T, apply_sigmoid, float(p_eps), tuple(float(w) for w in weights), *labels, *logits)
    return list(out[:T]), out[T]


class _SigmoidSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *addends):
        pred, _ = hip.sigmoid_bce_fwd([a.contiguous() for a in addends], None, True)
        ctx.shapes = [tuple(a.shape) for a in addends]
        ctx.save_for_backward(pred)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        (pred,) = ctx.saved_tensors
        dz = dpred * pred * (1 - pred)
        return tuple(dz.reshape(s) for s in ctx.shapes)


def sigmoid_sum(addends: Sequence[torch.Tensor]):
    return _SigmoidSum.apply(*addends)


# ----------------------------------------------------------------------------------------------
# K1/K2/K3  multi-table gather (+dense concat, +FM)   — layers/embedding.py:59-63 etc.
# `store` is the arena-backed EmbeddingLayer; its per-table Parameters are passed only so that the
# node is connected to the graph: their gradients are written by the kernel straight into the
# layer's dense gradient arena (exposed as each Parameter's .grad), not returned through autograd.
# ----------------------------------------------------------------------------------------------
class _EmbedGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, store, idx: List[torch.Tensor], dense: List[torch.Tensor], ldx: int, want_fm: bool, meta,
                *tables):
        need_grad = any(ctx.needs_input_grad[6:])
        pre = store._presorted  # set by the lazy-Adam replay: keys already computed and sorted for this batch
        store._presorted = None
        row_base, row_count = meta if meta is not None else (store.row_base, store.row_count)
        x, fm, ssum, keys = hip.embed_gather_fwd(store.arena, row_base, row_count, idx, dense, ldx,
                                                 want_fm, want_fm and need_grad, need_grad and pre is None,
                                                 store.err_flag)
        ctx.store, ctx.want_fm, ctx.B = store, want_fm, idx[0].shape[0]
        # link to the first Linear that consumes x (DeepFM passes it on as `fm_link`): FM fold / fused dgrad
        ctx.link = store._fm_link = FMFold(ssum, len(idx) * store.embedding_dim, store.embedding_dim) \
            if (need_grad and meta is None) else None
        ctx.presorted = None if (pre is None or not need_grad) else (pre[1], pre[2])
        if pre is not None:
            keys = pre[0] if need_grad else None
        ctx.save_for_backward(keys, ssum)
        if want_fm:
            return x, fm
        return x

    @staticmethod
    def backward(ctx, dx, dfm=None):
        keys, ssum = ctx.saved_tensors
        store = ctx.store
        fused = None
        if ctx.link is not None and ctx.link.dgrad is not None:
            fused, ctx.link.dgrad = ctx.link.dgrad, None
            ctx.link.fused = True
            if dx is not None and dx.dim() == 2 and dx.stride(0) == 0 and dx.stride(1) == 0:
                dx = None  # the placeholder of the fused Linear: no other consumer of x contributed
        if dx is not None:
            dx = _unit_inner(dx)
        if ctx.link is not None and ctx.link.extra is not None:
            extra, ctx.link.extra = ctx.link.extra, None
            if fused is not None and dx is None and extra.shape[1] == len(store.emb_feature) * store.embedding_dim:
                # the first Linear left its dgrad to this backward (rp_embed_grad_gemm forms the dX rows it needs from (dH, W^T)
                # inside the segmented reduce): the token consumer's gradient IS the dx operand of that launch — no dX, no add
                dx = extra
            elif dx is None or fused is not None:
                raise RuntimeError("token_view: the gathered activation needs a consumer that materialises its gradient")
            else:
                hip.add_rows_to(extra, dx[:, :extra.shape[1]])
        gfm = dfm.contiguous() if (ctx.want_fm and dfm is not None) else None
        if ctx.link is not None and ctx.link.folded:
            ssum = None  # g_fm * S is already inside dx (rp_linear_fwd_rowadd)
        store.accumulate_grad(keys, ctx.B, dx, gfm, ssum, presorted=ctx.presorted, fused=fused)
        return (None,) * (6 + len(store.emb_feature))


_WGRAD_STREAMS: dict = {}


def _wgrad_stream(device):
    """the stream the first layer's weight gradient runs on beside the gather backward (RP_WGRAD_OVERLAP=0: none); never
    inside a stream capture (a fork inside a captured graph is replayed without overlap anyway)"""
    if os.environ.get("RP_WGRAD_OVERLAP", "1") == "0" or torch.cuda.is_current_stream_capturing():
        return None
    st = _WGRAD_STREAMS.get(device)
    if st is None:
        st = _WGRAD_STREAMS[device] = hip.make_side_stream(device, "inline")
    return st


class _EmbedGatherLinear(torch.autograd.Function):
    """rp_embed_gather_linear_fwd: lookup + dense concat + FM + the first Linear (+ ReLU) of the MLP in one launch; the
    backward is the weight gradient (x is still written for it) and rp_embed_grad_gemm (dX never exists).  `out_link`: the
    ReluLink shared with the consumer of h1 (the fused MLP tail hands back a gradient that is already masked)."""

    @staticmethod
    def forward(ctx, store, idx, dense, ldx: int, weight, bias, out_link, shadow, *tables):
        pre = store._presorted
        store._presorted = None
        need_grad = any(ctx.needs_input_grad[8:])
        need_w = ctx.needs_input_grad[4] or (bias is not None and ctx.needs_input_grad[5])
        B, K, Kg = idx[0].shape[0], weight.shape[1], len(idx) * store.embedding_dim
        wt = None
        if (need_grad and weight.is_cuda and weight.dtype is torch.float32 and K % 4 != 0 and K > 64
                and os.environ.get("RP_STAGE_ONCE", "1") != "0"):
            # the backward will want W^T (rp_embed_grad_gemm) and this forward wants the aligned copy of W: one launch makes
            # both from one read of W, in front of the forward, and the transpose leaves the backward's critical path
            with torch.no_grad():
                wt, w16 = hip.transpose_copy(weight.detach(), rows_out=ldx, ld_copy=(K + 3) // 4 * 4)
        else:
            w16 = _rows16(weight)
        ctx.wt = wt
        if shadow is not None:
            # bf16-storage training (EmbeddingLayer.bf16_training): rows from the bf16 lookup copy, the activation stored as
            # bf16 for the weight gradient (rp_linear_wgrad_xbf16); everything else as below
            want_keys = pre is None
            # round 5: like the fp32 tables' path, no activation is stored when the tables get gradients — the embedding columns
            # of the weight gradient come from rp_embed_grad_seg over the fp32 MASTER rows (the forward used their bf16 images:
            # a 2^-9-relative inconsistency inside the mode's stated tolerance, and closer to the fp32 reference);
            # RP_GRAD_SEG=0: the stored bf16 activation + rp_linear_wgrad_xbf16 (round 4)
            seg16 = (need_grad and need_w and K > Kg and len(idx) <= 64 and os.environ.get("RP_GRAD_SEG", "1") != "0"
                     and weight.shape[0] == 64 and store.embedding_dim == 64)  # (what rp_embed_grad_seg_fits asks for: ADVICE r5)
            x, h1, fm, ssum, keys = hip.embed_gather_linear_fwd_bf16(shadow, store.row_base, store.row_count, idx, dense, w16, bias,
                                                                     store.err_flag, train_ldx=ldx, want_keys=want_keys,
                                                                     dense_only=seg16)
            ctx.store, ctx.B, ctx.K, ctx.out_link, ctx.has_bias = store, B, K, out_link, bias is not None
            ctx.ldx, ctx.x_mode, ctx.Kg, ctx.need_tables = ldx, ("seg" if seg16 else "bf16"), Kg, need_grad
            ctx.presorted = None if (pre is None or not need_grad) else (pre[1], pre[2])
            if pre is not None:
                keys = pre[0]
            ctx.save_for_backward(keys, ssum, x, h1, weight)
            store._fm_link = None
            return h1, fm
        # x exists only for the weight gradient; inference stores none.  RP_WGRAD_GATHER=1: the weight gradient gathers the
        # embedding rows itself (rp_linear_wgrad_gather: the arena does not change between this forward and its backward) and
        # only the dense columns are stored.  Measured at Criteo shape: forward 0.195 -> 0.139 ms, weight gradient 0.148 ->
        # 0.248 ms (random 256-byte rows, two per wave-instruction, against a streamed activation): a net loss of 0.015 ms per
        # step, so it is OFF by default (bit-identical either way: tests/test_hip_kernels.py).
        # "seg" (round 5, the default where it fits): NO activation is stored at all — the embedding columns of the weight
        # gradient come out of the gather backward itself (rp_embed_grad_seg: one matrix pass per run of equal rows over the
        # table rows it reads anyway), the dense columns from a small weight gradient over xd.  RP_GRAD_SEG=0: the stored x.
        if not need_w:
            x_mode = "none"
        elif os.environ.get("RP_WGRAD_GATHER", "0") == "1" and hip.linear_wgrad_gather_fits(B, 64, K, Kg):
            x_mode = "dense"
        elif (need_grad and K > Kg and len(idx) <= 64 and os.environ.get("RP_GRAD_SEG", "1") != "0"
              and store.embedding_dim == 64 and weight.shape[0] == 64):
            x_mode = "seg"
        else:
            x_mode = "full"
        want_keys = (need_grad or x_mode == "dense") and pre is None
        x, h1, fm, ssum, keys = hip.embed_gather_linear_fwd(store.arena, store.row_base, store.row_count, idx, dense, ldx, w16,
                                                            bias, True, need_grad, want_keys, store.err_flag,
                                                            x_mode="dense" if x_mode == "seg" else x_mode)
        ctx.store, ctx.B, ctx.K, ctx.out_link, ctx.has_bias = store, B, K, out_link, bias is not None
        ctx.ldx, ctx.x_mode, ctx.Kg, ctx.need_tables = ldx, x_mode, Kg, need_grad
        ctx.presorted = None if (pre is None or not need_grad) else (pre[1], pre[2])
        if pre is not None:
            keys = pre[0] if (need_grad or x_mode == "dense") else None
        ctx.save_for_backward(keys, ssum, x, h1, weight)
        store._fm_link = None
        return h1, fm

    @staticmethod
    def backward(ctx, dh1, dfm):
        keys, ssum, x, h1, weight = ctx.saved_tensors
        store = ctx.store
        dh1 = _unit_inner(dh1)
        lk = ctx.out_link
        masked = lk is not None and lk.dx is not None and lk.dx.data_ptr() == dh1.data_ptr() and lk.dx.shape == dh1.shape
        if lk is not None:
            lk.dx = None
        dpre = dh1 if masked else hip.relu_bwd(dh1, h1)
        dw = db = None
        need_w = ctx.needs_input_grad[4] or (ctx.has_bias and ctx.needs_input_grad[5])
        need_t = keys is not None and ctx.need_tables
        if need_w and ctx.x_mode == "none":
            raise RuntimeError("the fused lookup + first layer stored no activation (the forward ran without gradients for the "
                               "layer's weight)")

        seg = None
        if need_w and ctx.x_mode == "seg":
            # the weight gradient's three column groups come from three launches: the dense columns (+ the bias gradient)
            # from rp_linear_wgrad over xd below, the tiny tables' from rp_embed_grad_tiny and the other tables' from
            # rp_embed_grad_seg (both inside store.accumulate_grad)
            dw_seg = torch.empty((64, ctx.K), dtype=torch.float32, device=dpre.device)
            seg = (weight, dw_seg)

        def wgrad(keep=None):
            if ctx.x_mode == "seg":    # x holds the dense columns only (xd [B, 64]); their columns of dw and the bias gradient
                _, db_ = hip.linear_wgrad(dpre, x, ctx.K - ctx.Kg, dw=dw_seg[:, ctx.Kg:], want_bias=ctx.has_bias, keep=keep)
                return dw_seg, db_
            if ctx.x_mode == "bf16":   # the activation was stored as bf16 (bf16-storage training)
                return hip.linear_wgrad_xbf16(dpre, x, ctx.K, want_bias=ctx.has_bias, keep=keep)
            if ctx.x_mode == "dense":  # x holds the dense columns only: the embedding columns are gathered from the arena
                return hip.linear_wgrad_gather(dpre, store.arena, keys, ctx.Kg, x, ctx.K, want_bias=ctx.has_bias)
            return hip.linear_wgrad(dpre, x, ctx.K, want_bias=ctx.has_bias, keep=keep)

        # The weight gradient (streams x: 0.15 ms at Criteo shape) and the fused gather backward (bound by its random row
        # gathers, ~1 TB/s of HBM: 0.30 ms) both depend only on dpre and are independent of each other: the weight gradient
        # runs on a second stream BESIDE the gather backward instead of in front of it.  Same kernels, same inputs: bit-identical.
        wstream = _wgrad_stream(dpre.device) if (need_w and need_t) else None
        # ... and while a LAUNCH PLAN is being recorded (a captured step: one stream), the weight gradient's launches are
        # marked as an inline section: the replay issues them on the plan's second side stream, joined after the gather
        # backward.  Their workspace stays referenced until the join (the capture's allocator would reuse it at once).
        in_plan = (wstream is None and need_w and need_t and ctx.x_mode != "dense"
                   and os.environ.get("RP_WGRAD_OVERLAP", "1") != "0" and hip.LaunchPlan.is_recording())
        keep = []
        # (round 5) in a plan, with the segment-sum-first backward: the LONG main-stream launch is issued first and the short
        # side launches behind it, all forked from the same point (rp_plan_fork2_mark) — issued the other way round the side
        # launches filled every CU and rp_embed_grad_seg (77 KB of LDS per workgroup) started 58 us late (profiles/r05_trace_step.txt)
        # Measured (profiles/r05 lines, alternating runs on one box): 0.946 / 0.947 ms with the main launch first against
        # 0.929 / 0.932 the other way round (long-run means equal, 0.906-0.916): the side launches then stretch to twice their
        # time and the join comes later
        # (re-measured at the end of round 5, after the loss head moved into the MLP tail and the side streams went to the
        #  lowest priority: main launch first 0.8324 / 0.8375 ms against 0.8369 / 0.8384 in the 20-step window, 0.8086 / 0.8099
        #  against 0.8152 / 0.8161 over 600 steps — ON by default now; RP_SEG_FIRST=0: the side launches first)
        seg_first = (in_plan and seg is not None and need_t and not hip.LaunchPlan.ahead
                     and os.environ.get("RP_SEG_FIRST", "1") == "1")
        # catch-up ahead (graph_step): the step's last main-stream launch rewrites table rows, so only the tiny tables' gradient
        # (which READS their rows) runs beside rp_embed_grad_seg and is joined behind it; the weight gradient's dense columns
        # and the deferred side launches are issued after that join and run beside the catch-up, joined at the end of the replay
        ahead = in_plan and hip.LaunchPlan.ahead and seg is not None and need_t and not seg_first
        if seg_first:
            pass  # (the fork is marked inside accumulate_grad, behind the sample-major launch: round 6)
        elif ahead:
            pass
        elif in_plan:
            hip.LaunchPlan.section(2)
            try:
                dw, db = wgrad(keep)
                hip.LaunchPlan.run_deferred()  # (e.g. the MLP tail's second stage: behind the weight gradient, not in front)
            finally:
                hip.LaunchPlan.section(0)
        elif wstream is not None:
            main = torch.cuda.current_stream(dpre.device)
            wstream.wait_stream(main)
            with torch.cuda.stream(wstream):
                dw, db = wgrad()
        elif need_w:
            dw, db = wgrad()
        if need_t:
            wt = ctx.wt if ctx.wt is not None else hip.transpose(weight, rows_out=ctx.ldx)
            gfm = dfm.contiguous() if dfm is not None else None
            store.accumulate_grad(keys, ctx.B, None, gfm, ssum if gfm is not None else None, presorted=ctx.presorted,
                                  fused=(dpre, wt), plan_keep=keep if in_plan else None, seg=seg, seg_first=seg_first,
                                  fork2=hip.LaunchPlan.fork2_mark if seg_first else None)
        if seg_first:
            hip.LaunchPlan.section(2)
            try:
                dw, db = wgrad(keep)
                hip.LaunchPlan.run_deferred()
            finally:
                hip.LaunchPlan.section(0)
        if ahead:
            hip.LaunchPlan.join_only()
            hip.LaunchPlan.section(2)
            try:
                dw, db = wgrad(keep)
                hip.LaunchPlan.run_deferred()
            finally:
                hip.LaunchPlan.section(0)
            # (the workspaces stay alive until the plan is finished: nothing joins them here.  NOT dw / db: a second reference
            #  makes AccumulateGrad clone them — two memcpy nodes, and the step no longer replays as a plan; graph_step holds
            #  the parameters' .grad across zero_grad() instead)
            hip.LaunchPlan._ahead_keep.append(keep)
        elif in_plan:
            hip.LaunchPlan.join()
            del keep
        if wstream is not None:
            main.wait_stream(wstream)  # whoever consumes dw / db (AccumulateGrad, the optimizer) is ordered behind them
            dw.record_stream(main)     # (allocated under the second stream, consumed and freed on the main one)
            if db is not None:
                db.record_stream(main)
        return (None, None, None, None, dw, db, None, None) + (None,) * len(store.emb_feature)


def embed_gather_linear(store, idx, dense, ldx: int, weight, bias, out_link, shadow=None):
    tables = [store.embedding_layer[c].weight for c in store.emb_feature]
    return _EmbedGatherLinear.apply(store, idx, dense, ldx, weight, bias, out_link, shadow, *tables)


def embed_gather(store, idx, dense, ldx: int, want_fm: bool, meta=None):
    """`meta` = (row_base, row_count) of the tables `idx` addresses when that is a subset of the store's fields
    (single-field / sequence lookups); None = all fields in order."""
    tables = [store.embedding_layer[c].weight for c in store.emb_feature]
    return _EmbedGather.apply(store, idx, dense, ldx, want_fm, meta, *tables)


# ----------------------------------------------------------------------------------------------
# pooled multi-id lookup (north_star's CSR / segmented gather + sum-pool): embedding.py:64-71 (`_seq`) followed by
# layers/sequence.py:13-59 (MaskedAveragePooling / MaskedSumPooling) as ONE launch each way
# ----------------------------------------------------------------------------------------------
class _EmbedGatherPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, store, field: int, ids, offsets, L: int, B: int, mode: str, presorted, *tables):
        need_grad = any(ctx.needs_input_grad[8:])
        base, count = store.table_range(field)
        out, inv, bag = hip.embed_gather_pool_fwd(store.arena, base, count, ids, offsets, L, B, mode, store.err_flag,
                                                  need_grad)
        ctx.store, ctx.field, ctx.L, ctx.presorted = store, field, L, presorted
        ctx.save_for_backward(ids, inv, bag)
        return out

    @staticmethod
    def backward(ctx, g):
        ids, inv, bag = ctx.saved_tensors
        store = ctx.store
        store.accumulate_grad(None, 0, None, None, None, presorted=ctx.presorted,
                              pool=(g.contiguous(), inv, bag, ctx.L, ctx.field, ids))
        return (None,) * (8 + len(store.emb_feature))


def embed_gather_pool(store, field: int, ids, offsets, L: int, B: int, mode: str, presorted=None):
    tables = [store.embedding_layer[c].weight for c in store.emb_feature]
    return _EmbedGatherPool.apply(store, field, ids, offsets, L, B, mode, presorted, *tables)


class _SeqPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, mode: str):
        out, inv = hip.seq_pool_fwd(e, mode, ctx.needs_input_grad[0])
        ctx.L = e.shape[1]
        ctx.save_for_backward(inv)
        return out

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return hip.seq_pool_bwd(g.contiguous(), inv, ctx.L), None


def seq_pool(e, mode: str):
    """MaskedSumPooling ("sum") / MaskedAveragePooling ("average") of an explicit [B, L, D] HIP tensor"""
    return _SeqPool.apply(e.float().contiguous(), mode)
