"""Feature-interaction blocks of the hot path — drop-ins for rec_pangu/models/layers/interaction.py:
InnerProductLayer (:12-52, the two pooling outputs the ranking models use), FM_Layer (:225-235),
CrossInteractionLayer / CrossNet (:119-141), CompressedInteractionNet (:144-171).
Parameter names/shapes follow the reference so its checkpoints load.
"""
import torch
from torch import nn


class InnerProductLayer(nn.Module):
    """product_sum_pooling -> [B,1];  Bi_interaction_pooling -> [B,D]  (interaction.py:36-44).

    In DeepFM/FM on a HIP device the second-order term is produced inside the gather kernel
    (EmbeddingLayer.gather_concat(want_fm=True)), so this module is only reached with an explicit
    [B,F,D] tensor."""
    _SUPPORTED = ("product_sum_pooling", "Bi_interaction_pooling")

    def __init__(self, num_fields=None, output="product_sum_pooling"):
        super(InnerProductLayer, self).__init__()
        if output not in self._SUPPORTED:
            raise ValueError("InnerProductLayer output={} is not supported.".format(output))
        self._output_type = output

    def forward(self, feature_emb):
        if feature_emb.is_cuda and feature_emb.dtype == torch.float32 and feature_emb.shape[-1] % 4 == 0:
            from ... import functional as Fh
            return Fh.fm_pool(feature_emb, self._output_type == "Bi_interaction_pooling")
        field_sum = feature_emb.sum(dim=1)
        bi = 0.5 * (field_sum * field_sum - (feature_emb * feature_emb).sum(dim=1))
        if self._output_type == "Bi_interaction_pooling":
            return bi
        return bi.sum(dim=-1, keepdim=True)


class FM_Layer(nn.Module):
    def __init__(self, final_activation=None, use_bias=True):
        super(FM_Layer, self).__init__()
        self.inner_product_layer = InnerProductLayer(output="product_sum_pooling")
        self.final_activation = final_activation

    def forward(self, feature_emb_list):
        out = self.inner_product_layer(feature_emb_list)
        return out if self.final_activation is None else self.final_activation(out)


class CrossInteractionLayer(nn.Module):
    """one DCN-v1 cross layer: (X_i . w) * X_0 + b   (interaction.py:119-127)"""

    def __init__(self, input_dim):
        super(CrossInteractionLayer, self).__init__()
        self.weight = nn.Linear(input_dim, 1, bias=False)
        self.bias = nn.Parameter(torch.zeros(input_dim))

    def forward(self, X_0, X_i):
        return self.weight(X_i) * X_0 + self.bias


class CrossNet(nn.Module):
    """X_{l+1} = X_l + (X_l . w_l) X_0 + b_l   (interaction.py:130-141)"""

    def __init__(self, input_dim, num_layers):
        super(CrossNet, self).__init__()
        self.num_layers = num_layers
        self.input_dim = input_dim
        self.cross_net = nn.ModuleList(CrossInteractionLayer(input_dim) for _ in range(num_layers))

    def stacked(self):
        """([L,d] weights, [L,d] biases): the per-layer parameters stacked for the one-pass HIP kernel
        (torch.stack keeps them on the autograd tape, so the gradients flow back to each layer)."""
        W = torch.stack([layer.weight.weight.reshape(-1) for layer in self.cross_net])
        Bv = torch.stack([layer.bias for layer in self.cross_net])
        return W, Bv

    def forward(self, X_0, fc: nn.Linear = None):
        """X_L [B,d]; with `fc` (a Linear(d,1), as in DCN) the fused kernel returns fc(X_L) [B,1] instead."""
        if X_0.is_cuda:
            from ... import functional as Fh
            lw = [layer.weight.weight for layer in self.cross_net]
            lb = [layer.bias for layer in self.cross_net]
            if fc is not None:
                return Fh.crossnet(X_0, lw, lb, fc.weight, fc.bias)
            return Fh.crossnet(X_0, lw, lb)
        X_i = X_0
        for layer in self.cross_net:
            X_i = X_i + layer(X_0, X_i)
        return X_i if fc is None else fc(X_i)


class CompressedInteractionNet(nn.Module):
    """CIN (interaction.py:144-171): X_k[b,o,:] = sum_{h,m} W_k[o, h*M+m] X_0[b,h,:] X_{k-1}[b,m,:] + bias_k[o];
    no activation, no split-half; sum-pool over the embedding axis, concat, fc -> [B, output_dim].
    Weights are kept as Conv1d(kernel_size=1) modules for state_dict compatibility
    (`cin_layer.layer_k.weight` is [O, H*M, 1])."""

    def __init__(self, num_fields, cin_layer_units, output_dim=1):
        super(CompressedInteractionNet, self).__init__()
        self.cin_layer_units = cin_layer_units
        self.num_fields = num_fields
        self.fc = nn.Linear(sum(cin_layer_units), output_dim)
        self.cin_layer = nn.ModuleDict()
        prev = num_fields
        for i, unit in enumerate(cin_layer_units):
            self.cin_layer["layer_" + str(i + 1)] = nn.Conv1d(num_fields * prev, unit, kernel_size=1)
            prev = unit

    def _forward_hip(self, feature_emb, as_list: bool = False):
        """fp32-MFMA CIN kernels (rp_cin_layer_*).  Every layer but the last runs at full width and keeps X_k for
        the next one; the LAST layer only feeds sum-pooling and `fc`, both linear, and the CIN has no activation, so
            sum_o c[o] * sum_d X_L[b,o,d] = sum_d sum_{h,m} V[h,m] X_0[b,h,d] X_{L-1}[b,m,d] + D * (c . bias_L),
            V = sum_o c[o] W_L[o]            (c = the slice of fc.weight that multiplies the last layer's pooling)
        and it is evaluated as a single-output-channel layer with weights V: 1/O_L of the work, same algebra."""
        from ... import functional as Fh
        from ... import hip
        if feature_emb.dim() == 2:  # (already the [B, H D] row buffer: a model that hands over its own view of x)
            B, H, x0 = feature_emb.shape[0], self.num_fields, feature_emb
            D = x0.shape[1] // H
        else:
            B, H, D = feature_emb.shape
            x0 = feature_emb.reshape(B, H * D)
        if any(u > 32 for u in self.cin_layer_units[:-2]) and (x0.stride(0) % 4 != 0 or x0.data_ptr() % 16 != 0):
            x0 = x0.contiguous()  # the chunked bf16 middle layers need 16-byte aligned rows (the models' x buffer has them)
        L = len(self.cin_layer_units)
        if self.fc.weight.shape[0] != 1:
            raise NotImplementedError("CIN on HIP collapses the last layer into fc: output_dim must be 1")
        # round 6: with a layer in front of the last one and rp_cin_last covering the last, the whole head is library launches —
        # fc.weight split without autograd slice nodes, V = c . W_L / c . b_L / the scalars / their gradients in cin_head, the
        # last layer's gradient of X_0 added into the first layer's inside its backward (CINLink)
        M_last = self.cin_layer_units[-2] if L >= 2 else H
        fused_head = L >= 2 and hip.cin_last_fits(H, M_last, D)
        link = Fh.CINLink() if (fused_head and torch.is_grad_enabled() and x0.requires_grad) else None
        xp, M, pooled, n_prev = None, H, [], 0
        for i in range(L - 1):
            conv = self.cin_layer["layer_" + str(i + 1)]
            O = conv.weight.shape[0]
            if xp is not None and Fh.cin_middle_fits(H, D, x0, xp):
                # a middle layer (fed by any number of maps) on the bf16 matrix core, 32 maps per launch
                X_i, p_i = Fh.cin_middle(x0, xp, conv.weight.view(O, H * M), conv.bias, H, M, D)
            else:
                X_i, p_i = Fh.cin_layer(x0, xp, conv.weight.view(O, H * M), conv.bias, H, M, D, want_out=True,
                                        link=link if i == 0 else None)
            pooled.append(p_i)
            xp, M, n_prev = X_i.view(B, O * D), O, n_prev + O
        conv = self.cin_layer["layer_" + str(L)]
        O = conv.weight.shape[0]
        if fused_head:
            c_prev, c_last = Fh.row_split(self.fc.weight, n_prev)
            head = Fh.cin_head(x0, xp, conv.weight.view(O, H * M), conv.bias, c_last, self.fc.bias, H, M, D, link)
            pool_logit = Fh.linear_act(pooled[0] if len(pooled) == 1 else torch.cat(pooled, dim=-1), c_prev, None)
            return [head, pool_logit] if as_list else head + pool_logit
        c_last = self.fc.weight[:, n_prev:]                                  # [out_dim, O_L]
        V = c_last @ conv.weight.view(O, H * M)                              # [1, H*M]   (weight-space, tiny)
        vb = (c_last @ conv.bias.view(O, 1)).view(1)                         # c . bias_L
        from ... import hip
        if hip.cin_last_fits(H, M, D):
            # dedicated HBM-bound kernels for the single collapsed channel; the bias enters as D * (c . bias_L)
            logit = Fh.cin_last(x0, xp, V, H, M, D) + D * vb
        else:
            logit = Fh.cin_layer(x0, xp, V, vb, H, M, D, want_out=False)     # [B,1] = sum_o c[o] pooled_L[b,o]
        if pooled:
            logit = logit + Fh.linear_act(torch.cat(pooled, dim=-1), self.fc.weight[:, :n_prev].contiguous(), None)
        logit = logit + self.fc.bias
        return [logit] if as_list else logit

    def hip_supported(self, H, D=None) -> bool:
        """<= 32 fields and a single output.  Middle layers (not first, not last) fed by more than 32 maps run on the
        chunked bf16 form (functional._CINChunked: D in {32, 64}, not in the 'fp32' matrix-core mode); the f32-MFMA
        kernels of cin.hip cover middle layers of <= 32 maps; anything else is composed from device ops in forward()."""
        from ... import hip
        units = self.cin_layer_units
        if not (H <= 32 and self.fc.weight.shape[0] == 1):
            return False
        if all(u <= 32 for u in units[:-2]):
            return True
        return hip.get_matmul_precision() != "fp32" and D in (32, 64)

    def forward(self, feature_emb, as_list: bool = False):
        """as_list (HIP callers that add their logits inside the loss launch): the logit as a list of [B, 1] addends"""
        if feature_emb.dim() == 2:
            H, D = self.num_fields, feature_emb.shape[1] // self.num_fields
            if not (feature_emb.is_cuda and self.hip_supported(H, D)):
                feature_emb = feature_emb.reshape(feature_emb.shape[0], H, D)
        B = feature_emb.shape[0]
        if feature_emb.dim() == 3:
            _, H, D = feature_emb.shape
        if feature_emb.is_cuda and self.hip_supported(H, D):
            return self._forward_hip(feature_emb, as_list)
        if feature_emb.is_cuda:
            from ... import hip
            hip.note_torch_path(f"CompressedInteractionNet with {H} fields, D={D}, layers {list(self.cin_layer_units)}, "
                                f"{self.fc.weight.shape[0]} outputs (the kernels cover <= 32 fields, one output; wide middle "
                                "layers need D in {32, 64})")
        X_0, X_i, pooled = feature_emb, feature_emb, []
        for i in range(len(self.cin_layer_units)):
            conv = self.cin_layer["layer_" + str(i + 1)]
            M = X_i.shape[1]
            W = conv.weight.view(conv.weight.shape[0], H, M)
            # contract (h, m) without materialising the [B, H*M, D] outer product the reference builds
            t = torch.einsum("ohm,bmd->bohd", W, X_i)
            X_i = (t * X_0.unsqueeze(1)).sum(dim=2) + conv.bias.view(1, -1, 1)
            pooled.append(X_i.sum(dim=-1))
        out = self.fc(torch.cat(pooled, dim=-1))
        return [out] if as_list else out
