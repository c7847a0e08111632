"""MultiHeadAttention — mirror of reference utils/attentions.py:44-116 (version 'v2').

Parameter names (linear_k/v/q/final, layer_norm) are the checkpoint contract.  The TTA path only consumes
the attention map (multi_graph_matching.py:498), which comes from csrc/mha.hip without autograd; ``forward`` returns the
(output, attention) pair and is differentiable when gradients are enabled (the source-training loss U_sup.forward
back-propagates through ``output``, multi_graph_matching.py:100-106): the four projections run on the MFMA GEMM
(ops.LinearFn), softmax / LayerNorm are torch glue."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops


class MultiHeadAttention(nn.Module):
    def __init__(self, model_dim=256, num_heads=4, dropout=0.0, version='v2'):
        super().__init__()
        self.dim_per_head = model_dim // num_heads
        self.num_heads = num_heads
        self.linear_k = nn.Linear(model_dim, self.dim_per_head * num_heads)
        self.linear_v = nn.Linear(model_dim, self.dim_per_head * num_heads)
        self.linear_q = nn.Linear(model_dim, self.dim_per_head * num_heads)
        self.linear_final = nn.Linear(model_dim, model_dim)
        self.dropout = nn.Dropout(dropout)
        self.drop_p = float(dropout)
        self.layer_norm = nn.LayerNorm(model_dim)
        self.version = version
        self._calls = 0

    def attention_only(self, x, seed=None):
        """(n, d) -> (n, n) attention map; no autograd (nothing downstream of it carries gradient on this path)."""
        if self.version != 'v2' or self.num_heads != 1:
            raise NotImplementedError("only the 1-head 'v2' configuration used by MGM3_unsup is implemented")
        with torch.no_grad():
            x = x.detach().contiguous()
            n = x.shape[0]
            q = ops.linear_raw(x, self.linear_q.weight, self.linear_q.bias)
            k = ops.linear_raw(x, self.linear_k.weight, self.linear_k.bias)
            p = self.drop_p if self.training else 0.0
            self._calls += 1
            sd = self._calls if seed is None else seed
            scale = (k.shape[-1] // self.num_heads) ** -0.5            # attentions.py:80
            a = ops.mha_adjacency(q, k, ops.graphs([n]), [n], scale, p, sd, zero_diag=False).view(n, n)
            return a

    def forward(self, key_value_query, attn_mask=None):
        key, value, query = key_value_query
        if torch.is_grad_enabled() and (query.requires_grad or self.linear_q.weight.requires_grad):
            if self.version != 'v2' or self.num_heads != 1:
                raise NotImplementedError("only the 1-head 'v2' configuration used by the matching losses is implemented")
            lin = lambda x, l: ops.LinearFn.apply(x.float().contiguous(), l.weight, l.bias)
            k, v, q = lin(key, self.linear_k), lin(value, self.linear_v), lin(query, self.linear_q)
            scale = (k.shape[-1] // self.num_heads) ** -0.5            # attentions.py:80
            attention = self.dropout(torch.softmax((q @ k.t()) * scale, dim=1))   # attentions.py:31-42
            out = self.dropout(lin(attention @ v, self.linear_final))
            return self.layer_norm(query + out).squeeze(), attention.squeeze()
        attention = self.attention_only(query)
        with torch.no_grad():
            v = ops.linear_raw(value.detach().contiguous(), self.linear_v.weight, self.linear_v.bias)
            ctxv = attention @ v
            out = ops.linear_raw(ctxv.contiguous(), self.linear_final.weight, self.linear_final.bias)
            out = self.layer_norm(query.detach() + self.dropout(out))
        return out.squeeze(), attention.squeeze()
