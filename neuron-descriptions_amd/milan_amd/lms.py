"""The LSTM language model used for PMI decoding / reranking.

Mirror of the reference's `src/milan/lms.py:17-101` (inference surface):
Embedding(V,E,padding_idx) -> 2-layer LSTM -> Linear(V) + LogSoftmax, and
`forward(inputs, reduce=True)` = masked sequence log-probability with the
reference's stop-mask off-by-one (lms.py:93-96).  Works attached to a Decoder
(sharing its HIP context) or standalone, like the reference's module.
Training (`fit`) and text scoring (`logp`, needs the spaCy tokenizer) are out
of scope.
"""
from typing import Any, Mapping, Optional

import torch
from torch import nn

from milan_amd import hip, lang, params


class LanguageModel(nn.Module):
    """Parameter owner + HIP entry point for the LM."""

    def __init__(self,
                 indexer: lang.Indexer,
                 embedding_size: int = 128,
                 hidden_size: int = 512,
                 layers: int = 2,
                 dropout: float = .5):
        super().__init__()
        self.indexer = indexer
        self.embedding_size = embedding_size
        self.hidden_size = hidden_size
        self.layers = layers
        self.dropout = dropout
        f = torch.float32
        v, e, h = len(indexer), embedding_size, hidden_size
        spec = {'embedding.weight': ((v, e), f)}
        for layer in range(layers):
            spec[f'lstm.weight_ih_l{layer}'] = ((4 * h, e if layer == 0 else h),
                                                f)
            spec[f'lstm.weight_hh_l{layer}'] = ((4 * h, h), f)
            spec[f'lstm.bias_ih_l{layer}'] = ((4 * h,), f)
            spec[f'lstm.bias_hh_l{layer}'] = ((4 * h,), f)
        spec['output.0.weight'] = ((v, h), f)
        spec['output.0.bias'] = ((v,), f)
        params.build(spec, root=self)
        # inside a Decoder the LM scores through the decoder's HIP context (one
        # packed copy of the weights); on its own it builds an LM-only context
        self._owner = None
        self._ctx: Optional[hip.Context] = None
        self._ctx_key = None

    def _context(self) -> hip.Context:
        if self._owner is not None and self._owner() is not None:
            return self._owner()._context()
        device = hip.require_device(self.embedding.weight.device)
        key = (device, tuple(p._version for p in self.parameters()))
        if self._ctx is None or self._ctx_key != key:
            sd = {f'lm.{k}': v for k, v in self.state_dict().items()}
            dims = hip.make_dims(sd, len(self.indexer.vocab))
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = hip.Context(dims, sd, device)
            self._ctx_key = key
        return self._ctx

    def forward(self,
                inputs: torch.Tensor,
                reduce: bool = False,
                masks: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Log-probabilities under the LM (reference lms.py:58-101).

        `reduce=False`: (batch, length, vocab) next-token log-probs at every
        position.  `reduce=True`: (batch,) sequence log-probs, the first token
        taken as given; `masks` (batch, length-1) weights the token terms, by
        default everything after the first stop token is dropped -- with the
        reference's off-by-one: the term predicting the token AFTER the stop
        still counts (lms.py:93-95).
        """
        ctx = self._context()
        if reduce and masks is None:
            return ctx.lm_score(inputs)  # fused: never materialises (B,L,V)
        lps = ctx.lm_logprobs(inputs)
        if not reduce:
            return lps
        targets = inputs[:, 1:].to(lps.device)
        picked = lps[:, :-1].gather(2, targets.unsqueeze(-1)).squeeze(-1)
        return picked.mul(masks.to(lps.device)).sum(dim=-1)

    def properties(self) -> Mapping[str, Any]:
        return {
            'indexer': self.indexer,
            'embedding_size': self.embedding_size,
            'hidden_size': self.hidden_size,
            'layers': self.layers,
            'dropout': self.dropout,
        }
