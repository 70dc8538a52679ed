"""Stand-in for `allennlp.nn.beam_search` (allennlp==2.10, requirements.txt:7).

TEST INFRASTRUCTURE.  allennlp is not installable offline, so the golden
generator (`make_golden_beam.py`) installs this module under the name
`allennlp.nn.beam_search` and then runs the REFERENCE's own, unmodified
`Decoder.forward(strategy='beam' | 'rerank')` / `Decoder.predict`
(src/milan/decoders.py:465-523, :809-871) on top of it: the `BeamSearch(...)`
construction, the `step` closure, the `AllenNLPDecoderState` dict with its
`h_lm` / `c_lm` permutes (:153-198) and the rerank assembly all execute as
written upstream.

It is a class-shaped restatement of the 2.10 module -- `Sampler` /
`DeterministicSampler` (`sample_nodes`, `sample_beams`), `FinalSequenceScorer`
/ `SequenceLogProbabilityScorer`, and `BeamSearch` with `search`, `_search`,
`_update_initial_state` and `_update_state` over a GENERIC state dict
(gather by backpointer, whatever the tensors are) and
`_reconstruct_sequences` -- written from the published semantics, deliberately
structured like the library and NOT like `oracle.beam_search` (which reorders
by flat row index and knows the decoder), so that the two are independent
statements of the same algorithm.  Constraints and stochastic samplers are not
used by the reference's call site and are not restated.
"""
import inspect
import warnings
from typing import Callable, Dict, List, Optional, Tuple

import torch

StateType = Dict[str, torch.Tensor]


def min_value_of_dtype(dtype: torch.dtype) -> float:
    """allennlp.nn.util.min_value_of_dtype: the most negative finite value."""
    if not dtype.is_floating_point:
        raise TypeError('min_value_of_dtype is used on float tensors here')
    return torch.finfo(dtype).min


class Sampler:
    """Base sampler: `sample_beams` defaults to a plain top-k."""

    def init_state(self, start_class_log_probabilities: torch.Tensor,
                   batch_size: int, num_classes: int) -> StateType:
        del start_class_log_probabilities, batch_size, num_classes
        return {}

    def sample_nodes(self, log_probs: torch.Tensor, per_node_beam_size: int,
                     state: StateType):
        raise NotImplementedError

    def sample_beams(self, log_probs: torch.Tensor, beam_size: int,
                     state: StateType):
        del state
        selected_log_probs, selected_indices = torch.topk(log_probs,
                                                          beam_size,
                                                          dim=-1)
        return selected_log_probs, selected_indices, {}


class DeterministicSampler(Sampler):
    """The default sampler: take the `per_node_beam_size` best classes."""

    def sample_nodes(self, log_probs: torch.Tensor, per_node_beam_size: int,
                     state: StateType):
        del state
        selected_log_probs, selected_indices = torch.topk(log_probs,
                                                          per_node_beam_size,
                                                          dim=-1)
        return selected_log_probs, selected_indices, {}


class FinalSequenceScorer:

    def score(self, predictions: torch.Tensor, log_probabilities: torch.Tensor,
              end_index: int) -> torch.Tensor:
        raise NotImplementedError


class SequenceLogProbabilityScorer(FinalSequenceScorer):
    """Default scorer: the accumulated log-probability, unchanged."""

    def score(self, predictions: torch.Tensor, log_probabilities: torch.Tensor,
              end_index: int) -> torch.Tensor:
        del predictions, end_index
        return log_probabilities


class BeamSearch:

    def __init__(self,
                 end_index: int,
                 max_steps: int = 50,
                 beam_size: int = 10,
                 per_node_beam_size: Optional[int] = None,
                 sampler: Optional[Sampler] = None,
                 min_steps: Optional[int] = None,
                 final_sequence_scorer: Optional[FinalSequenceScorer] = None,
                 constraints=None) -> None:
        if not max_steps > 0:
            raise ValueError('max_steps must be positive')
        if not beam_size > 0:
            raise ValueError('beam_size must be positive')
        if per_node_beam_size is not None and not per_node_beam_size > 0:
            raise ValueError('per_node_beam_size must be positive')
        if min_steps is not None:
            if not min_steps >= 0:
                raise ValueError('min_steps must be non-negative')
            if not min_steps <= max_steps:
                raise ValueError('min_steps must be <= max_steps')
        if constraints:
            raise NotImplementedError('constraints are not restated')
        self._end_index = end_index
        self.max_steps = max_steps
        self.beam_size = beam_size
        self.per_node_beam_size = per_node_beam_size or beam_size
        self.sampler = sampler or DeterministicSampler()
        self.min_steps = min_steps or 0
        self.final_sequence_scorer = (final_sequence_scorer or
                                      SequenceLogProbabilityScorer())

    @staticmethod
    def _reconstruct_sequences(predictions: List[torch.Tensor],
                               backpointers: List[torch.Tensor]
                               ) -> List[torch.Tensor]:
        # Walk the backpointers from the last step to the first; the result is
        # in reverse time order.
        reconstructed_predictions = [predictions[-1].unsqueeze(2)]
        if not backpointers:
            return reconstructed_predictions
        cur_backpointers = backpointers[-1]
        for timestep in range(len(predictions) - 2, 0, -1):
            cur_preds = predictions[timestep].gather(
                1, cur_backpointers).unsqueeze(2)
            reconstructed_predictions.append(cur_preds)
            cur_backpointers = backpointers[timestep - 1].gather(
                1, cur_backpointers)
        final_preds = predictions[0].gather(1, cur_backpointers).unsqueeze(2)
        reconstructed_predictions.append(final_preds)
        return reconstructed_predictions

    @torch.no_grad()
    def search(self, start_predictions: torch.Tensor, start_state: StateType,
               step: Callable) -> Tuple[torch.Tensor, torch.Tensor]:
        step_signature = inspect.signature(step)
        if len(step_signature.parameters) < 3:
            # step functions without a timestep argument are wrapped
            old_step = step

            def new_step(last_predictions: torch.Tensor, state: StateType,
                         time_step: int):
                del time_step
                return old_step(last_predictions, state)

            return self._search(start_predictions, start_state, new_step)
        return self._search(start_predictions, start_state, step)

    def _search(self, start_predictions: torch.Tensor, start_state: StateType,
                step: Callable) -> Tuple[torch.Tensor, torch.Tensor]:
        batch_size = start_predictions.size()[0]

        # One (batch_size, beam_size) tensor per step; start symbols implicit.
        predictions: List[torch.Tensor] = []
        # One (batch_size, beam_size) tensor per step after the first: index
        # of the parent in the previous step's beam.
        backpointers: List[torch.Tensor] = []

        # First step: batch_size rows -> the top `beam_size` continuations.
        start_class_log_probabilities, state = step(start_predictions,
                                                    start_state, 0)
        num_classes = start_class_log_probabilities.size()[1]
        if self.per_node_beam_size > num_classes:
            raise ValueError(
                f'Target vocab size ({num_classes:d}) too small relative to '
                f'per_node_beam_size ({self.per_node_beam_size:d}).')
        sampler_state = self.sampler.init_state(start_class_log_probabilities,
                                                batch_size, num_classes)
        if self.min_steps >= 1:
            start_class_log_probabilities[:, self._end_index] = \
                min_value_of_dtype(start_class_log_probabilities.dtype)

        (start_top_log_probabilities, start_predicted_classes,
         sampler_state) = self.sampler.sample_beams(
             start_class_log_probabilities, self.beam_size, sampler_state)

        if self.beam_size == 1 and (start_predicted_classes
                                    == self._end_index).all():
            warnings.warn(
                'Empty sequences predicted. You may want to increase the beam '
                'size or ensure your step function is working properly.',
                RuntimeWarning)
            return (start_predicted_classes.unsqueeze(-1),
                    start_top_log_probabilities)

        last_log_probabilities = start_top_log_probabilities
        predictions.append(start_predicted_classes)

        # Distribution that forces <end> once a beam has ended.
        log_probs_after_end = start_class_log_probabilities.new_full(
            (batch_size * self.beam_size, num_classes),
            min_value_of_dtype(start_class_log_probabilities.dtype))
        log_probs_after_end[:, self._end_index] = 0.0

        # Every state tensor gets one copy per beam element.
        self._update_initial_state(state, batch_size)

        for timestep in range(self.max_steps - 1):
            last_predictions = predictions[-1].reshape(batch_size *
                                                       self.beam_size)
            # Stop early when every beam of every batch element has ended.
            if (last_predictions == self._end_index).all():
                break
            class_log_probabilities, state = step(last_predictions, state,
                                                  timestep + 1)
            # This iteration generates token number `timestep + 2`.
            if timestep + 2 <= self.min_steps:
                class_log_probabilities[:, self._end_index] = \
                    min_value_of_dtype(class_log_probabilities.dtype)

            last_predictions_expanded = last_predictions.unsqueeze(-1).expand(
                batch_size * self.beam_size, num_classes)
            cleaned_log_probabilities = torch.where(
                last_predictions_expanded == self._end_index,
                log_probs_after_end,
                class_log_probabilities,
            )

            # (batch_size * beam_size, per_node_beam_size)
            (top_log_probabilities, predicted_classes,
             sampler_state) = self.sampler.sample_nodes(
                 cleaned_log_probabilities, self.per_node_beam_size,
                 sampler_state)

            expanded_last_log_probabilities = (
                last_log_probabilities.unsqueeze(2).expand(
                    batch_size, self.beam_size,
                    self.per_node_beam_size).reshape(
                        batch_size * self.beam_size, self.per_node_beam_size))
            summed_top_log_probabilities = (top_log_probabilities +
                                            expanded_last_log_probabilities)

            reshaped_summed = summed_top_log_probabilities.reshape(
                batch_size, self.beam_size * self.per_node_beam_size)
            reshaped_predicted_classes = predicted_classes.reshape(
                batch_size, self.beam_size * self.per_node_beam_size)

            (restricted_beam_log_probs, restricted_beam_indices,
             sampler_state) = self.sampler.sample_beams(
                 reshaped_summed, self.beam_size, sampler_state)
            restricted_predicted_classes = reshaped_predicted_classes.gather(
                1, restricted_beam_indices)

            predictions.append(restricted_predicted_classes)
            last_log_probabilities = restricted_beam_log_probs

            # Candidates with a common ancestor are adjacent, so the integer
            # quotient by per_node_beam_size is the ancestor.
            backpointer = torch.divide(restricted_beam_indices,
                                       self.per_node_beam_size,
                                       rounding_mode='trunc')
            backpointers.append(backpointer)

            # Keep only the state of the surviving ancestors.
            self._update_state(state, backpointer)

        if not torch.isfinite(last_log_probabilities).all():
            warnings.warn(
                'Negligible log probabilities encountered (-inf or < dtype '
                'min).', RuntimeWarning)

        reconstructed_predictions = self._reconstruct_sequences(
            predictions, backpointers)
        # (batch_size, beam_size, steps taken)
        all_predictions = torch.cat(list(reversed(reconstructed_predictions)),
                                    2)
        final_scores = self.final_sequence_scorer.score(all_predictions,
                                                        last_log_probabilities,
                                                        self._end_index)
        # Best sequence first.
        sorted_final_scores, sorted_indices = torch.sort(final_scores,
                                                         dim=1,
                                                         descending=True)
        sorted_all_predictions = torch.gather(
            all_predictions, 1,
            sorted_indices.unsqueeze(-1).expand_as(all_predictions))
        return sorted_all_predictions, sorted_final_scores

    @staticmethod
    def _is_multilayer_rnn_decoder(key: str,
                                   state_tensor: torch.Tensor) -> bool:
        return state_tensor.dim() == 3 and key in {
            'decoder_hidden', 'decoder_context'
        }

    def _update_initial_state(self, state: StateType, batch_size: int):
        for key, state_tensor in state.items():
            if state_tensor is None:
                continue
            multilayer_rnn_decoder = self._is_multilayer_rnn_decoder(
                key, state_tensor)
            if multilayer_rnn_decoder:
                # (num_layers, batch, *) layout, special-cased upstream for
                # these two key names only
                num_layers, _, *last_dims = state_tensor.size()
                state[key] = (state_tensor.unsqueeze(2).expand(
                    num_layers, batch_size, self.beam_size,
                    *last_dims).reshape(num_layers,
                                        batch_size * self.beam_size,
                                        *last_dims))
            else:
                _, *last_dims = state_tensor.size()
                state[key] = (state_tensor.unsqueeze(1).expand(
                    batch_size, self.beam_size,
                    *last_dims).reshape(batch_size * self.beam_size,
                                        *last_dims))

    def _update_state(self, state: StateType, backpointer: torch.Tensor):
        batch_size = backpointer.size()[0]
        for key, state_tensor in state.items():
            if state_tensor is None:
                continue
            multilayer_rnn_decoder = self._is_multilayer_rnn_decoder(
                key, state_tensor)
            if multilayer_rnn_decoder:
                num_layers, _, *last_dims = state_tensor.size()
                expanded_backpointer = backpointer.view(
                    batch_size, self.beam_size,
                    *([1] * len(last_dims))).expand(batch_size,
                                                    self.beam_size,
                                                    *last_dims)
                expanded_backpointer = expanded_backpointer.unsqueeze(
                    0).repeat(num_layers, 1, 1, 1)
                state[key] = (state_tensor.reshape(
                    num_layers, batch_size, self.beam_size,
                    *last_dims).gather(2, expanded_backpointer).reshape(
                        num_layers, batch_size * self.beam_size, *last_dims))
            else:
                _, *last_dims = state_tensor.size()
                expanded_backpointer = backpointer.view(
                    batch_size, self.beam_size,
                    *([1] * len(last_dims))).expand(batch_size,
                                                    self.beam_size,
                                                    *last_dims)
                state[key] = (state_tensor.reshape(
                    batch_size, self.beam_size,
                    *last_dims).gather(1, expanded_backpointer).reshape(
                        batch_size * self.beam_size, *last_dims))


# ---------------------------------------------------------------------------
# Hand-worked known-answer case ("G8"): V = 8, beam 3,
# max_steps 4, un-normalised integer-ish scores so every sum is exact and the
# whole search can be followed on paper.  Tokens 0..3 words, 4 <start>,
# 5 <stop>.  Scores depend on the last token only (row = last token).
#
# Case A (one batch element):
#   <start>: 0:-1  1:-2  2:-3                     -> beams [0] -1, [1] -2, [2] -3
#   step 1  row 0: 1:-1 stop:-2 3:-4 | row 1: 0:-.5 2:-3 stop:-5 |
#           row 2: stop:-.25 3:-1 0:-6
#     candidates (parent*3+j): -2 -3 -5 | -2.5 -5 -7 | -3.25 -4 -9
#     keep -2 (p0,tok1)  -2.5 (p1,tok0)  -3 (p0,stop)
#                                  -> [0,1] -2, [1,0] -2.5, [0,stop] -3
#   step 2  [0,1]: row 1 -> -2.5 -5 -7 | [1,0]: row 0 -> -3.5 -4.5 -6.5 |
#           [0,stop]: ended -> stop +0 = -3, the rest finfo.min
#     keep -2.5 (p0,tok0)  -3 (p2,stop)  -3.5 (p1,tok1)
#                         -> [0,1,0] -2.5, [0,stop,stop] -3, [1,0,1] -3.5
#   step 3  [0,1,0]: row 0 -> -3.5 -4.5 -6.5 | ended -> -3 |
#           [1,0,1]: row 1 -> -4 -6.5 -8.5
#     keep -3 (p1,stop)  -3.5 (p0,tok1)  -4 (p2,tok0)
#   result  [0,stop,stop,stop] -3 | [0,1,0,1] -3.5 | [1,0,1,0] -4
#
# Case B (early exit): <start>: stop:-1 0:-2 1:-3; rows 0 and 1: stop:-.5.
#   step 1 candidates: -1 (forced stop) | -2.5 (stop) .. | -3.5 (stop) ..
#   every beam now ends in <stop> -> the loop breaks: T' = 2
#   result  [stop,stop] -1 | [0,stop] -2.5 | [1,stop] -3.5
# ---------------------------------------------------------------------------
HAND_V, HAND_START, HAND_STOP, HAND_BEAM, HAND_STEPS = 8, 4, 5, 3, 4


def hand_table(case: str) -> torch.Tensor:
    t = torch.full((HAND_V, HAND_V), -9.0)
    S, E = HAND_START, HAND_STOP
    if case == 'A':
        t[S, 0], t[S, 1], t[S, 2] = -1., -2., -3.
        t[0, 1], t[0, E], t[0, 3] = -1., -2., -4.
        t[1, 0], t[1, 2], t[1, E] = -.5, -3., -5.
        t[2, E], t[2, 3], t[2, 0] = -.25, -1., -6.
    elif case == 'B':
        t[S, E], t[S, 0], t[S, 1] = -1., -2., -3.
        t[0, E] = t[1, E] = -.5
    else:
        raise KeyError(case)
    return t


HAND_EXPECTED = {
    'A': ([[0, 5, 5, 5], [0, 1, 0, 1], [1, 0, 1, 0]], [-3.0, -3.5, -4.0]),
    'B': ([[5, 5], [0, 5], [1, 5]], [-1.0, -2.5, -3.5]),
}


def run_hand_case(case: str):
    """The hand-worked case through this module's BeamSearch."""
    table = hand_table(case)

    def step(tokens: torch.Tensor, state: StateType):
        return table[tokens].clone(), state

    runner = BeamSearch(HAND_STOP, max_steps=HAND_STEPS, beam_size=HAND_BEAM)
    start = torch.full((1,), HAND_START, dtype=torch.long)
    tokens, scores = runner.search(start, {'dummy': torch.zeros(1, 2)}, step)
    return tokens[0].tolist(), scores[0].tolist()
