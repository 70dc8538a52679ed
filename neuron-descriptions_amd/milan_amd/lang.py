"""Output side of the path: token ids -> caption strings.

Mirrors the inference-relevant half of the reference's `src/utils/lang.py`:
`Vocab` (:94-172), `Indexer` special ids (:231-260), `unindex` (:573-612) and
`reconstruct` (:678-730), plus `Indexer.index` / `__call__` (:331-514) for
`Decoder.score`.  The tokenizer itself (spaCy) is not in this image: `__call__`
takes any callable (or pre-tokenized captions) and the serialized tokenizer
payload is carried opaquely so checkpoints round-trip.

`reconstruct` is called on every (neuron, beam) sequence by the reference
(`src/milan/decoders.py:486-487`): 4096 x 50 Python calls per run.  Here ids
are mapped through a pre-built table and identical sequences are cached, and
`LazyCaptions` defers the per-beam strings until somebody reads them.
"""
import dataclasses
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple, Union

START_TOKEN = '<start>'
STOP_TOKEN = '<stop>'
PAD_TOKEN = '<pad>'
UNK_TOKEN = '<unk>'


@dataclasses.dataclass(frozen=True)
class Vocab:
    """Ordered token list (reference `lang.Vocab`)."""
    tokens: Tuple[str, ...]

    def __post_init__(self):
        object.__setattr__(self, 'tokens', tuple(self.tokens))
        object.__setattr__(self, 'ids',
                           {t: i for i, t in enumerate(self.tokens)})

    def __getitem__(self, token):
        if isinstance(token, (int, slice)):
            return self.tokens[token]
        return self.ids[token]

    def __len__(self) -> int:
        return len(self.tokens)

    def __contains__(self, token) -> bool:
        if isinstance(token, int):
            return 0 <= token < len(self)
        return token in self.ids

    @property
    def unique(self):
        return frozenset(self.ids)

    def properties(self) -> Mapping[str, Any]:
        return {'tokens': self.tokens}


@dataclasses.dataclass(frozen=True)
class Indexer:
    """Maps id sequences back to text (reference `lang.Indexer`).

    `tokenize` is whatever the checkpoint stored (opaque here).
    """
    vocab: Vocab
    tokenize: Any = None
    start: bool = False
    stop: bool = False
    pad: bool = False
    unk: bool = False
    length: Optional[int] = None

    @property
    def start_index(self) -> int:
        return len(self.vocab)

    @property
    def stop_index(self) -> int:
        return len(self.vocab) + 1

    @property
    def pad_index(self) -> int:
        return len(self.vocab) + 2

    @property
    def unk_index(self) -> int:
        return len(self.vocab) + 3

    @property
    def specials(self) -> Mapping[int, str]:
        return {
            self.start_index: START_TOKEN,
            self.stop_index: STOP_TOKEN,
            self.pad_index: PAD_TOKEN,
            self.unk_index: UNK_TOKEN,
        }

    @property
    def tokens(self) -> Tuple[str, ...]:
        return tuple(self.vocab.tokens) + tuple(self.specials.values())

    @property
    def ids(self) -> Mapping[str, int]:
        ids = dict(self.vocab.ids)
        for index, token in self.specials.items():
            ids[token] = index
        return ids

    @property
    def unique(self):
        return frozenset(self.ids)

    def __len__(self) -> int:
        return len(self.vocab) + 4

    def __getitem__(self, token):
        if isinstance(token, int):
            return self.tokens[token]
        if isinstance(token, slice):
            return self.tokens[token]
        return self.ids[token]

    def __contains__(self, token) -> bool:
        if isinstance(token, int):
            return 0 <= token < len(self)
        return token in self.ids

    # -- tokens -> ids (reference lang.py:331-514) ------------------------------
    def __call__(self, texts, **kwargs):
        """Tokenize then index.  `tokenize` is the spaCy-backed `Tokenizer` in
        the reference; any callable str|[str] -> tokens|[tokens] works here."""
        if not callable(self.tokenize):
            raise NotImplementedError(
                'text -> ids indexing needs a tokenizer: the checkpoint\'s spaCy '
                'tokenizer is not available in this build; set '
                '`indexer.tokenize` to a callable or pass pre-tokenized '
                'sequences to `Indexer.index`')
        return self.index(self.tokenize(texts), **kwargs)

    def index(self,
              tokenized,
              start: Optional[bool] = None,
              stop: Optional[bool] = None,
              pad: Optional[bool] = None,
              unk: Optional[bool] = None,
              length: Optional[int] = None):
        """Map token strings to ids; one sequence or a batch of sequences.

        `length` does not count the start/stop tokens; sequences are truncated
        (keeping room for `<stop>`) and, with `pad`, padded to it.  Unknown
        tokens become `<unk>` when `unk`, otherwise they are dropped.
        """
        if not tokenized:
            return ()
        singleton = isinstance(tokenized[0], str)
        start = self.start if start is None else start
        stop = self.stop if stop is None else stop
        pad = self.pad if pad is None else pad
        unk = self.unk if unk is None else unk
        batch = [tokenized] if singleton else tokenized
        length = length or self.length or max(len(toks) for toks in tokenized)
        length += int(bool(start)) + int(bool(stop))
        ids = self.vocab.ids
        indexed = []
        for tokens in batch:
            row = [self.start_index] if start else []
            if unk:
                row += [ids.get(tok, self.unk_index) for tok in tokens]
            else:
                row += [ids[tok] for tok in tokens if tok in ids]
            if stop:
                del row[max(length - 1, 0):]
                row.append(self.stop_index)
            if len(row) < length and pad:
                row += [self.pad_index] * (length - len(row))
            del row[length:]
            indexed.append(tuple(row))
        return indexed[0] if singleton else tuple(indexed)

    # -- ids -> tokens ------------------------------------------------------
    def unindex(self,
                indexed,
                specials: bool = True,
                start: bool = True,
                stop: bool = True,
                pad: bool = True,
                unk: bool = True):
        """reference lang.py:573-612."""
        if not indexed:
            return ()
        singleton = isinstance(indexed[0], int)
        keep = dict(zip(self.specials, (start, stop, pad, unk)))
        names = self.specials
        n = len(self.vocab)
        out = []
        for indices in ([indexed] if singleton else indexed):
            toks = []
            for index in indices:
                if index < n:
                    toks.append(self.vocab.tokens[index])
                elif index in names:
                    if specials and keep[index]:
                        toks.append(names[index])
                else:
                    raise ValueError(f'unknown index: {index}')
            out.append(tuple(toks))
        return out[0] if singleton else tuple(out)

    # -- ids / tokens -> caption --------------------------------------------
    def _text(self, tokens: Sequence[str]) -> str:
        """reference lang.py:703-727."""
        if STOP_TOKEN in tokens:
            tokens = tokens[:tokens.index(STOP_TOKEN)]
        special = (START_TOKEN, STOP_TOKEN, PAD_TOKEN, UNK_TOKEN)
        text = ' '.join(t for t in tokens if t not in special)
        for p in ('.', ',', ';', ':'):
            text = text.replace(' ' + p, p)
        text = text.replace(' -', '-').replace('- ', '-')
        return '. '.join(s.strip().capitalize()
                         for s in text.split('.')).strip()

    def reconstruct(self, inputs) -> Union[str, Tuple[str, ...]]:
        """reference lang.py:678-730 (all four overloads)."""
        if not inputs:
            raise ValueError('must provide at least one seq')
        for index, item in enumerate(inputs):
            if not isinstance(item, (int, str)) and not item:
                raise ValueError(f'input seq {index} is empty')
        first = inputs[0]
        if isinstance(first, str):
            return self._text(tuple(inputs))
        if isinstance(first, int):
            return self._text(self.unindex(list(inputs)))
        if isinstance(first[0], str):
            return tuple(self._text(tuple(t)) for t in inputs)
        cache: Dict[Tuple[int, ...], str] = {}
        out = []
        for seq in inputs:
            key = tuple(seq)
            text = cache.get(key)
            if text is None:
                text = cache[key] = self._text(self.unindex(list(key)))
            out.append(text)
        return tuple(out)

    def properties(self) -> Mapping[str, Any]:
        return {
            'vocab': self.vocab,
            'tokenize': self.tokenize,
            'start': self.start,
            'stop': self.stop,
            'pad': self.pad,
            'unk': self.unk,
            'length': self.length,
        }


class LazyCaptions(Sequence):
    """`beam_captions`: a (B,) sequence of (beam,) string tuples, built on
    first access instead of eagerly for all B x beam sequences."""

    def __init__(self, indexer: Indexer, beam_tokens):
        self._indexer = indexer
        self._tokens = beam_tokens  # (B, beam, T) tensor
        self._cache: Optional[Tuple[Tuple[str, ...], ...]] = None

    def _all(self):
        if self._cache is None:
            rows = self._tokens.tolist()
            self._cache = tuple(
                self._indexer.reconstruct(beams) for beams in rows)
        return self._cache

    def __len__(self) -> int:
        return int(self._tokens.shape[0])

    def __getitem__(self, i):
        return self._all()[i]

    def __iter__(self):
        return iter(self._all())

    def __eq__(self, other):
        return tuple(self._all()) == tuple(other)

    def __repr__(self):
        return f'LazyCaptions({len(self)} x {self._tokens.shape[1]})'


def join(texts: Any, delimiter: str = ' ') -> str:
    """A string, or an iterable of strings joined by `delimiter` (sets in
    sorted order); anything else is a ValueError (reference lang.py:781-800)."""
    if isinstance(texts, (set, frozenset)):
        texts = tuple(sorted(texts))
    if isinstance(texts, (list, tuple)):
        texts = delimiter.join(texts)
    if not isinstance(texts, str):
        raise ValueError(f'unknown annotation type: {type(texts).__name__}')
    return texts
