"""The known-answer tables of the reference's own language-utility tests
(tests/utils/lang_test.py: Vocab, Indexer.__call__/index/unindex, join) run
against `milan_amd.lang`.  The cases live in
tests/golden/reference_lang_cases.json as data; the spaCy tokenizer of the
reference fixture is replaced by a whitespace split, which tokenizes these
particular texts identically."""
import json

import pytest

from milan_amd import lang
from tests.conftest import GOLDEN_DIR

with open(GOLDEN_DIR / 'reference_lang_cases.json') as _f:
    CASES = json.load(_f)
TOKENS = tuple(CASES['tokens'])


def tup(x):
    return tuple(tup(i) for i in x) if isinstance(x, list) else x


def tokenize(texts):
    if isinstance(texts, str):
        return tuple(texts.split())
    return tuple(tuple(t.split()) for t in texts)


@pytest.fixture
def vocab():
    return lang.Vocab(TOKENS)


@pytest.fixture
def indexer(vocab):
    return lang.Indexer(vocab, tokenize)


def test_vocab_basics(vocab):
    assert vocab[1] == 'bar' and vocab['bar'] == 1
    assert vocab[slice(0, 2)] == ('foo', 'bar')
    assert len(vocab) == len(TOKENS)
    assert vocab.ids == {'foo': 0, 'bar': 1, 'baz': 2}
    assert vocab.unique == frozenset(TOKENS)
    for token, expected in CASES['vocab_contains']:
        assert (token in vocab) is expected, token


def test_indexer_special_indices_and_views(indexer):
    assert indexer.start_index == CASES['start_index']
    assert indexer.stop_index == CASES['stop_index']
    assert indexer.pad_index == CASES['pad_index']
    assert indexer.unk_index == CASES['unk_index']
    assert dict(indexer.specials) == {3: lang.START_TOKEN, 4: lang.STOP_TOKEN,
                                      5: lang.PAD_TOKEN, 6: lang.UNK_TOKEN}
    assert list(indexer.specials) == [3, 4, 5, 6]
    assert indexer.tokens == (*TOKENS, lang.START_TOKEN, lang.STOP_TOKEN,
                              lang.PAD_TOKEN, lang.UNK_TOKEN)
    assert indexer.ids == {'foo': 0, 'bar': 1, 'baz': 2, lang.START_TOKEN: 3,
                           lang.STOP_TOKEN: 4, lang.PAD_TOKEN: 5,
                           lang.UNK_TOKEN: 6}
    assert indexer.unique == set(TOKENS) | {lang.START_TOKEN, lang.STOP_TOKEN,
                                            lang.PAD_TOKEN, lang.UNK_TOKEN}
    assert indexer['bar'] == 1 and indexer[1] == 'bar'
    assert indexer[slice(0, 2)] == ('foo', 'bar')
    assert indexer[lang.START_TOKEN] == 3 and indexer[5] == lang.PAD_TOKEN
    assert len(indexer) == len(TOKENS) + 4
    for token, expected in CASES['indexer_contains']:
        assert (token in indexer) is expected, token


@pytest.mark.parametrize('case', CASES['indexer_call'],
                         ids=lambda c: json.dumps([c['init'], c['call']]))
def test_indexer_call(vocab, case):
    indexer = lang.Indexer(vocab, tokenize, **case['init'])
    texts = case['texts']
    texts = texts if isinstance(texts, str) else tuple(texts)
    assert indexer(texts, **case['call']) == tup(case['expected'])


def test_indexer_index_empty(indexer):
    assert indexer.index(()) == ()


@pytest.mark.parametrize('case', CASES['indexer_unindex'],
                         ids=lambda c: json.dumps(c['kwargs']))
def test_indexer_unindex(indexer, case):
    actual = indexer.unindex(tup(case['indexed']), **case['kwargs'])
    assert actual == tup(case['expected'])


def test_indexer_unindex_empty_and_bad_index(indexer):
    assert indexer.unindex(()) == ()
    bad = CASES['unindex_bad_index']
    with pytest.raises(ValueError, match=f'.*{bad}.*'):
        indexer.unindex((0, bad))


@pytest.mark.parametrize('case', CASES['join'], ids=lambda c: c['kind'])
def test_join(case):
    make = {'list': list, 'tuple': tuple, 'set': set,
            'frozenset': frozenset, 'str': str}[case['kind']]
    assert lang.join(make(case['texts'])) == case['expected']


def test_join_bad_input():
    with pytest.raises(ValueError, match='.*dict.*'):
        lang.join({'foo': 'bar'})
