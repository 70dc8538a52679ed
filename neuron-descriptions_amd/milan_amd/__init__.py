"""MI355X-native MILAN inference path (drop-in for `src.milan`).

    from milan_amd import pretrained, Decoder
    decoder = pretrained('base').to('cuda')
    captions = decoder.predict(dataset, strategy='rerank', temperature=.2,
                               beam_size=50, device='cuda')
"""
from milan_amd.decoders import (STRATEGIES, STRATEGY_BEAM, STRATEGY_GREEDY,
                                STRATEGY_RERANK, STRATEGY_SAMPLE, Decoder,
                                DecoderOutput, DecoderState, DecoderStep)
from milan_amd.encoders import Encoder, PyramidConvEncoder, encoder
from milan_amd.lms import LanguageModel
from milan_amd.loaders import pretrained, pretrained_sharded

__all__ = [
    'Decoder', 'DecoderOutput', 'DecoderState', 'DecoderStep', 'Encoder',
    'PyramidConvEncoder', 'LanguageModel', 'encoder', 'pretrained',
    'pretrained_sharded',
    'STRATEGIES', 'STRATEGY_BEAM', 'STRATEGY_GREEDY', 'STRATEGY_RERANK',
    'STRATEGY_SAMPLE'
]
