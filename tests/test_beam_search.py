"""Known answers of the reference's decoder tests (unit_tests/decoder_test.py:14-80) for the prefix beam
search decoder.  Host-only (no GPU)."""
import numpy as np
import pytest
import torch

from wav2letter_pytorch_amd.beam_search import PrefixBeamSearchLMDecoder, get_time_per_word, prefix_beam_search
from wav2letter_pytorch_amd.data.label_sets import english_labels


def _asr_sample():
    s = np.zeros((10, len(english_labels)))
    s[0, 2] = 0.5
    s[1, 20] = 0.5
    s[2, 19] = 0.5
    s[3:, 0] = 0.5
    return s


def test_sanity():                                   # decoder_test.py:14-21
    assert prefix_beam_search(_asr_sample(), english_labels) == 'ASR'


def test_inconsistent_sizes():                       # decoder_test.py:23-27
    with pytest.raises(AssertionError):
        prefix_beam_search(np.zeros((10, len(english_labels) - 1)), english_labels)


def test_beam_is_not_greedy():                       # decoder_test.py:30-42
    labels = ['_', 'A', 'B', ' ']
    samples = np.array([[0.8, 0.2, 0, 0], [0.6, 0.4, 0, 0]])
    assert prefix_beam_search(samples, labels, blank_index=0, return_weights=True) == ('A', 0.52)


def test_beam_width_changes():                       # decoder_test.py:44-59
    def the_lm(s):
        return 0.5 if s == 'A' else 1
    labels = ['_', 'A', ' ']
    samples = np.array([[0.8, 0.2, 0], [0.7, 0.3, 0], [0.6, 0.4, 0], [0.0, 0.0, 1]])
    assert prefix_beam_search(samples, labels, lm=the_lm, return_weights=False, k=25, alpha=1, beta=0) == ' '
    assert prefix_beam_search(samples, labels, lm=the_lm, return_weights=False, k=1, alpha=1, beta=0) == 'A '


def test_class_wrapper():                            # decoder_test.py:61-70
    assert PrefixBeamSearchLMDecoder('', english_labels).decode(_asr_sample()) == 'ASR'


def test_pbs_batch_dimensions():                     # decoder_test.py:72-80
    sample = torch.from_numpy(_asr_sample()).unsqueeze(0)
    res = PrefixBeamSearchLMDecoder('', english_labels).decode(sample, english_labels)
    assert res == ['ASR']


def test_time_per_word():
    words = get_time_per_word('hi  yo', [0, 1, 2, 3, 5, 9], ratio=0.5)
    assert words == [('hi', 0.0, 0.5), ('yo', 2.5, 4.5)]
