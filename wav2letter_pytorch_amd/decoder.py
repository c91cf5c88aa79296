"""Greedy CTC decoder + CER/WER (reference: decoder.py:11-145).

The per-frame argmax (torch.max(probs, 2), decoder.py:136) runs as a HIP kernel
(w2l_argmax: ties -> lowest index); ONE device-to-host copy of the int32 index matrix
replaces the reference's per-frame ``.item()`` calls (decoder.py:108,111); repeat/blank
collapsing is vectorised on the host.  Edit distance is the C routine
w2l_levenshtein_host (python-Levenshtein is not a dependency).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .data import label_sets


def _edit_distance(a, b) -> int:
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    return int(lib.w2l_levenshtein_host(a.ctypes.data_as(C.c_void_p), len(a), b.ctypes.data_as(C.c_void_p), len(b)))


class Decoder(object):
    """Base decoder (decoder.py:11-83): label bookkeeping + WER/CER helpers."""

    def __init__(self, labels, blank_index=0):
        """``labels``: the label sequence itself or the name of one in data.label_sets; index -> character lookups are built
        once.  A label set without a space gets ``space_index = len(labels)``, an index no frame can take (decoder.py:22-29)."""
        alphabet = label_sets.labels_map[labels] if isinstance(labels, str) else labels
        self.labels = alphabet
        self.blank_index = blank_index
        self.int_to_char = {index: char for index, char in enumerate(alphabet)}
        self.space_index = next((index for index, char in enumerate(alphabet) if char == ' '), len(alphabet))

    def wer(self, s1, s2):
        """word-level edit distance (decoder.py:31-49)"""
        vocab = {}
        w1 = [vocab.setdefault(w, len(vocab)) for w in s1.split()]
        w2 = [vocab.setdefault(w, len(vocab)) for w in s2.split()]
        return _edit_distance(w1, w2)

    def cer(self, s1, s2):
        """character-level edit distance with spaces removed (decoder.py:51-60)"""
        s1, s2 = s1.replace(' ', ''), s2.replace(' ', '')
        return _edit_distance([ord(c) for c in s1], [ord(c) for c in s2])

    def cer_ratio(self, expected, predicted):
        return self.cer(expected, predicted), len(expected.replace(' ', ''))

    def wer_ratio(self, expected, predicted):
        return self.wer(expected, predicted), len(expected.split())

    def decode(self, probs, sizes=None):
        raise NotImplementedError


def argmax_indices(probs: torch.Tensor) -> torch.Tensor:
    """int32 [N, T] argmax over the label axis on the device (first maximal index)."""
    if not probs.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.W2LError('GreedyDecoder needs the MI355X device (no CPU fallback)')
        probs = probs.cuda()
    p = probs.detach().contiguous().float()
    n, t, c = p.shape
    idx = torch.empty(n, t, dtype=torch.int32, device=p.device)
    check(lib.w2l_argmax(ptr(p), n * t, c, ptr(idx), stream_ptr()), 'w2l_argmax')
    return idx


class GreedyDecoder(Decoder):
    def __init__(self, labels, blank_index=0):
        super(GreedyDecoder, self).__init__(labels, blank_index)

    def process_string(self, sequence, size, remove_repetitions=False):
        """decoder.py:104-119 for one index sequence already on the host."""
        seq = np.asarray(sequence[:size].cpu() if torch.is_tensor(sequence) else sequence[:size]).astype(np.int64)
        keep = seq != self.blank_index
        if remove_repetitions and len(seq) > 1:
            keep[1:] &= seq[1:] != seq[:-1]
        offsets = np.nonzero(keep)[0]
        chars = []
        for i in seq[keep]:
            chars.append(' ' if i == self.space_index else self.int_to_char[int(i)])
        return ''.join(chars), torch.IntTensor(offsets.astype(np.int32))

    def convert_to_strings(self, sequences, sizes=None, remove_repetitions=False, return_offsets=False):
        """index rows -> text, one single-element list per utterance (the nesting beam-search decoders fill with n-best lists;
        decoder.py:89-102); ``sizes[n]`` limits row n to its valid frames"""
        rows = sequences.cpu().numpy() if torch.is_tensor(sequences) else np.asarray(sequences)
        decoded = [self.process_string(row, len(row) if sizes is None else int(sizes[n]), remove_repetitions)
                   for n, row in enumerate(rows)]
        texts = [[text] for text, _ in decoded]
        if not return_offsets:
            return texts
        return texts, [[frames] for _, frames in decoded]

    def score_batch(self, indices, sizes, texts):
        """What ConvCTCASR.add_string_metrics needs from one batch (base_asr_models.py:53-69), from argmax indices that are
        already on the host: (decoded strings, (cer_err, cer_ref, wer_err, wer_ref)) -- collapse, CER and WER totals in ONE
        C call (w2l_greedy_score_host) that holds no interpreter lock, instead of ~100 Python-level calls per batch.
        ``indices``: int32 [N, T] host tensor / array; ``sizes``: [N] valid frames or None; ``texts``: the N transcripts."""
        idx = np.ascontiguousarray(indices.numpy() if torch.is_tensor(indices) else indices, dtype=np.int32)
        n, t = idx.shape
        lut = self.__dict__.get('_cp_lut')
        if lut is None:
            chars = [self.int_to_char[i] for i in range(len(self.int_to_char))]
            # (labels of one code point each -- both shipped label sets; anything else takes the string path below)
            lut = self.__dict__['_cp_lut'] = (np.array([ord(c) for c in chars], dtype=np.int32)
                                              if all(isinstance(c, str) and len(c) == 1 for c in chars) else False)
        if lut is False or type(self).cer is not Decoder.cer or type(self).wer is not Decoder.wer:
            hyps = [h[0] for h in self.convert_to_strings(idx, sizes, remove_repetitions=True)]
            cer = [self.cer_ratio(ref, hyp) for ref, hyp in zip(texts, hyps)]
            wer = [self.wer_ratio(ref, hyp) for ref, hyp in zip(texts, hyps)]
            return hyps, (sum(e for e, _ in cer), sum(d for _, d in cer), sum(e for e, _ in wer), sum(d for _, d in wer))
        sz = None if sizes is None else np.ascontiguousarray(sizes.numpy() if torch.is_tensor(sizes) else sizes, dtype=np.int32)
        # (utf-32 with surrogatepass: one int32 per code point, whatever the transcript holds)
        ref = np.frombuffer(''.join(texts).encode('utf-32-le', 'surrogatepass'), dtype=np.int32)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(x) for x in texts], out=off[1:])
        totals = np.zeros(5, dtype=np.int64)
        hyp = np.empty(max(n * t, 1), dtype=np.int32)
        hoff = np.zeros(n + 1, dtype=np.int64)
        vp = C.c_void_p
        check(lib.w2l_greedy_score_host(idx.ctypes.data_as(vp), n, t, sz.ctypes.data_as(vp) if sz is not None else None,
                                        int(self.blank_index), lut.ctypes.data_as(vp), len(lut), ref.ctypes.data_as(vp),
                                        off.ctypes.data_as(vp), totals.ctypes.data_as(vp), hyp.ctypes.data_as(vp),
                                        hoff.ctypes.data_as(vp)), 'w2l_greedy_score_host')
        text = hyp[: hoff[n]].tobytes().decode('utf-32-le', 'surrogatepass')
        hyps = [text[hoff[i]: hoff[i + 1]] for i in range(n)]
        return hyps, tuple(int(v) for v in totals[:4])

    def decode(self, probs, sizes=None, return_offsets=False):
        """argmax decoding, repeats and blanks removed (decoder.py:121-145).
        probs: [batch, seq_length, output_dim] (or 2-D for one utterance)."""
        if len(probs.shape) == 2:
            return self.decode(probs.unsqueeze(0), sizes, return_offsets)
        idx = argmax_indices(probs)
        if sizes is not None and torch.is_tensor(sizes):
            sizes = sizes.detach().cpu()
        host = idx.cpu()                      # the one D2H copy of the step
        strings, offsets = self.convert_to_strings(host, sizes, remove_repetitions=True, return_offsets=True)
        strings = [s[0] for s in strings]
        if return_offsets:
            return strings, offsets
        return strings
