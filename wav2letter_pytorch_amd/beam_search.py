"""CTC prefix beam search with an optional language model (reference: decoder.py:147-267, itself after
Hannun et al. 2014 / corticph prefix-beam-search) -- an inference-side, host-only decoder.

Each hypothesis (prefix string) carries two masses per time step: ending in blank (``pb``) and ending in a
non-blank (``pnb``).  The reference's observable behaviour that callers rely on is kept: probabilities (not
logs) in, the first index of a label wins on duplicates, an ``end_char`` freezes a prefix, the LM is applied
when a word is closed (space or end char) with weight ``alpha``, ranking multiplies by (#words+1)**beta, only
strictly positive masses survive a step, ties keep first-seen order."""
from __future__ import annotations

import re
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .decoder import Decoder

_WORD_RE = re.compile(r'\w+[\s|>]')


def _n_words(prefix: str) -> int:
    return len(_WORD_RE.findall(prefix))


class _Step:
    """masses of one time step, remembering first-insertion order of prefixes (pb first, then pnb)"""

    def __init__(self):
        self.pb: Dict[str, float] = {}
        self.pnb: Dict[str, float] = {}

    def add_b(self, prefix, v):
        self.pb[prefix] = self.pb.get(prefix, 0.0) + v

    def add_nb(self, prefix, v):
        self.pnb[prefix] = self.pnb.get(prefix, 0.0) + v

    def total(self) -> Dict[str, float]:
        """pb + pnb keeping only positive sums (collections.Counter addition semantics)"""
        out: Dict[str, float] = {}
        for src in (self.pb, self.pnb):
            for k in src:
                if k not in out:
                    s = self.pb.get(k, 0.0) + self.pnb.get(k, 0.0)
                    if s > 0:
                        out[k] = s
        return out


def prefix_beam_search(ctc, labels: Sequence[str], blank_index: int = 0, lm: Optional[Callable[[str], float]] = None,
                       k: int = 5, alpha: float = 0.3, beta: float = 5, prune: float = 0.001, end_char: str = '>',
                       return_weights: bool = False):
    """ctc: [timesteps, alphabet] probabilities.  Returns the best prefix (and its ranking weight)."""
    ctc = np.asarray(ctc)
    assert ctc.shape[1] == len(labels), "ctc size:%d, labels: %d" % (ctc.shape[1], len(labels))
    assert ctc.shape[0] > 1, "ctc length: %d was too short" % ctc.shape[0]
    assert (ctc >= 0).all(), 'ctc output contains negative numbers'
    score_lm = lm if lm is not None else (lambda _s: 1)
    labels = list(labels)
    first_index = {}
    for i, ch in enumerate(labels):
        first_index.setdefault(ch, i)
    blank = labels[blank_index]

    prev = _Step()
    prev.pb[''] = 1.0
    prev.pnb[''] = 0.0
    beam: List[str] = ['']
    ranked: Dict[str, float] = {}
    for t in range(ctc.shape[0]):
        frame = ctc[t]
        cur = _Step()
        alphabet = [labels[i] for i in np.where(frame > prune)[0]]
        in_beam = set(beam)
        for pre in beam:
            pb_prev, pnb_prev = prev.pb.get(pre, 0.0), prev.pnb.get(pre, 0.0)
            if pre and pre[-1] == end_char:             # a closed hypothesis is carried unchanged
                cur.pb[pre] = pb_prev
                cur.pnb[pre] = pnb_prev
                continue
            for ch in alphabet:
                p = frame[first_index[ch]]
                if ch == blank:
                    cur.add_b(pre, frame[blank_index] * (pb_prev + pnb_prev))
                    continue
                ext = pre + ch
                if pre and ch == pre[-1]:                # repeated character: only a blank separates two copies
                    cur.add_nb(ext, p * pb_prev)
                    cur.add_nb(pre, p * pnb_prev)
                elif pre.replace(' ', '') and ch in (' ', end_char):   # a word closes: language-model weight
                    w = score_lm(ext.strip(' ' + end_char)) ** alpha
                    cur.add_nb(ext, w * p * (pb_prev + pnb_prev))
                else:
                    cur.add_nb(ext, p * (pb_prev + pnb_prev))
                if ext not in in_beam:                   # mass of an extension that fell off the beam earlier
                    cur.add_b(ext, frame[blank_index] * (prev.pb.get(ext, 0.0) + prev.pnb.get(ext, 0.0)))
                    cur.add_nb(ext, p * prev.pnb.get(ext, 0.0))
        ranked = cur.total()
        order = sorted(ranked, key=lambda s: ranked[s] * (_n_words(s) + 1) ** beta, reverse=True)   # stable
        beam = order[:k]
        prev = cur
    if not beam:
        beam = ['']
    best = beam[0]
    if return_weights:
        return best, ranked.get(best, 0) * (_n_words(best) + 1) ** beta
    return best


class PrefixBeamSearchLMDecoder(Decoder):
    """decoder.py:232-267: wraps prefix_beam_search; ``lm_path`` selects a kenlm model (optional dependency)."""

    def __init__(self, lm_path, labels, blank_index=0, k=5, alpha=0.3, beta=5, prune=1e-3):
        super(PrefixBeamSearchLMDecoder, self).__init__(labels, blank_index)
        if lm_path:
            import kenlm  # noqa: F401  (not shipped with this repo; raises ImportError if absent)
            self.lm = kenlm.Model(lm_path)
            self.lm_weigh = lambda f: 10 ** (self.lm.score(f))
        else:
            self.lm_weigh = lambda s: 1
        self.k, self.alpha, self.beta, self.prune = k, alpha, beta, prune

    def decode(self, probs, sizes=None, return_offsets=False):
        if return_offsets:
            raise NotImplementedError("Prefix beam search does not support offsets (yet).")
        if hasattr(probs, 'detach'):
            probs = probs.detach().cpu().numpy()
        if len(probs.shape) == 2:
            return prefix_beam_search(probs, self.labels, self.blank_index, self.lm_weigh, self.k, self.alpha, self.beta,
                                      self.prune)
        if len(probs.shape) == 3:
            return [self.decode(p) for p in probs]
        raise RuntimeError('Decoding with wrong shape: %s, expected either [Batch X Frames X Labels] or '
                           '[Frames X Labels]' % str(probs.shape))


def get_time_per_word(predictions, offsets, ratio=1.0):
    """(word, start, end) triples from per-character offsets (decoder.py:270-302); the end time of a word is the
    first frame of its last character."""
    assert len(predictions) == len(offsets)
    words = []
    cur, start, end = '', -1, -1
    for ch, off in zip(predictions, offsets):
        if ch == ' ':
            if cur:
                words.append((cur, start, end))
                cur, start, end = '', -1, -1
            continue
        if cur:
            end = off * ratio
            cur += ch
        else:
            start = end = off * ratio
            cur = ch
    if cur:
        words.append((cur, start, end))
    return words
