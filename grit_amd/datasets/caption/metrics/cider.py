"""CIDEr-D, the reward of self-critical training (reference datasets/caption/metrics/cider/{cider,cider_scorer}.py, used at
engine/caption_engine.py:433-438 and train_caption.py:78).

Same interface: `Cider(gts)` fixes the document frequencies and the (log) corpus size from the tokenised training captions;
`compute_score(gts, res)` returns (mean, per-image scores) for hypotheses `res[k][0]` against references `gts[k]`.
Same definition (Vedantam et al., arXiv:1411.5726, with the "-D" clipping and Gaussian length penalty): for n = 1..4 the
tf-idf vectors of hypothesis and reference n-grams, clipped cosine similarity, exp(-(len_h - len_r)^2 / (2 sigma^2)),
mean over n, mean over references, x 10.

Arranged differently from the reference: every sentence is reduced ONCE to four sparse tf-idf vectors (the reference rebuilds
the hypothesis vector per image and the reference vectors per (image, reference) pair from defaultdicts), and the per-step call
only touches the hypotheses -- the references of a training batch can be pre-cooked with `cook()` and passed in cooked form."""
import math
from collections import Counter

import numpy as np


def ngram_counts(sentence, n=4):
    words = sentence.split()
    counts = Counter()
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return counts


class _Vec(object):
    """tf-idf vectors of one sentence: weights[n] = {ngram: w}, norm[n], and the sentence length in bigrams' terms."""
    __slots__ = ('weights', 'norm', 'length')

    def __init__(self, counts, doc_frequency, log_corpus, n):
        self.weights = [dict() for _ in range(n)]
        sq = [0.0] * n
        self.length = 0
        for ngram, tf in counts.items():
            df = math.log(max(1.0, doc_frequency.get(ngram, 0.0)))  # unseen n-grams count as document frequency 1
            k = len(ngram) - 1
            w = float(tf) * (log_corpus - df)
            self.weights[k][ngram] = w
            sq[k] += w * w
            if k == 1:  # the reference measures sentence length in bigram occurrences (cider_scorer.py:107-108)
                self.length += tf
        self.norm = [math.sqrt(v) for v in sq]


class Cider(object):

    def __init__(self, gts=None, n=4, sigma=6.0):
        self._n, self._sigma = n, sigma
        self.doc_frequency, self.ref_len = None, None
        if gts is not None:
            self.doc_frequency, self.ref_len = self._corpus_statistics(gts)

    def _corpus_statistics(self, gts):
        df = {}
        for refs in gts.values():
            seen = set()
            for ref in refs:
                seen.update(ngram_counts(ref, self._n).keys())
            for ngram in seen:
                df[ngram] = df.get(ngram, 0.0) + 1.0
        return df, math.log(float(len(gts)))

    def cook(self, sentences, doc_frequency=None, ref_len=None):
        """Sentences -> reusable tf-idf vectors (under this object's corpus statistics unless others are given)."""
        df = self.doc_frequency if doc_frequency is None else doc_frequency
        rl = self.ref_len if ref_len is None else ref_len
        return [_Vec(ngram_counts(s, self._n), df, rl, self._n) for s in sentences]

    def _similarity(self, hyp, ref):
        delta = float(hyp.length - ref.length)
        penalty = math.e ** (-(delta ** 2) / (2 * self._sigma ** 2))
        total = 0.0
        for k in range(self._n):
            wr = ref.weights[k]
            val = 0.0
            for ngram, wh in hyp.weights[k].items():
                r = wr.get(ngram)
                if r is not None:
                    val += min(wh, r) * r
            if hyp.norm[k] != 0 and ref.norm[k] != 0:
                val /= hyp.norm[k] * ref.norm[k]
            total += val * penalty
        return total / self._n

    def compute_score(self, gts, res):
        """gts[k] = list of reference strings (or cooked vectors), res[k] = [hypothesis string]."""
        assert gts.keys() == res.keys()
        df, rl = self.doc_frequency, self.ref_len
        if df is None:  # no training corpus given: statistics of this very call, as the reference does
            df, rl = self._corpus_statistics(gts)
        scores = []
        for k in gts.keys():
            refs = gts[k]
            refs = refs if (refs and isinstance(refs[0], _Vec)) else self.cook(refs, df, rl)
            hyp = self.cook([res[k][0]], df, rl)[0]
            s = sum(self._similarity(hyp, r) for r in refs)
            scores.append(s / len(refs) * 10.0)
        scores = np.array(scores)
        return float(np.mean(scores)), scores

    def __str__(self):
        return 'CIDEr'
