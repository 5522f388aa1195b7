"""Host-side caption metrics of the validation loop.

corpus_bleu: BLEU-4 as the reference calls it (`nltk.translate.bleu_score.corpus_bleu(references,
hypotheses)` with default weights, stylenet/train_multitask.py:341). nltk is not installable here,
so this is a restatement of the published definition (Papineni et al. 2002: clipped n-gram
precisions pooled over the corpus, uniform weights over n = 1..4, brevity penalty against the
closest reference length). Pinned to nltk only through the worked example of nltk's own docstring
(corpus 0.5920..., sentence 0.5045... / 0.7400...; tests/test_oracle_cpu.py) plus hand-computed
corpora; anything that example does not exercise is parity unpinned. Where nltk's default
smoothing substitutes `sys.float_info.min` for an empty n-gram match count, this returns 0.0.
Token lists are lists of ints; pure Python, runs on the host like the reference's.
"""
import math
from collections import Counter


def _ngrams(tokens, n):
    return Counter(tuple(tokens[i:i + n]) for i in range(len(tokens) - n + 1))


def _closest_ref_length(refs, hyp_len):
    return min((len(r) for r in refs), key=lambda rl: (abs(rl - hyp_len), rl))


def corpus_bleu(list_of_references, hypotheses, weights=(0.25, 0.25, 0.25, 0.25)):
    assert len(list_of_references) == len(hypotheses), \
        "The number of hypotheses and their reference(s) should be the same"
    num = [0] * len(weights)
    den = [0] * len(weights)
    hyp_lengths, ref_lengths = 0, 0
    for refs, hyp in zip(list_of_references, hypotheses):
        for i in range(len(weights)):
            counts = _ngrams(hyp, i + 1)
            max_counts = Counter()
            for r in refs:
                rc = _ngrams(r, i + 1)
                for g in counts:
                    max_counts[g] = max(max_counts[g], rc[g])
            num[i] += sum(min(c, max_counts[g]) for g, c in counts.items())
            den[i] += max(1, sum(counts.values()))
        hyp_lengths += len(hyp)
        ref_lengths += _closest_ref_length(refs, len(hyp))
    if hyp_lengths == 0 or num[0] == 0:
        return 0.0
    bp = 1.0 if hyp_lengths > ref_lengths else math.exp(1.0 - ref_lengths / hyp_lengths)
    if any(n == 0 for n in num):
        return 0.0
    return bp * math.exp(sum(w * math.log(n / d) for w, n, d in zip(weights, num, den)))
