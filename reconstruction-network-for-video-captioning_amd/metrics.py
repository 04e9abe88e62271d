"""Caption metrics in Python 3: BLEU-1..4, ROUGE-L and CIDEr with the definitions of the
coco-caption scorers the reference vendors under coco_caption/pycocoevalcap (bleu/bleu_scorer.py, rouge/rouge.py,
cider/cider_scorer.py; called from eval.py:158-166 through COCOEvalCap).  METEOR and the PTB tokenizer are Java programs
and are not available here: captions are expected already tokenised (lower-case words separated by single spaces),
which is what `evaluate` below produces from vocabulary indices.

Inputs follow the scorers' convention: gts = {id: [reference strings]}, res = {id: [one candidate string]}.
Pinned against the reference's own scorer code on a seeded corpus: tests/golden/make_golden_metrics.py.
"""
import math
from collections import Counter

import numpy as np


def _ngrams(words, n_max=4):
    c = Counter()
    for n in range(1, n_max + 1):
        for i in range(len(words) - n + 1):
            c[tuple(words[i:i + n])] += 1
    return c


def _check(gts, res):
    if set(gts.keys()) != set(res.keys()):
        raise ValueError("gts and res must have the same ids")
    for k in res:
        if not isinstance(res[k], list) or len(res[k]) != 1 or not isinstance(gts[k], list) or not gts[k]:
            raise ValueError("res[id] must be [candidate], gts[id] a non-empty list of references")
    return list(gts.keys())


# ----------------------------------------------------------------------------- BLEU
def bleu(gts, res, n_max=4):
    """Corpus BLEU-1..n with the 'closest' effective reference length (bleu.py:44).  Returns ([B1..Bn], per-id lists)."""
    ids = _check(gts, res)
    tiny, small = 1e-15, 1e-9
    tot_guess, tot_correct = [0] * n_max, [0] * n_max
    tot_test = tot_ref = 0
    per_id = [[] for _ in range(n_max)]
    for k in ids:
        hyp = res[k][0].split()
        refs = [r.split() for r in gts[k]]
        cap = Counter()
        for r in refs:                                     # clip by the max count over the references
            for g, c in _ngrams(r, n_max).items():
                if c > cap[g]:
                    cap[g] = c
        tl = len(hyp)
        rl = min((abs(len(r) - tl), len(r)) for r in refs)[1]     # closest, ties -> shorter
        guess = [max(0, tl - n + 1) for n in range(1, n_max + 1)]
        correct = [0] * n_max
        for g, c in _ngrams(hyp, n_max).items():
            correct[len(g) - 1] += min(cap.get(g, 0), c)
        tot_test += tl
        tot_ref += rl
        ratio = (tl + tiny) / (rl + small)
        bp = math.exp(1 - 1 / ratio) if ratio < 1 else 1.0
        acc = 1.0
        for n in range(n_max):
            tot_guess[n] += guess[n]
            tot_correct[n] += correct[n]
            acc *= (correct[n] + tiny) / (guess[n] + small)
            per_id[n].append(acc ** (1.0 / (n + 1)) * bp)
    ratio = (tot_test + tiny) / (tot_ref + small)
    bp = math.exp(1 - 1 / ratio) if ratio < 1 else 1.0
    out, acc = [], 1.0
    for n in range(n_max):
        acc *= (tot_correct[n] + tiny) / (tot_guess[n] + small)
        out.append(acc ** (1.0 / (n + 1)) * bp)
    return out, per_id


# ----------------------------------------------------------------------------- ROUGE-L
def _lcs(a, b):
    if len(a) < len(b):
        a, b = b, a
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b, 1):
            cur.append(prev[j - 1] + 1 if x == y else max(prev[j], cur[j - 1]))
        prev = cur
    return prev[-1]


def rouge_l(gts, res, beta=1.2):
    """Mean over ids of the F_beta of (max precision, max recall) of the LCS over the references (rouge.py:43-71)."""
    ids = _check(gts, res)
    scores = []
    for k in ids:
        c = res[k][0].split(" ")
        p_max = r_max = 0.0
        for ref in gts[k]:
            r = ref.split(" ")
            l = _lcs(r, c)
            p_max, r_max = max(p_max, l / float(len(c))), max(r_max, l / float(len(r)))
        scores.append((1 + beta ** 2) * p_max * r_max / (r_max + beta ** 2 * p_max) if p_max and r_max else 0.0)
    return float(np.mean(scores)), np.array(scores)


# ----------------------------------------------------------------------------- CIDEr
def cider(gts, res, n_max=4, sigma=6.0):
    """CIDEr (cider_scorer.py:95-197): tf-idf n-gram vectors (idf over the reference sets of the corpus), clipped cosine
    similarity, Gaussian length penalty (lengths counted in bigram occurrences, as the scorer does), mean over n, mean
    over the references, times 10."""
    ids = _check(gts, res)
    hyps = [_ngrams(res[k][0].split(), n_max) for k in ids]
    refs = [[_ngrams(r.split(), n_max) for r in gts[k]] for k in ids]
    df = Counter()
    for rs in refs:
        for g in set(g for r in rs for g in r):
            df[g] += 1
    log_n = math.log(float(len(ids)))

    def vec(cnt):
        v = [dict() for _ in range(n_max)]
        norm = [0.0] * n_max
        length = 0
        for g, tf in cnt.items():
            n = len(g) - 1
            w = float(tf) * (log_n - math.log(max(1.0, df.get(g, 0.0))))
            v[n][g] = w
            norm[n] += w * w
            if n == 1:
                length += tf
        return v, [math.sqrt(x) for x in norm], length

    scores = []
    for h, rs in zip(hyps, refs):
        hv, hn, hl = vec(h)
        acc = np.zeros(n_max)
        for r in rs:
            rv, rn, rl = vec(r)
            pen = math.e ** (-(float(hl - rl) ** 2) / (2 * sigma ** 2))
            for n in range(n_max):
                s = sum(min(w, rv[n].get(g, 0.0)) * rv[n].get(g, 0.0) for g, w in hv[n].items())
                if hn[n] != 0 and rn[n] != 0:
                    s /= hn[n] * rn[n]
                acc[n] += s * pen
        scores.append(float(np.mean(acc)) / len(rs) * 10.0)
    return float(np.mean(scores)), np.array(scores)


def score_all(gts, res):
    """The score dict eval.py:158-166 builds (without METEOR)."""
    b, _ = bleu(gts, res)
    return {"Bleu_1": b[0], "Bleu_2": b[1], "Bleu_3": b[2], "Bleu_4": b[3], "CIDEr": cider(gts, res)[0],
            "ROUGE_L": rouge_l(gts, res)[0]}


def indices_to_sentence(idxs, idx2word, eos=2):
    """Words up to (excluding) the first <EOS> (train.py: convert_idxs_to_sentences / eval.py:143-150)."""
    words = []
    for i in idxs:
        i = int(i)
        if i == eos:
            break
        words.append(idx2word[i])
    return " ".join(words)
