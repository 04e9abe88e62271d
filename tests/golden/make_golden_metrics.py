#!/usr/bin/env python3
"""Golden scores for recnet_amd.metrics from the reference's own scorers (coco_caption/pycocoevalcap/{bleu,cider,rouge}).
Those files are Python 2; they are converted IN MEMORY with lib2to3 and executed here (build container only) — nothing of
them is written to the repo.  Output: tests/golden/metrics.json = the seeded corpus + the scores."""
import json
import os
import random
import sys
sys.dont_write_bytecode = True      # the reference tree under /root/reference stays untouched (no __pycache__ beside its modules)
import types
import warnings

warnings.simplefilter("ignore")
from lib2to3 import refactor  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/coco_caption/pycocoevalcap"
tool = refactor.RefactoringTool([f for f in refactor.get_fixers_from_package("lib2to3.fixes") if not f.endswith("fix_import")])


def load(modname, path, inject=None):
    src = open(path).read()
    if not src.endswith("\n"):
        src += "\n"
    src3 = str(tool.refactor_string(src, path))
    m = types.ModuleType(modname)
    m.__dict__.update(inject or {})
    sys.modules[modname] = m
    exec(compile(src3, path, "exec"), m.__dict__)
    return m


load("bleu_scorer", REF + "/bleu/bleu_scorer.py")
bleu = load("bleu", REF + "/bleu/bleu.py")
load("cider_scorer", REF + "/cider/cider_scorer.py")
cider = load("cider", REF + "/cider/cider.py")
rouge = load("rouge", REF + "/rouge/rouge.py")

rnd = random.Random(3)
vocab = ["a", "man", "woman", "dog", "is", "are", "playing", "guitar", "running", "the", "on", "field", "cooking",
         "with", "ball", "cat", "two", "people", "dancing", "in", "kitchen", "riding", "horse", "water", "person"]
gts, res = {}, {}
for i in range(40):
    base = [rnd.choice(vocab) for _ in range(rnd.randint(3, 10))]
    refs = []
    for _ in range(rnd.randint(1, 5)):
        r = [w if rnd.random() > 0.3 else rnd.choice(vocab) for w in base]
        if rnd.random() < 0.4:
            r = r[:max(1, len(r) - rnd.randint(1, 3))]
        if rnd.random() < 0.3:
            r = r + [rnd.choice(vocab) for _ in range(rnd.randint(1, 3))]
        refs.append(" ".join(r))
    hyp = [w if rnd.random() > 0.35 else rnd.choice(vocab) for w in base][:rnd.randint(1, 12)]
    gts["v%d" % i], res["v%d" % i] = refs, [" ".join(hyp)]
gts["exact"], res["exact"] = ["a man is playing guitar", "a man plays the guitar"], ["a man is playing guitar"]
gts["none"], res["none"] = ["two people dancing"], ["cat"]

_stdout = sys.stdout
sys.stdout = open(os.devnull, "w")          # the BLEU scorer prints its totals
b, b_per = bleu.Bleu(4).compute_score(gts, res)
sys.stdout = _stdout
c, c_per = cider.Cider().compute_score(gts, res)
r, r_per = rouge.Rouge().compute_score(gts, res)
out = {"gts": gts, "res": res, "ids": list(gts.keys()), "bleu": [float(x) for x in b],
       "bleu_per_id": [[float(x) for x in row] for row in b_per], "cider": float(c), "cider_per_id": [float(x) for x in c_per],
       "rouge": float(r), "rouge_per_id": [float(x) for x in r_per]}
json.dump(out, open(os.path.join(HERE, "metrics.json"), "w"), indent=0)
print("bleu", out["bleu"], "cider", out["cider"], "rouge", out["rouge"])
