"""The shell around the train step: what train.py:227-420 and eval.py:123-169 do around the hot
path — iterate batches, log the three losses, run the validation pass, score captions with a search method, write
checkpoints — with the device work routed through this package:

  train step        feed.DeviceFeeder (H2D one batch ahead) -> hipGraph replay of the fused step, one graph per T
  validation        forward_decoder (free running, on-device arg-max feedback) + reconstructor forward, eval mode
  scoring           search.greedy_search / beam_search on the device, metrics.score_all (BLEU / CIDEr / ROUGE-L)
  checkpoints       checkpoint.save_checkpoint (the reference's dict layout)

Losses are read back only when they are logged (`log_every`), not three `.item()` syncs per iteration (train.py:275-277).
Out of scope here as in the tier: tensorboard, the MSVD files, vocabulary building.
"""
import os

import torch

from . import metrics
from .api import (GraphedStep, build_decoder, build_reconstructor, forward_decoder, forward_global_reconstructor,
                  forward_local_reconstructor)
from .checkpoint import load_checkpoint, save_checkpoint
from .dp import DataParallelTrainStep
from .feed import DeviceFeeder
from .search import beam_search, greedy_search


class Trainer:
    """decoder / reconstructor dicts as train.py:222-225 builds them, plus the step machinery."""

    def __init__(self, C, n_vocabs, rank=0, world_size=1, group=None, use_graphs=True, eager_steps=2,
                 defer_reconstructor_update=False):
        self.C, self.rank, self.world = C, rank, world_size
        self.decoder = build_decoder(n_vocabs, C)
        self.reconstructor = build_reconstructor(C) if C.use_recon else None
        self.global_batch = C.batch_size
        self.dp = DataParallelTrainStep(self.decoder, self.reconstructor, self.global_batch, rank, world_size, group=group)
        self.use_graphs, self.eager_steps = use_graphs, eager_steps
        # opt-in (measured slower at the benchmark shape): graph steps on one rank leave the
        # reconstructor's update pending for the next step to run under its decoder forward chain (api.GraphedStep);
        # flush() completes it wherever the parameters are read (validation, checkpoints, the end of fit)
        # (True: the whole update, global reconstructor; "recurrent": the split update of round 4 — d W_hh and its Adam step only)
        self.defer = defer_reconstructor_update if (defer_reconstructor_update and world_size == 1 and use_graphs and self.reconstructor is not None) else False
        self._graphs, self._static = {}, None
        self.iteration = 0
        self._eager_until = eager_steps       # optimiser step count up to which steps are launched eagerly

    def flush(self):
        """Completes a pending deferred reconstructor update (stream-ordered; a no-op when there is none)."""
        self.dp.step_impl.engine.flush()

    def resume(self, path):
        """Continue a run from a checkpoint written by `fit` (or by the reference's train.py:398-420): parameters, Adam
        state, step count and iteration are restored, the packed operand images of every bound engine are refreshed
        (checkpoint.load_checkpoint), captured graphs are dropped and the next `eager_steps` steps run eagerly again
        (module loading, RCCL set-up) before a graph is captured."""
        self.flush()
        ckpt = load_checkpoint(path, self.decoder, self.reconstructor)
        self.iteration = int(ckpt.get("iteration", self.decoder["_state"].step))
        self._graphs.clear()
        self._eager_until = self.decoder["_state"].step + self.eager_steps
        return ckpt

    def check_health(self, scalars=None):
        """Raises if a persistent chain kernel gave up a bounded wait (the step's gradients were garbage; the optimiser
        kernels skipped the update) or the loss is not finite.  Before raising, the handle is switched to the per-step
        kernels and the captured graphs are dropped, so a caller that catches the error can keep training.

        Data parallel: a collective — every rank calls it at the same iterations (fit does) and the status words are
        MAX-reduced first, so all ranks raise (and fall back) together instead of the healthy ones blocking in the next
        all-reduce.  The update itself was already skipped on EVERY rank: the affected rank marks the last gradient
        bucket with a NaN before the all-reduce (csrc/kernels_util.hpp: poison_mark_kernel / poison_collect_kernel)."""
        eng = self.dp.step_impl.engine
        st = eng.chain_status()
        bad_loss = scalars is not None and not bool(torch.isfinite(scalars[6]))
        where = ""
        if self.world > 1:
            import torch.distributed as dist
            # gloo has no bitwise OR: one MAX per status bit (9 bits) + the loss flag
            word = torch.tensor([float((st >> k) & 1) for k in range(9)] + [float(bad_loss)], device=scalars.device if scalars is not None
                                else next(self.decoder["model"].parameters()).device)
            mine = (st, bad_loss)
            dist.all_reduce(word, op=dist.ReduceOp.MAX, group=self.dp.group)
            w = word.cpu().tolist()
            st = sum(1 << k for k in range(9) if w[k] > 0)
            bad_loss = w[9] > 0
            if (st & 0xFF) and not (mine[0] & 0xFF):
                where = " (raised by another rank)"
        if st or bad_loss:
            eng.chain_reset(disable_persistent=True)
            self._graphs.clear()
            raise RuntimeError("train step %d: %s%s; the optimiser updates of the affected steps were skipped on every rank, the "
                               "engine now uses the per-step kernels" % (self.iteration, "persistent chain kernel gave up waiting "
                               "(status 0x%x: another kernel held the CUs it needs)" % st if st else "non-finite loss", where))

    def global_scalars(self, acc):
        """acc: device scalars summed over steps on THIS rank.  Under data parallelism the CE / MSE parts ([0], [3]) are
        per-rank partial sums (dp.py): all-reduce them (a collective: every rank must call this) and rebuild the derived
        values.  Returns a host list in _lib.SCALAR_NAMES order."""
        a = acc.clone()
        if self.world > 1:
            import torch.distributed as dist
            part = torch.stack([a[0], a[3]])
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.dp.group)
            a[0], a[3] = part[0], part[1]
        v = a.cpu().tolist()
        hy = self.decoder["_hyper"]
        v[2] = v[0] + hy["decoder_lambda_reg"] * v[1]
        v[5] = v[3] + hy["reconstructor_lambda_reg"] * v[4] if self.reconstructor else 0.0
        v[6] = v[2] + (hy["lambda_recon"] * v[5] if self.reconstructor else 0.0)
        return v

    # ------------------------------------------------------------------ one optimiser step
    def step(self, enc, targets, T, w):
        """enc [B_local,F,D], targets [31,B_local] on the device; T / w from the GLOBAL batch (feed.DeviceFeeder)."""
        self.iteration += 1
        # (decoder_teacher_forcing_ratio < 1: the draw of train.py:38 is made on the host in every iteration — eager steps only)
        if not self.use_graphs or self.dp.step_impl.teacher_forcing_ratio < 1.0 or self.decoder["_state"].step < self._eager_until:
            return self.dp(enc, targets, T, w)         # first steps eagerly: module loading, RCCL set-up
        if self._static is None:
            self._static = (torch.empty_like(enc), torch.empty_like(targets))
        senc, stg = self._static
        senc.copy_(enc); stg.copy_(targets)
        g = self._graphs.get(T)
        if g is None:                                  # one graph per loop length T (<= 31 of them)
            g = self._graphs[T] = (GraphedStep(self.dp, senc, stg, T, torch.empty_like(w), warmup=0,
                                               defer_reconstructor_update=self.defer))
        g.w.copy_(w)
        return g()

    # ------------------------------------------------------------------ validation pass, train.py:310-372
    @torch.no_grad()
    def validate(self, batches, idx2word=None):
        """batches: host (enc, targets) pairs of the GLOBAL batch size; single-rank evaluation of full batches."""
        C, dec, rec = self.C, self.decoder, self.reconstructor
        self.flush()
        dec["model"].eval()
        if rec:
            rec["model"].eval()
        tot = {"loss": 0.0, "dec": 0.0, "rec": 0.0}
        n, gt, pd = 0, [], []
        dev = next(dec["model"].parameters()).device
        for enc, targets in batches:
            enc = torch.as_tensor(enc, dtype=torch.float32).to(dev)
            targets = torch.as_tensor(targets).long().to(dev)
            dl, hid, idx = forward_decoder(dec, enc, targets, targets > 0)            # default ratio 0: free running
            rl = None
            if rec:
                fwd = forward_global_reconstructor if C.reconstructor_type == "global" else forward_local_reconstructor
                rl = fwd(hid, enc, rec)
            loss = dl + C.lambda_recon * rl if rec else dl
            tot["loss"] += float(loss) * C.batch_size
            tot["dec"] += float(dl) * C.batch_size
            tot["rec"] += (float(rl) * C.batch_size) if rec else 0.0
            n += C.batch_size
            if idx2word is not None:
                gt += [metrics.indices_to_sentence(c, idx2word) for c in targets.t().cpu().tolist()]
                pd += [metrics.indices_to_sentence(c, idx2word) for c in idx.t().cpu().tolist()]
        dec["model"].train()
        if rec:
            rec["model"].train()
        out = {k: v / max(n, 1) for k, v in tot.items()}
        out["captions"] = list(zip(gt, pd))
        return out

    # ------------------------------------------------------------------ the loop, train.py:227-420
    def fit(self, batches, n_iterations, log_every=None, val_batches=None, validate_every=None, save_every=None,
            save_dpath=None, idx2word=None, log=print, health_every=200):
        C = self.C
        lo, hi = self.dp.lo, self.dp.hi
        dev = next(self.decoder["model"].parameters()).device
        feeder = DeviceFeeder(batches, dev, C.caption_max_len, shard=(lo, hi))
        hist = []
        acc = torch.zeros(8, device=dev)
        n_acc = 0
        for enc, targets, T, w in feeder:
            sc = self.step(enc, targets, T, w)
            acc += sc                                   # stays on the device; read at log time only
            n_acc += 1
            it = self.iteration
            checked = False
            if log_every and it % log_every == 0:
                self.check_health(sc); checked = True
                # the reference divides its running sums by log_every * batch_size (train.py:282-286)
                a = [x / (n_acc * C.batch_size) for x in self.global_scalars(acc)]
                rec = {"iteration": it, "loss": a[6], "dec": a[2], "rec": a[5]}
                hist.append(rec)
                msg = "Iter {} / {} ({:.1f}%): loss {:.5f}".format(it, n_iterations, it / n_iterations * 100, a[6])
                if C.use_recon:
                    msg += " (dec {:.5f} + rec {:.5f})".format(a[2], a[5])
                log(msg)
                acc.zero_()
                n_acc = 0
            if validate_every and val_batches is not None and it % validate_every == 0 and self.rank == 0:
                v = self.validate(val_batches() if callable(val_batches) else val_batches, idx2word)
                hist.append({"iteration": it, "val_loss": v["loss"], "val_dec": v["dec"], "val_rec": v["rec"]})
                log("[Validation] Iter {} / {}: loss {:.5f} (dec {:.5f} + rec {:.5f})".format(it, n_iterations, v["loss"],
                                                                                            v["dec"], v["rec"]))
            if save_every and save_dpath and it % save_every == 0:
                if not checked:
                    self.check_health(sc); checked = True   # never write weights of a step that went wrong
                gl = self.global_scalars(sc)           # collective under data parallelism: every rank
                self.flush()
                if self.rank == 0:
                    os.makedirs(save_dpath, exist_ok=True)
                    save_checkpoint(os.path.join(save_dpath, "{}_checkpoint.tar".format(it)), it, self.decoder,
                                    self.reconstructor, loss=torch.tensor(gl[6]), config=C)
            # health is checked even when nothing is logged or saved (one stream sync + a 40-byte read every health_every steps)
            if not checked and health_every and (it % health_every == 0 or it >= n_iterations):
                self.check_health(sc)
            if it >= n_iterations:
                break
        self.flush()
        return hist


# ---------------------------------------------------------------------- scoring, eval.py:123-169
@torch.no_grad()
def evaluate(config, score_batches, decoder, search_method, idx2word, references):
    """score_batches: iterable of (vids, enc [B,F,D]) with B == config.batch_size (the score loader repeats its last
    sample to fill the batch, dataset/MSVD.py:76-93; "PAD" ids are dropped like eval.py:145).  search_method: "greedy" or
    ("beam", width).  references: {vid: [caption strings]}.  Returns the score dict of metrics.score_all."""
    decoder.eval()
    dev = next(decoder.parameters()).device
    B, H = config.batch_size, decoder.hidden_size
    res = {}
    for vids, enc in score_batches:
        enc = torch.as_tensor(enc, dtype=torch.float32).to(dev)
        inp = torch.full((1, B), 1, dtype=torch.long, device=dev)                      # eval.py:131
        hid = torch.zeros(1, B, H, device=dev)
        if config.decoder_model == "LSTM":
            hid = (hid, torch.zeros(1, B, H, device=dev))
        if isinstance(search_method, str):
            if search_method != "greedy":
                raise NotImplementedError("Unknown search method: {}".format(search_method))
            steps = greedy_search(config, decoder, inp, hid, enc)                     # [n_steps][B]
            caps = list(map(list, zip(*steps)))
        else:
            method, width = search_method
            if method != "beam":
                raise NotImplementedError("Unknown search method: {}".format(method))
            caps = beam_search(config, width, None, decoder, inp, hid, enc)
        for vid, c in zip(vids, caps):
            if vid != "PAD" and vid not in res:
                res[vid] = [metrics.indices_to_sentence(c, idx2word)]
    gts = {v: references[v] for v in res}
    return metrics.score_all(gts, res)
