"""Feature feed: what sits between the reference's DataLoader and the train step.

Host side, restating the batch semantics of the reference (dataset/MSVD.py:53-74, dataset/transform.py:9-63):
frame sampling to `encoder_output_len` frames, zero padding of short clips, collate to `[B,F,D]` float32 +
time-major `[L,B]` int64 targets, short final batches padded to B by REPEATING THE LAST SAMPLE (so B is always
exactly batch_size, SURVEY §8a trap 13).

Device side: `DeviceFeeder` double-buffers pinned host staging and issues the H2D copies on its own stream, so
the 17-206 MB of features per step move over PCIe while the previous step computes; it also derives, on the
host, the loop length T and the loss normalisers the step needs (api.decode_len / api.step_weights) — the train
loop never synchronises to find them (the reference syncs twice per decoder time step for this, train.py:54,66).
"""
import math

import numpy as np
import torch


# ----------------------------------------------------------------------------- frame samplers (transform.py:9-63)
def sample_frames(frames, n_sample, method="uniform", rng=None):
    """frames: [n, D] array.  Returns [n_sample, D] float32; clips shorter than n_sample are zero padded at the end
    (transform.py:ZeroPadIfLessThan)."""
    frames = np.asarray(frames)
    n = len(frames)
    if n >= n_sample:
        base = [int(i) for i in np.linspace(0, n - 1, n_sample)]
        if method == "uniform":
            idx = base
        elif method == "random":
            rng = rng or np.random
            idx = sorted(rng.choice(n, n_sample, replace=False))
        elif method == "uniform_jitter":
            rng = rng or np.random
            std = int(math.sqrt(n / n_sample / 2 / 2))
            idx = sorted(min(max(0, int(i + rng.normal(0, std))), n - 1) for i in base)
        else:
            raise NotImplementedError("Unknown frame sampling method: {}".format(method))
        frames = frames[idx]
    out = np.zeros((n_sample,) + frames.shape[1:], dtype=np.float32)
    out[:len(frames)] = frames
    return out


def pad_caption(tokens, max_sentence_len, eos=2, pad=0):
    """word indices -> [max_sentence_len + 1] int64: tokens, <EOS>, then <PAD> (transform.py PadLast + PadToLength)."""
    tokens = list(tokens)[:max_sentence_len]
    out = np.full(max_sentence_len + 1, pad, dtype=np.int64)
    out[:len(tokens)] = tokens
    out[len(tokens)] = eos
    return out


def collate_batch(videos, captions, batch_size):
    """dataset/MSVD.py:53-74.  videos: list of [F,D]; captions: list of [L] -> (enc [B,F,D] f32, targets [L,B] i64)."""
    videos, captions = list(videos), list(captions)
    if not videos:
        raise ValueError("empty batch")
    while len(videos) < batch_size:                     # repeat the last sample (MSVD.py:57-61)
        videos.append(videos[-1])
        captions.append(captions[-1])
    enc = np.stack([np.asarray(v, dtype=np.float32) for v in videos])
    tg = np.stack([np.asarray(c, dtype=np.int64) for c in captions]).T.copy()      # time-major (MSVD.py:72)
    return enc, tg


# ----------------------------------------------------------------------------- device feed
class DeviceFeeder:
    """Wraps an iterator of host batches (enc [B,F,D] float32 array/tensor, targets [L,B] integer array/tensor;
    L <= caption_max_len + 1) and yields (enc_dev, targets_dev, T, step_weight_dev) with batch i+1 already on its way
    while batch i trains.  targets are padded to caption_max_len + 1 rows.  `shard=(lo, hi)` keeps only the rank's
    captions on the device while T / step weights still come from the whole (global) batch.

    Three stages, each on its own resource, over a ring of `depth` slots (pinned staging + device buffers):
      worker thread : next(batches), host-side T / step weights, memcpy into the slot's pinned staging (numpy releases the
                      GIL; 17 MB take 0.34 ms of one core of the GPU box's host, 2 ms in the build container; `copy_threads`
                      > 1 splits the features into row blocks on a small pool — no gain on either host, default 1).
                      threaded=False stages on the calling thread instead (the bench's default: 1.70 against 1.81 ms per step)
      copy stream   : H2D of the staged slot, issued one batch ahead; waits (on the stream, not the host) for the
                      step that last read the slot's device buffers
      caller stream : waits for the slot's copy event only.
    The returned tensors are the slot's device buffers: they are valid until the call after next."""

    def __init__(self, batches, device, caption_max_len=30, shard=None, depth=3, threaded=True, ahead=1, copy_threads=1):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.copy_threads = max(1, int(copy_threads))
        self._pool = ThreadPoolExecutor(self.copy_threads) if self.copy_threads > 1 else None
        from .api import decode_len, step_weights
        self._decode_len, self._step_weights = decode_len, step_weights
        self.it = iter(batches)
        self.device = torch.device(device)
        self.Tm = caption_max_len + 1
        self.shard = shard
        self.ahead = max(1, int(ahead))                  # H2D copies kept in flight beyond the batch being handed out
        self.depth = max(self.ahead + 2, depth)
        self.stream = torch.cuda.Stream(device=self.device)
        self.free, self.staged = queue.Queue(), queue.Queue()
        self.inflight, self._last, self._done = [], None, False
        for _ in range(self.depth):
            self.free.put(None)                          # slots are allocated by the worker at first use (shapes)
        self._thread = threading.Thread(target=self._work, daemon=True) if threaded else None
        if threaded:
            self._thread.start()

    # ---- worker thread: host-only work
    def _new_slot(self, eshape, tshape):
        # ONE pinned staging blob and ONE device blob per slot (features | targets | step weights, each 256-byte aligned): a batch
        # is one H2D copy — three copies per batch cost the copy stream and the host three launches for 17 MB + 25 KB + 124 B
        import math
        ne, nt = int(np.prod(eshape)) * 4, int(np.prod(tshape)) * 8
        o_t = (ne + 255) // 256 * 256
        o_w = o_t + (nt + 255) // 256 * 256
        total = o_w + (self.Tm * 4 + 255) // 256 * 256
        blob_h = torch.empty(total, dtype=torch.uint8).pin_memory()
        blob_d = torch.empty(total, dtype=torch.uint8, device=self.device)

        def views(blob):
            return (blob[:ne].view(torch.float32).view(*eshape), blob[o_t:o_t + nt].view(torch.int64).view(*tshape),
                    blob[o_w:o_w + self.Tm * 4].view(torch.float32))
        eh, th, wh = views(blob_h)
        ed, td, wd = views(blob_d)
        wh.zero_()
        s = dict(blob_h=blob_h, blob_d=blob_d, enc_h=eh, tg_h=th, w_h=wh, enc_d=ed, tg_d=td, w_d=wd, ev=None, copied=None)
        s["enc_n"], s["tg_n"], s["w_n"] = s["enc_h"].numpy(), s["tg_h"].numpy(), s["w_h"].numpy()
        return s

    def _stage_one(self):
        """Host-only: pull one batch, derive T / step weights, memcpy into a free slot's pinned staging."""
        try:
            enc, tg = next(self.it)
        except StopIteration:
            return None
        enc = enc.numpy() if isinstance(enc, torch.Tensor) else np.asarray(enc)
        tg = tg.numpy() if isinstance(tg, torch.Tensor) else np.asarray(tg)
        full = np.zeros((self.Tm, tg.shape[1]), dtype=np.int64)
        full[:tg.shape[0]] = tg
        masks = full > 0
        T = self._decode_len(masks, self.Tm - 1)
        w = self._step_weights(masks, T)
        lo, hi = self.shard if self.shard else (0, enc.shape[0])
        enc, tgl = enc[lo:hi], full[:, lo:hi]
        s = self.free.get()
        if s is None or s["enc_n"].shape != enc.shape:
            s = self._new_slot(enc.shape, tgl.shape)
        if s["copied"] is not None and not s["copied"].query():
            # the staging's previous H2D has not drained: the copy stream holds it behind the step that last read the slot's
            # device buffers, i.e. the host has run ~6 steps ahead of the device.  This wait is the loop's back-pressure (round 4
            # measured 0.8 ms per batch here at the benchmark shape: the device, not the host, sets the pace)
            s["copied"].synchronize()
        n, k = enc.shape[0], self.copy_threads
        if self._pool is not None and enc.nbytes >= (4 << 20) and n >= k:
            futs = [self._pool.submit(np.copyto, s["enc_n"][i * n // k:(i + 1) * n // k], enc[i * n // k:(i + 1) * n // k], "same_kind")
                    for i in range(k)]
            for f in futs:
                f.result()
        else:
            np.copyto(s["enc_n"], enc, casting="same_kind")
        np.copyto(s["tg_n"], tgl); s["w_n"][:T] = w
        return s, T

    def _work(self):
        try:
            while True:
                item = self._stage_one()
                self.staged.put(item)
                if item is None:
                    return
        except BaseException as e:                       # surfaced by the consumer
            self.staged.put(e)

    # ---- consumer side: all GPU calls
    def _issue(self, block):
        if self._done:
            return False
        if self._thread is None:                         # inline staging (threaded=False)
            if self.free.empty():
                return False
            item = self._stage_one()
        else:
            try:
                item = self.staged.get(block=block)
            except Exception:
                return False
        if item is None or isinstance(item, BaseException):
            self._done = True
            if item is not None:
                raise item
            return False
        s, T = item
        with torch.cuda.stream(self.stream):
            if s["ev"] is not None:
                self.stream.wait_event(s["ev"])          # the step that read these device buffers has finished
            s["blob_d"].copy_(s["blob_h"], non_blocking=True)
            s["copied"] = torch.cuda.Event()
            s["copied"].record(self.stream)
        self.inflight.append((s, T))
        return True

    def __iter__(self):
        return self

    def __next__(self):
        # the batch handed out by the previous call has been consumed by work enqueued on the current stream since
        # then: fence its device buffers with an event and give the slot back to the worker
        if self._last is not None:
            self._last["ev"] = torch.cuda.Event()
            self._last["ev"].record(torch.cuda.current_stream())
            self.free.put(self._last)
            self._last = None
        if not self.inflight and not self._issue(block=True):
            raise StopIteration
        s, T = self.inflight.pop(0)
        while len(self.inflight) < self.ahead and self._issue(block=False):      # the next batches' H2D go out before this step is enqueued
            pass
        torch.cuda.current_stream().wait_event(s["copied"])
        self._last = s
        return s["enc_d"], s["tg_d"], T, s["w_d"][:T]
