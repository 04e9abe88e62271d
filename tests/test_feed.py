"""Host-side feed semantics (CPU): frame samplers, caption padding and the collate rule of the reference
(dataset/MSVD.py:53-74, dataset/transform.py)."""
import numpy as np
import pytest

from recnet_amd import feed


def test_uniform_sampling_indices_and_zero_padding():
    fr = np.arange(100, dtype=np.float32)[:, None] * np.ones((1, 3), dtype=np.float32)
    out = feed.sample_frames(fr, 28, "uniform")
    assert out.shape == (28, 3)
    assert np.array_equal(out[:, 0], np.array([int(i) for i in np.linspace(0, 99, 28)], dtype=np.float32))
    short = feed.sample_frames(fr[:5], 28, "uniform")
    assert np.array_equal(short[:5], fr[:5]) and not short[5:].any()        # ZeroPadIfLessThan


def test_random_and_jitter_sampling_are_sorted_and_in_range():
    rng = np.random.RandomState(0)
    fr = np.arange(300, dtype=np.float32)[:, None]
    for m in ("random", "uniform_jitter"):
        out = feed.sample_frames(fr, 28, m, rng)[:, 0]
        assert len(out) == 28 and np.all(np.diff(out) >= 0) and out.min() >= 0 and out.max() <= 299
    assert len(set(feed.sample_frames(fr, 28, "random", rng)[:, 0])) == 28   # without replacement
    with pytest.raises(NotImplementedError):
        feed.sample_frames(fr, 28, "nearest")


def test_caption_padding():
    c = feed.pad_caption([5, 6, 7], 30)
    assert c.shape == (31,) and list(c[:5]) == [5, 6, 7, 2, 0] and not c[4:].any()
    c = feed.pad_caption(list(range(3, 40)), 30)                              # truncated to 30 words + <EOS>
    assert c[29] == 32 and c[30] == 2


def test_collate_repeats_last_sample_and_is_time_major():
    vids = [np.full((28, 4), i, dtype=np.float32) for i in range(3)]
    caps = [feed.pad_caption([3 + i, 9], 30) for i in range(3)]
    enc, tg = feed.collate_batch(vids, caps, 5)
    assert enc.shape == (5, 28, 4) and tg.shape == (31, 5) and tg.dtype == np.int64
    assert np.array_equal(enc[3], vids[2]) and np.array_equal(enc[4], vids[2])   # MSVD.py:57-61
    assert list(tg[0]) == [3, 4, 5, 5, 5] and list(tg[2]) == [2] * 5


# ----------------------------------------------------------------------------- against the reference's own code
import os
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feed.npz"))


@pytest.mark.parametrize("ci", range(7))
@pytest.mark.parametrize("mi,method", [(0, "uniform"), (1, "random"), (2, "uniform_jitter")])
def test_samplers_match_reference(ci, mi, method):
    """Same np.random seed, same draws -> the exact frames dataset/transform.py picks."""
    np.random.seed(100 + 10 * ci + mi)
    got = feed.sample_frames(G["frames_%d" % ci], int(G["nsample_%d" % ci]), method, np.random)
    assert np.array_equal(got, G["sampled_%d_%d" % (ci, mi)])


@pytest.mark.parametrize("i", range(4))
def test_caption_padding_matches_reference(i):
    assert np.array_equal(feed.pad_caption(G["cap_in_%d" % i], 30), G["cap_out_%d" % i])


def test_collate_matches_reference():
    enc, tg = feed.collate_batch([G["sampled_%d_0" % i] for i in range(3)], [G["cap_out_%d" % i] for i in range(3)], 5)
    assert np.array_equal(enc, G["collate_videos"]) and np.array_equal(tg, G["collate_captions"].astype(np.int64))
