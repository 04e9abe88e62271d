"""Synthetic batches of the benchmark shape: randn features, caption word counts
U{min_len..30} with caption 0 at the maximum (so the decoder runs all 31 steps), tokens U{3..V-1},
<EOS>=2 after the last word, <PAD>=0 after that; targets are time-major [31, B] int64 like the
reference's collate output (dataset/MSVD.py:53-74 after train.py:245)."""
import torch


def synthetic_targets(B, V, seed=1234, caption_max_len=30, full_length=True, min_len=4, lengths="uniform"):
    """lengths: "uniform" (the benchmark: U{min_len..30}, caption 0 at 30 words so T = 31) or "msvd" (3 + Poisson(5)
    words clipped to 30, the shape of MSVD captions — the batch then leaves the loop early, train.py:66)."""
    g = torch.Generator().manual_seed(seed)
    if lengths == "msvd":
        lens = (3 + torch.poisson(torch.full((B,), 5.0), generator=g)).clamp(max=caption_max_len).long()
        full_length = False
    else:
        lens = torch.randint(min_len, caption_max_len + 1, (B,), generator=g)
    if full_length:
        lens[0] = caption_max_len
    targets = torch.zeros(caption_max_len + 1, B, dtype=torch.long)
    for b in range(B):
        L = int(lens[b])
        targets[:L, b] = torch.randint(3, V, (L,), generator=g)
        targets[L, b] = 2
    return targets


def synthetic_features(B, F, D, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, F, D, generator=g)
