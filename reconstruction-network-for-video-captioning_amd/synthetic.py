"""Synthetic batches of the benchmark shape: randn features, caption word counts
U{min_len..30} with caption 0 at the maximum (so the decoder runs all 31 steps), tokens U{3..V-1},
<EOS>=2 after the last word, <PAD>=0 after that; targets are time-major [31, B] int64 like the
reference's collate output (dataset/MSVD.py:53-74 after train.py:245)."""
import torch


def synthetic_targets(B, V, seed=1234, caption_max_len=30, full_length=True, min_len=4):
    g = torch.Generator().manual_seed(seed)
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
