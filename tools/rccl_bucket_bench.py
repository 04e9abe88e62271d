#!/usr/bin/env python3
"""RCCL cost of the train step's gradient buckets, measured with however many ranks the launcher provides (this pool: ONE —
so what is measured is the fixed cost of the collective path per bucket: RCCL kernel launch + its device-side copy, and
for the direct transport the staging / accumulation kernels around it; no byte crosses xGMI).  Buckets are the ones
api.GraphedStep reduces (dp.DataParallelTrainStep.early_buffers / late_buffers) for the named configuration.

    python tools/rccl_bucket_bench.py [--rec global|local] [--feat 1536]      (or under torch.distributed.run)

Prints one JSON object: per bucket and transport the mean time of one start()+finish() pair in microseconds, stream
time by hipEvents, over 50 calls after 10 warm-up calls."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recnet_amd.dp import GradTransport  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rec", default="global")
ap.add_argument("--feat", type=int, default=1536)
args = ap.parse_args()
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dev = torch.device("cuda", torch.cuda.current_device())
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
V, E, H, A, RA, D = 4188, 468, 512, 128, 128, args.feat
R = D
p_dec = A + V * E + A * H + A * D + A + 4 * H * (E + D) + 4 * H * H + 8 * H + V * H + V
p_out = V * H + V
p_rec = (RA + RA * R + RA * H + RA + 4 * R * H + 4 * R * R + 8 * R + R * R + R) if args.rec == "local" else \
        (4 * R * 2 * H + 4 * R * R + 8 * R + R * R + R)
buckets = {"early: reconstructor": p_rec, "early: decoder out.*": p_out, "late: rest of the decoder": p_dec - p_out}
res = {"world_size": world, "rec": args.rec, "feat": D, "buckets_mb_fp32": {k: round(v * 4 / 1e6, 1) for k, v in buckets.items()},
       "us_per_call": {}}
for name, n in buckets.items():
    buf = torch.randn(n, device=dev)
    for dtype, algo in (("f32", "ring"), ("f32", "direct"), ("bf16", "direct")):
        tr = GradTransport(dtype, None, algo)
        for _ in range(10):
            tr.finish(tr.start(buf))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            tr.finish(tr.start(buf))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        gb = n * 4 / 1e9
        res["us_per_call"]["%s | %s %s" % (name, dtype, algo)] = {"us": round(us, 1), "fp32_GB_per_s": round(gb / (us * 1e-6), 1)}
        buf.normal_()
if rank == 0:
    print(json.dumps(res, indent=1))
dist.barrier()
dist.destroy_process_group()
