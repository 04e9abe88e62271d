"""Can two RCCL ranks share ONE GPU on this pool ("a 2-process-on-one-GPU RCCL test that actually moves bytes
between ranks if the box allows it, else say it does not")?  Two processes, both on cuda:0, world size 2, backend "nccl":
an all-reduce of a small tensor.  Prints what happened; exit code 0 either way (the answer is the output)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp


def worker(rank, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda:0"))
        x = torch.full((1 << 20,), float(rank + 1), device="cuda:0")
        dist.all_reduce(x)
        torch.cuda.synchronize()
        q.put((rank, "ok", float(x[0].item())))
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        q.put((rank, "error", repr(e)[:400]))


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = []
    for p in ps:
        p.join(120)
    while not q.empty():
        out.append(q.get())
    for p in ps:
        if p.is_alive():
            p.terminate(); out.append(("?", "timeout", "a rank did not return within 120 s"))
    for o in sorted(out, key=str):
        print("rank %s: %s %s" % o)
    ok = len(out) == 2 and all(o[1] == "ok" and abs(o[2] - 3.0) < 1e-6 for o in out)
    print("two RCCL ranks on one GPU: %s" % ("an all-reduce moved bytes between the ranks (sum 3.0 on both)" if ok else "NOT available on this box"))
    sys.exit(0)
