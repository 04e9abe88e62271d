"""Where does the host spend its time in the fed train loop?  next(feeder) and the graph replay timed separately (host clocks),
with the device running the real step (bench.py --feed 1 shape)."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recnet_amd as R
from recnet_amd.feed import DeviceFeeder
from recnet_amd.synthetic import synthetic_features, synthetic_targets
from bench import build_models

B, F, D, V = 100, 28, 1536, 4188
C, dec, rec = build_models(R, dict(batch_size=B, use_recon=True, reconstructor_type="global", encoder_output_len=F, encoder_output_size=D,
                                   reconstructor_hidden_size=D, precision="bf16", device="cuda:0"), V)
tg = synthetic_targets(B, V, seed=1234)
step = R.DataParallelTrainStep(dec, rec, B, 0, 1, n_frames=F)
host = [(synthetic_features(B, F, D, seed=77 + i).numpy(), tg.numpy()) for i in range(3)]
for threaded, ahead in ((True, 2), (False, 2), (True, 1)):
    feeder = DeviceFeeder(itertools.cycle(host), "cuda:0", 30, threaded=threaded, depth=4, ahead=ahead)
    graphs = {}
    tn = tg_ = 0.0

    def run(timed):
        global tn, tg_
        t0 = time.perf_counter()
        e, t, T_, wd = next(feeder)
        t1 = time.perf_counter()
        g = graphs.get(e.data_ptr())
        if g is None:
            g = graphs[e.data_ptr()] = R.GraphedStep(step, e, t, T_, wd, warmup=0, defer_reconstructor_update="recurrent")
        g()
        t2 = time.perf_counter()
        if timed:
            tn += t1 - t0; tg_ += t2 - t1
    for _ in range(12):
        run(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        run(True)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("threaded=%s ahead=%d: %.3f ms per step (host loop %.3f): next(feeder) %.3f ms, replay %.3f ms" % (threaded, ahead, tt * 10, th * 10, tn * 10, tg_ * 10))
    list(graphs.values())[0].flush()
    torch.cuda.synchronize()
    del feeder
