"""Where are the ~35-55 us between two replayed train steps?  The step's first kernel (advance_step_kernel) and last kernel
(export_scalars_kernel) keep rings of their own wall-clock stamps (Engine.step_ring): after a burst of back-to-back replays
   end_i - start_i          = the step's span on the device
   start_{i+1} - end_i      = idle time between the last kernel of one replay and the first kernel of the next
are read directly.  Variants: one step per graph (the bench), two and four steps per graph (is it the replay boundary?)."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import recnet_amd as R
from recnet_amd import _lib
from recnet_amd.synthetic import synthetic_features, synthetic_targets
kind = sys.argv[1] if len(sys.argv) > 1 else "global"
B, F, D, V = 100, 28, 1536, 4188
C = R.make_config(batch_size=B, encoder_output_len=F, encoder_output_size=D, use_recon=True, reconstructor_type=kind, reconstructor_hidden_size=D, precision="bf16")
torch.manual_seed(0)
dec = R.build_decoder(V, C); rec = R.build_reconstructor(C)
step = R.DataParallelTrainStep(dec, rec, B, 0, 1, n_frames=F)
enc = synthetic_features(B, F, D, seed=1234).cuda(); tg = synthetic_targets(B, V, seed=1234)
T, w = step.prepare(tg.numpy()); tg = tg.cuda()
run = R.GraphedStep(step, enc, tg, T, w, warmup=3, defer_reconstructor_update="recurrent")
eng = step.step_impl.engine
def burst(fn, n):
    for _ in range(12): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
ms = burst(run, 28)
ring = eng.step_ring()
spans = [b - a for a, b in ring]; gaps = [ring[i + 1][0] - ring[i][1] for i in range(len(ring) - 1)]
print("one step per graph : %.4f ms per step | spans %s | between steps %s" % (ms, " ".join("%.1f" % x for x in spans), " ".join("%.1f" % x for x in gaps)))
# the host keeps at most `depth` replays queued behind the running one (an event per replay, waited for `depth` replays later)
for depth in (1, 2, 4):
    evs = [torch.cuda.Event() for _ in range(64)]
    def paced(n):
        for i in range(n):
            if i >= depth: evs[(i - depth) % 64].synchronize()
            run(); evs[i % 64].record()
    paced(12); torch.cuda.synchronize()
    t0 = time.perf_counter(); paced(28); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 28 * 1e3
    ring = eng.step_ring()
    spans = [b - a for a, b in ring]; gaps = [ring[i + 1][0] - ring[i][1] for i in range(len(ring) - 1)]
    print("at most %d queued     : %.4f ms per step | spans %s | between steps %s" % (depth, ms, " ".join("%.1f" % x for x in spans), " ".join("%.1f" % x for x in gaps)))
# k steps per graph — only on request (`... global multi`): capturing several steps into ONE graph records the handle's fork / join events
# more than once inside a capture, and about one run in eight of this section then aborts inside the HIP runtime's heap (free():
# invalid pointer in hipGraphLaunch or at exit; 2 of 16 runs with and without the vendor-library products) — the pattern round 4 found
# with the profile brackets.  The product captures ONE step per graph.
for k in ((2, 4) if len(sys.argv) > 2 and sys.argv[2] == "multi" else ()):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(k):
            eng.train_step_dev(enc, tg, T, w, run.seed_base, run.flags)
    def rep():
        g.replay(); eng.mark_pending()
    ms = burst(rep, 28 // k) / k
    ring = eng.step_ring()
    spans = [b - a for a, b in ring]; gaps = [ring[i + 1][0] - ring[i][1] for i in range(len(ring) - 1)]
    print("%d steps per graph: %.4f ms per step | spans %s | between steps %s" % (k, ms, " ".join("%.1f" % x for x in spans), " ".join("%.1f" % x for x in gaps)))
run.flush(); torch.cuda.synchronize()
