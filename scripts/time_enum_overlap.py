"""Does the breadth-first phase of one part of the rank space hide under the leaf phase of another?  C(32,16) cut
into `world` cost-balanced shards that run AT THE SAME TIME on one GPU (one host thread, context and problem
replica per shard; the exchange goes through host memory): wall time per full solve against the single pass."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplexmethod_amd import capi
m, n = 16, 32
A, b, c, _ = capi.gen_lp(0, m, n)
STAGGERS = [float(v) for v in os.environ.get("LP_STAGGER_MS", "0").split(",")]
for world in (1, 2, 3, 4):
  for stagger in STAGGERS:
    if world == 1 and stagger: continue
    comms = capi.Comm.local(world) if world > 1 else [None]
    ctxs = [capi.Context(0) for _ in range(world)]
    probs = [cx.enum_problem(A, b, c, True) for cx in ctxs]
    res = [None] * world
    def run(r):
        if stagger and r:
            t_end = time.perf_counter() + stagger * r * 1e-3
            while time.perf_counter() < t_end: pass
        res[r] = probs[r].solve_sharded(comms[r], n - m, want_vertex=False)
    best = 1e9
    for rep in range(5):
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        if rep: best = min(best, dt)
    print("world %d on one GPU, shard r starts %.1f ms x r late: %.3f ms per full C(32,16) solve (rank %d, counts %s)" % (world, stagger, best * 1e3, res[0]["rank"], res[0]["counts"]), flush=True)
    for p in probs: p.free()
    for cm in comms:
        if cm is not None: cm.destroy()
    for cx in ctxs: cx.close()
