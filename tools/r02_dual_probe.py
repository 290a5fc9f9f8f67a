"""GPU box: do two queue pipelines, each on its own stream with half of the slots and half of the resident trace blocks, overlap?
Two DeviceScenes render the two interleaved-tile shards of one frame from two host threads (ctypes releases the GIL); compared with
one DeviceScene rendering the whole frame.  usage: python tools/r02_dual_probe.py <workload> [blocks_per_cu_each]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "p1000000"
per_cu = sys.argv[2] if len(sys.argv) > 2 else "3"
text, label, _ = bench.workload(wl)
os.environ["PBRT_HIP_PIPELINE"] = "1"

def make(rank, world, slots, blocks):
    os.environ["PBRT_HIP_PIPE_SLOTS"] = str(slots)
    if blocks: os.environ["PBRT_HIP_TRACE_BLOCKS_PER_CU"] = blocks
    else: os.environ.pop("PBRT_HIP_TRACE_BLOCKS_PER_CU", None)
    ps = pkg.ParsedScene(text=text); ps.set_shard(rank, world, 48)
    ds = pkg.DeviceScene(ps, device=0)
    film = torch.zeros((5, ps.height, ps.width), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    ds.set_stream(st.cuda_stream); ds.bind_film(film.data_ptr())
    return ds, film, st

def timed(scenes, reps=3):
    def run(ds):
        ds.render(sync=False)
    best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.time()
        th = [threading.Thread(target=run, args=(s[0],)) for s in scenes]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); best = min(best, time.time() - t0)
    return best * 1e3

# PBRT_HIP_PIPE_SLOTS / TRACE_BLOCKS_PER_CU are read at scene creation (blocks) and at render time (slots): keep one setting per phase
one = [make(0, 1, 1 << 23, None)]
print(wl, "one pipeline, 8 M slots: %.2f ms" % timed(one), flush=True)
del one; torch.cuda.empty_cache()
two = [make(0, 2, 1 << 22, per_cu), make(1, 2, 1 << 22, per_cu)]
os.environ["PBRT_HIP_PIPE_SLOTS"] = str(1 << 22)
print(wl, "two pipelines on two streams, 4 M slots and %s trace blocks per CU each: %.2f ms" % (per_cu, timed(two)), flush=True)
one_half = two[:1]
print(wl, "  (one of the two halves alone: %.2f ms)" % timed(one_half), flush=True)
