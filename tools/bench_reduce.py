"""Times lic_reduce_batch alone on job mixes shaped like the end-of-step batch of config 3 (tuning aid)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_image_compression_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")


def slab_job(splitk, taps, Cm, Cn):
    src = torch.randn(splitk * taps * Cm * Cn, device=dev)
    dst = torch.empty(taps * Cm * Cn, device=dev)
    j = L.ReduceJob()
    j.src, j.dst, j.kind, j.splitk, j.ntaps, j.Cm, j.Cn = src.data_ptr(), dst.data_ptr(), L.REDUCE_SLABS, splitk, taps, Cm, Cn
    j.sm, j.sn, j.stap, j.scale = Cn, 1, Cm * Cn, 1.0
    return j, (src, dst)


def col_job(rows, Cn):
    src = torch.randn(rows * Cn, device=dev)
    dst = torch.empty(Cn, device=dev)
    j = L.ReduceJob()
    j.src, j.dst, j.kind, j.splitk, j.Cn, j.scale = src.data_ptr(), dst.data_ptr(), L.REDUCE_COLUMNS, rows, Cn, 1.0
    return j, (src, dst)


def run(name, jobs):
    keep = [k for _, k in jobs]
    arr = (L.ReduceJob * len(jobs))(*[j for j, _ in jobs])
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    mb = sum(k[0].numel() * 4 for k in keep) / 1e6
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(6):
        flush.zero_()   # cold caches, as at the end of a step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.lic_reduce_batch(arr, len(jobs), st), "reduce")
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print(f"{name:42s} {mb:7.1f} MB  {t:7.1f} us  {mb / t * 1e-3 * 1e3:6.2f} TB/s" if t else name)


enc_slabs = lambda: [slab_job(16, 25, 128, 192), slab_job(30, 25, 128, 128), slab_job(30, 25, 128, 128), slab_job(512, 1, 80, 128)]
gam = lambda: [slab_job(256, 1, 128, 128) for _ in range(3)]
cols = lambda: [col_job(2048, 128), col_job(2048, 128), col_job(512, 128), col_job(512, 128), col_job(128, 128), col_job(128, 128),
                col_job(256, 192)]
run("encoder slabs (a4, a3, a2, stem)", enc_slabs())
run("GDN gamma slabs x3", gam())
run("column jobs (CS rows, bias sums)", cols())
run("stem slab job alone (512 x 80 x 128)", [slab_job(512, 1, 80, 128)])
run("one 2048-row column job", [col_job(2048, 128)])
run("everything", enc_slabs() + gam() + cols())
