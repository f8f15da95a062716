"""Would the two weight-gradient GEMMs of a step (dW1: needs nothing of layer 0's backward; dW0: the step's last GEMM) gain from running
in ONE launch?  Upper bound without writing that kernel: the two launches on two streams against back to back on one.
usage: python profiles/debug/tn_pair_overlap.py"""
import sys
sys.path.insert(0, ".")
import importlib, torch
ops = importlib.import_module("gnn-tableextraction_amd.ops")
dev = torch.device("cuda:0")
N = 24317


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for f0, hid in ((831, 256), (13, 218), (363, 149), (831, 1000)):
    img = lambda c: ops.p3_from_f32(torch.randn(N, c, device=dev))
    dz1, q1, y0, dz0, x, ahn = img(hid), img(hid), img(hid), img(hid), img(f0), img(f0)
    o1, o0 = torch.empty(hid, 2 * hid, device=dev), torch.empty(hid, 2 * f0, device=dev)
    dw1 = lambda: ops.gemm_p3_tn(dz1, y0, a2=q1, two_segments=True, out=o1)
    dw0 = lambda: ops.gemm_p3_tn(dz0, x, b2=ahn, two_segments=True, out=o0)

    def both_seq():
        dw1(); dw0()

    def both_par():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        s1.wait_event(ev); s2.wait_event(ev)
        with torch.cuda.stream(s1):
            dw1()
        with torch.cuda.stream(s2):
            dw0()
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(s1); e2.record(s2)
        cur.wait_event(e1); cur.wait_event(e2)
    both_par()
    t1, t0, ts, tp = timeit(dw1), timeit(dw0), timeit(both_seq), timeit(both_par)
    print(f"(F0 {f0}, H {hid}): dW1 {t1:6.1f} us  dW0 {t0:6.1f} us  back to back {ts:6.1f} us  two streams {tp:6.1f} us  (each with its fold launch)", flush=True)
