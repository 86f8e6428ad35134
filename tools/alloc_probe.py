"""How long does hipMalloc / hipFree of a lattice-arena-sized block take, alone and from two threads at once?
(the seconds-long stalls once blamed on overlapping cluster launches: profiles/EXPERIMENTS.md, round 4)

    python tools/alloc_probe.py [GB]
"""
import ctypes
import sys
import threading
import time

hip = ctypes.CDLL("libamdhip64.so")
GB = float(sys.argv[1]) if len(sys.argv) > 1 else 34.0
size = ctypes.c_size_t(int(GB * 1e9))


class NoLock:
    def __enter__(self): return self
    def __exit__(self, *a): return False


def once(out, i, lock=NoLock()):
    p = ctypes.c_void_p()
    hip.hipSetDevice(0)
    t0 = time.time()
    with lock:
        rc = hip.hipMalloc(ctypes.byref(p), size)
    t1 = time.time()
    hip.hipMemset(p, 0, ctypes.c_size_t(1 << 20))
    hip.hipDeviceSynchronize()
    t2 = time.time()
    with lock:
        rc2 = hip.hipFree(p)
    t3 = time.time()
    out[i] = "malloc %.3f s (rc %d), first touch %.3f s, free %.3f s (rc %d)" % (t1 - t0, rc, t2 - t1, t3 - t2, rc2)


hip.hipSetDevice(0)
for it in range(6):
    out = [None]
    once(out, 0)
    print("alone      ", it, out[0], flush=True)
for it in range(6):
    out = [None, None]
    ts = [threading.Thread(target=once, args=(out, i)) for i in range(2)]
    t0 = time.time()
    [t.start() for t in ts]
    [t.join() for t in ts]
    print("two threads", it, "%.2f s:" % (time.time() - t0), " | ".join(out), flush=True)
mutex = threading.Lock()
for it in range(8):
    out = [None, None]
    ts = [threading.Thread(target=once, args=(out, i, mutex)) for i in range(2)]
    t0 = time.time()
    [t.start() for t in ts]
    [t.join() for t in ts]
    print("two threads, one at a time in the allocator", it, "%.2f s:" % (time.time() - t0), " | ".join(out), flush=True)
