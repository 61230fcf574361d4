"""Host side of the input pipeline (replaces the per-item `Image.open(...).convert("RGB")` + host transform of
data/dataset.py:56-89): image files -> decoded uint8 RGB pixels packed into ONE staging buffer, in parallel.

Two back ends with the same result (`Packed`):
  * threads  -- Pillow's decoders release the GIL; each thread opens, decodes and copies its image straight into the staging
                buffer (a bump allocator under a lock hands out the slots), so nothing of the chunk is serial but the slot grant;
  * processes -- N worker processes (plain `python decode.py <shm>` children speaking JSON lines over pipes: no fork of a process
                that holds a HIP context, no re-import of the parent's __main__) decode into disjoint regions of a shared-memory
                segment; the parent copies the packed bytes into its page-locked staging buffer.
This module imports neither torch nor the native library (the workers stay light)."""
import json
import os
import subprocess
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ALIGN = 256


def _round(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


def decode_file(path):
    """uint8 [H, W, 3] array of an image file, as `Image.open(path).convert("RGB")` gives it (an image that already is RGB is
    not copied a second time)."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode != "RGB":
            im = im.convert("RGB")
        return np.asarray(im, dtype=np.uint8)


def usable_cpus():
    """CPUs this process can actually use: the affinity mask, capped by the cgroup CPU quota (the MI355X box shows 256 logical CPUs
    under a 16-CPU quota: more decode processes than the quota only add throttling)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def default_processes():
    """Decode processes for GRIP_DECODE_PROCS=auto: 1.5 per usable CPU, at most 48; 0 (= the thread back end) below 4 CPUs.
    More workers than CPUs because a worker also waits (file reads, the job / reply pipes, the parent's copy of its region):
    measured on the MI355X box's 16-CPU quota, files -> embeddings with the encode running: 12 / 16 / 20 / 24 processes =
    9.5k / 10.3k / 11.8k / 13.3k images/s (tools/files_bench.py).  Under a multi-rank launcher ($LOCAL_WORLD_SIZE) the CPUs are divided among the ranks."""
    n = usable_cpus()
    # every rank of a node sees the same affinity mask / cgroup quota: share it (an 8-rank launch on a 16-CPU quota would start 192 workers)
    local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    n = max(1, n // local)
    return min(48, (3 * n + 1) // 2) if n >= 4 else 0


class Packed:
    """Decoded images of one chunk inside a staging buffer: `offsets[i]` (bytes), `shapes[i]` = (H, W); image i occupies
    H*W*3 bytes at its offset.  `used` = high-water mark of the buffer.  `regions` (process back end): the spans
    (first byte, bytes written, first image, one past the last image) each worker filled inside its own part of the segment --
    the gaps between them hold nothing, `compact_into` copies only the spans."""
    __slots__ = ("offsets", "shapes", "used", "regions")

    def __init__(self, offsets, shapes, used, regions=None):
        self.offsets, self.shapes, self.used, self.regions = offsets, shapes, used, regions

    def compact_into(self, src, dst, pool=None):
        """Copy the decoded bytes from the shared segment `src` into the staging buffer `dst` WITHOUT the gaps between the
        workers' regions (the segment is sized 1.5x and split evenly: copying / uploading it whole moves ~1.5x the bytes) and
        return the Packed that describes `dst`.  Region copies run on `pool` (a ThreadPoolExecutor) when given."""
        if not self.regions:
            dst[:self.used] = src[:self.used]
            return self
        offsets = self.offsets.copy()
        jobs, at = [], 0
        for base, nbytes, lo, hi in self.regions:
            if nbytes:
                jobs.append((base, nbytes, at))
                offsets[lo:hi] += at - base
                at += nbytes

        def copy(job):
            base, nbytes, to = job
            dst[to:to + nbytes] = src[base:base + nbytes]
        if pool is not None and len(jobs) > 1:
            list(pool.map(copy, jobs))
        else:
            for j in jobs:
                copy(j)
        return Packed(offsets, self.shapes, at)


def decode_threads(paths, buf, pool):
    """Decode `paths` into the uint8 numpy buffer `buf` on the ThreadPoolExecutor `pool`.  Returns (Packed, overflow) where
    overflow = {index: array} holds the images that did not fit (the caller uploads them on their own and sizes the next buffer
    from the bytes per image it has seen)."""
    n = len(paths)
    offsets = np.zeros(n, dtype=np.int64)
    shapes = np.zeros((n, 2), dtype=np.int32)
    lock = threading.Lock()
    state = {"cursor": 0}
    overflow = {}
    cap = buf.shape[0]

    def one(i):
        arr = decode_file(paths[i])
        size = arr.shape[0] * arr.shape[1] * 3
        with lock:
            off = state["cursor"]
            fits = off + size <= cap
            if fits:
                state["cursor"] = off + _round(size)
        shapes[i] = arr.shape[:2]
        if fits:
            offsets[i] = off
            buf[off:off + size] = arr.reshape(-1)
        else:
            overflow[i] = arr
    if pool is None or n < 2:
        for i in range(n):
            one(i)
    else:
        list(pool.map(one, range(n)))
    return Packed(offsets, shapes, state["cursor"]), overflow


class ProcessDecoder:
    """`n_proc` decode workers around `slots` shared-memory staging segments (one per chunk in flight, so chunk i+1 can be decoded
    while chunk i is still being uploaded; each segment grows on its own while it is free).  Worker w of a job owns the w-th
    sub-region of the slot's segment; what does not fit comes back as overflow (decoded again in the parent -- rare: the caller
    sizes the segments from the bytes per image it has seen)."""

    def __init__(self, n_proc, slot_bytes, slots=2):
        self.n_proc, self.slots = n_proc, slots
        self.segs, self.bufs = [None] * slots, [None] * slots
        self._dropped = []
        for k in range(slots):
            self.ensure(k, slot_bytes)
        self.workers = [subprocess.Popen([sys.executable, os.path.abspath(__file__)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                         env=dict(os.environ), text=True, bufsize=1) for _ in range(n_proc)]
        self._lock = threading.Lock()
        import atexit
        import weakref
        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close())     # no worker or segment outlives the interpreter

    def ensure(self, slot, nbytes):
        """Make slot `slot`'s segment at least `nbytes` large.  Only while nothing reads or writes the slot.  Returns True when the
        segment was (re)created (the caller re-registers it for DMA)."""
        from multiprocessing import shared_memory
        if self.segs[slot] is not None and self.segs[slot].size >= nbytes:
            return False
        self._release(slot)
        self.segs[slot] = shared_memory.SharedMemory(create=True, size=_round(int(nbytes)))
        self.bufs[slot] = np.ndarray((self.segs[slot].size,), dtype=np.uint8, buffer=self.segs[slot].buf)
        return True

    def _release(self, slot):
        seg = self.segs[slot]
        if seg is None:
            return
        self._dropped.append(seg.name)
        self.bufs[slot] = None
        try:
            seg.close()
        except BufferError:          # a view is still alive somewhere: the mapping goes with it
            pass
        seg.unlink()
        self.segs[slot] = None

    def slot_view(self, slot):
        return self.bufs[slot]

    def decode(self, paths, slot):
        """Decode `paths` into slot `slot`'s segment.  Returns (Packed, overflow dict)."""
        n = len(paths)
        offsets = np.zeros(n, dtype=np.int64)
        shapes = np.zeros((n, 2), dtype=np.int32)
        overflow = {}
        if n == 0:
            return Packed(offsets, shapes, 0), overflow
        nw = min(self.n_proc, n)
        region = (self.segs[slot].size // nw) // ALIGN * ALIGN
        bounds = [n * w // nw for w in range(nw + 1)]
        used = 0
        with self._lock:
            dropped, self._dropped = self._dropped, []
            for w in range(self.n_proc):
                if w >= nw and not dropped:
                    continue
                job = {"shm": self.segs[slot].name, "paths": list(paths[bounds[w]:bounds[w + 1]]) if w < nw else [], "base": w * region, "cap": region,
                       "drop": dropped}
                self.workers[w].stdin.write(json.dumps(job) + "\n")
                self.workers[w].stdin.flush()
            replies = []
            for w in range(self.n_proc):        # every reply is consumed before an error is raised: the pipes stay in step
                if w >= nw and not dropped:
                    continue
                line = self.workers[w].stdout.readline()
                replies.append(json.loads(line) if line else {"error": f"worker died (exit code {self.workers[w].poll()})"})
            for w, rep in enumerate(replies):
                if "error" in rep:
                    raise RuntimeError(f"decode worker {w}: {rep['error']}")
            regions = []
            for w, rep in enumerate(replies[:nw]):
                end = w * region
                for j, (off, h, wd) in enumerate(rep["items"]):
                    i = bounds[w] + j
                    shapes[i] = (h, wd)
                    if off < 0:
                        overflow[i] = None
                    else:
                        offsets[i] = off
                        end = max(end, off + _round(h * wd * 3))
                regions.append((w * region, end - w * region, bounds[w], bounds[w + 1]))
                used = max(used, end)
        for i in overflow:
            overflow[i] = decode_file(paths[i])
        return Packed(offsets, shapes, used, regions), overflow

    def close(self):
        for p in self.workers:
            try:
                p.stdin.close()
                p.wait(timeout=5)
            except Exception:
                p.kill()
        self.workers = []
        for k in range(self.slots):
            self._release(k)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_thread_pool(workers):
    return ThreadPoolExecutor(max_workers=workers) if workers > 1 else None


def _worker_main():
    from multiprocessing import resource_tracker, shared_memory
    attached = {}           # segment name -> (SharedMemory, uint8 view)
    for line in sys.stdin:
        try:
            job = json.loads(line)
            for name in job.get("drop", []):
                ent = attached.pop(name, None)
                if ent is not None:
                    seg, view = ent
                    del view, ent
                    seg.close()
            if job["shm"] not in attached:
                seg = shared_memory.SharedMemory(name=job["shm"])
                try:
                    resource_tracker.unregister(seg._name, "shared_memory")     # the parent owns the segment
                except Exception:
                    pass
                attached[job["shm"]] = (seg, np.ndarray((seg.size,), dtype=np.uint8, buffer=seg.buf))
            buf = attached[job["shm"]][1]
            cur, end, items = job["base"], job["base"] + job["cap"], []
            for p in job["paths"]:
                arr = decode_file(p)
                size = arr.shape[0] * arr.shape[1] * 3
                if cur + size <= end:
                    buf[cur:cur + size] = arr.reshape(-1)
                    items.append((cur, int(arr.shape[0]), int(arr.shape[1])))
                    cur += _round(size)
                else:
                    items.append((-1, int(arr.shape[0]), int(arr.shape[1])))
            del buf
            sys.stdout.write(json.dumps({"items": items}) + "\n")
        except Exception as e:      # reported to the parent, which raises
            sys.stdout.write(json.dumps({"error": f"{type(e).__name__}: {e}"}) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    _worker_main()
