"""Recorded step: host side of csrc/replay.hip (include/focr.h focr_replay_*).

`record(fn)` runs `fn()` once under HIP stream capture (torch owns the capture: its caching allocator serves the step's
temporaries from a private pool, so every address the recorded launches carry stays valid and untouched between replays)
and turns the captured graph into a launch list inside the library; `Recorded.launch()` re-issues the whole step with ONE
library call.  hipGraphLaunch is never used (on this runtime it costs as much host time as the
eager step).

What a recorded step may contain: kernel launches (this library's and ATen's), memsets, linear memcpys between stable
addresses, on any number of streams joined by events.  What it may not: host synchronisation, `.item()`, pageable
host-to-device copies, anything whose ARGUMENTS change from step to step -- per-step scalars (dropout epoch, Adam step
count) therefore live in device memory (`focr_step_state_*`), inputs and labels in static buffers the caller fills
before each launch.
"""
import ctypes

import torch

from . import _lib

N_LANES = 6
_RAW_STREAM = torch._C._cuda_getCurrentRawStream


class Recorded:
    def __init__(self, graph, handle, lanes, info):
        self.graph, self.handle, self.lanes, self.info = graph, handle, lanes, info
        self._launch = _lib.load().focr_replay_launch
        self._h = ctypes.c_void_p(handle)

    def launch(self, stream=None):
        """re-issue the recorded step, ordered on `stream` (raw pointer; default: torch's current stream) -- one library call"""
        if self._launch(self._h, stream if stream is not None else _RAW_STREAM(torch.cuda.current_device())) != 0:
            raise RuntimeError("focr_replay_launch failed: " + _lib.load().focr_last_error().decode())

    def node_names(self):
        buf = ctypes.create_string_buffer(512)
        out = []
        for i in range(self.info["nodes"]):
            _lib.call("focr_replay_node_name", self._h, i, buf, 512)
            out.append(buf.value.decode())
        return out

    def node_lanes(self):
        arr = (ctypes.c_int * self.info["nodes"])()
        _lib.call("focr_replay_lanes", self._h, arr, self.info["nodes"])
        return list(arr)

    def probe(self, pattern, depth=1):
        """time every kernel node whose (mangled) name contains `pattern` in all later launches, keeping the last `depth`
        launches -> number of nodes"""
        n = _lib.load().focr_replay_probe(self._h, pattern.encode(), int(depth))
        if n < 0:
            raise RuntimeError("focr_replay_probe failed: " + _lib.load().focr_last_error().decode())
        return n

    def probe_read(self):
        """[(node index, mean ms over the launches held, launches held)] (synchronises)"""
        torch.cuda.synchronize()
        lib = _lib.load()
        n = lib.focr_replay_probe_read(self._h, None, None, None, 0)
        if n <= 0:
            return []
        ms, node, cnt = (ctypes.c_float * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
        if lib.focr_replay_probe_read(self._h, ms, node, cnt, n) < 0:
            raise RuntimeError("focr_replay_probe_read failed: " + lib.focr_last_error().decode())
        return [(int(node[i]), float(ms[i]), int(cnt[i])) for i in range(n)]

    def close(self):
        if self.handle:
            _lib.call("focr_replay_destroy", self._h)
            self.handle = None
            self.graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def record(fn, pool=None, lanes=None):
    """Capture one call of `fn` on the current stream and build its launch list.  Returns (Recorded, fn's return value).
    The caller has run `fn` eagerly a few times before (lazy one-time work -- weight tables, workspaces, autotuning -- must
    not end up inside the recording) and keeps every tensor `fn` reads at a fixed address."""
    lib = _lib.load()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    torch.cuda.synchronize()
    kw = {"pool": pool} if pool is not None else {}
    # (torch captures on a side stream of its own; which stream a launch is ordered on is chosen per launch)
    with torch.cuda.graph(g, **kw):
        out = fn()
    raw = g.raw_cuda_graph()
    # lanes: streams for the chains of the recording, longest chain first.  Streams share a small number of hardware
    # queues (two streams on one queue run strictly one after the other), so the caller passes streams it KNOWS to run
    # beside its main stream -- the engine's weight-gradient side stream --; further lanes are fresh streams.
    lanes = list(lanes or [torch.cuda.current_stream()])
    lanes = lanes[:N_LANES] + [torch.cuda.Stream() for _ in range(N_LANES - len(lanes))]
    arr = (ctypes.c_void_p * N_LANES)(*[s.cuda_stream for s in lanes])
    handle = ctypes.c_void_p()
    rc = lib.focr_replay_build(ctypes.c_void_p(raw), arr, N_LANES, ctypes.byref(handle))
    if rc != 0:
        raise RuntimeError("focr_replay_build failed (%d): %s" % (rc, lib.focr_last_error().decode()))
    cnt = (ctypes.c_int * 8)()
    _lib.call("focr_replay_info", handle, cnt)
    keys = ("nodes", "kernels", "memsets", "memcpys", "empty", "lanes", "waits", "events")
    return Recorded(g, handle.value, lanes, dict(zip(keys, cnt))), out
