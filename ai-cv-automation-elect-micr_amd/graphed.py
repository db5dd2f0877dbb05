"""An engine's forward pass captured once per input shape into a hipGraph and replayed.

A forward pass of graph S is 45 launches of a few microseconds each; replaying them as one hipGraph removes the host's
launch work and most of the gap between kernels (0.91 -> 0.89 ms per 32 crops).  Measured on the big graphs (D 125 launches
in 28 ms, X, G) a replay changes nothing: their launch queue never drains, so bench.py keeps them eager.  The
engines are fixed launch sequences with no host read-back (batch statistics are folded on the device), so a capture is a
faithful recording: same kernels, same arguments, same bits.  Activations allocated during the capture live in the graph's
private pool and stay reserved for its lifetime -- one graph per (shape, device), evicted oldest-first beyond ``max_graphs``.
"""
from __future__ import annotations

from collections import OrderedDict


class GraphedForward:
    """``g = GraphedForward(engine); y = g(x)`` == ``engine.forward(x)``.  ``y`` is the graph's static output tensor: it is
    overwritten by the next call with the same shape (clone it to keep it)."""

    def __init__(self, engine, max_graphs: int = 4):
        self.engine = engine
        self.max_graphs = max_graphs
        self._graphs = OrderedDict()

    def _capture(self, x):
        import torch

        sx = x.clone()
        # one eager pass on a side stream first: lazy initialisation (library load, weight packing) must not be recorded
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            self.engine.forward(sx)
        torch.cuda.current_stream(x.device).wait_stream(side)
        torch.cuda.synchronize(x.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            sy = self.engine.forward(sx)
        return g, sx, sy

    def __call__(self, x):
        key = (tuple(x.shape), x.dtype, str(x.device))
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._capture(x)
            self._graphs[key] = entry
            while len(self._graphs) > self.max_graphs:
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        g, sx, sy = entry
        sx.copy_(x)
        g.replay()
        return sy
