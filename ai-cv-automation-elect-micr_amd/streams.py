"""Two halves of a batch down two HIP streams.

A chain of separable convs at 1/16 resolution alternates a depthwise kernel (HBM-bound, no LDS) with a pointwise GEMM
(matrix-core bound, one 144 KB-LDS workgroup per CU): run back to back, each kind leaves the other's unit idle.  The images
of a batch are independent in graphs D and G (folded batch norms) and every kernel treats them so -- same tiles, same order,
tests/test_d_gpu.py compares batch sizes bit for bit -- so the two halves can go down two streams, block by block, and one
half's depthwise kernels share the chip with the other half's GEMMs: same bits, graph D 28.0 -> 27.5 ms per 32 images.
(Measured alternatives: 4 quarter batches 28.2 ms, alternating layer by layer 27.7 ms; residual 1x1 convs forked beside
their block's separable convs: no change.)  Not for graph X: its norms use the statistics of the whole batch.
"""
from __future__ import annotations


class TwoHalves:
    def __init__(self, device):
        self.device = device
        self._side = None

    def run(self, x, out, make_chain):
        """make_chain(x_part, out_part) -> generator that issues the launches of one part and yields after every block.
        x, out: ops.Act over the same batch (out allocated by the caller on the current stream)."""
        import torch

        half = x.B // 2
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        chains = []
        for h, st in enumerate(self._side):
            st.wait_stream(main)          # the inputs are ready; they stay referenced by the caller until the join below
            chains.append(make_chain(x.images(h * half, (h + 1) * half), out.images(h * half, (h + 1) * half)))
        live = [True, True]
        while any(live):
            for h, st in enumerate(self._side):
                if live[h]:
                    with torch.cuda.stream(st):   # temporaries are allocated and freed on the part's own stream
                        live[h] = next(chains[h], "done") != "done"
        for st in self._side:
            main.wait_stream(st)
        return out
