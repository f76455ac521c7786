"""Partition of one frame over ranks and its recombination with torch.distributed.

Whitted / binary: 8-row bands dealt round-robin (rt_params.band_*), disjoint outputs -> gather.
Path trace: strata ranges (rt_params.sample_begin/sample_end), partial radiance sums -> reduce(SUM)
BEFORE the non-linear stages (division by pixelSamples, tone map, exposure, gamma, clamp:
/root/reference/src/RayTracer.cpp:301-315).  The same functions run over RCCL on GPUs (bench.py)
and over gloo on CPU tensors (tests/test_multiproc_gloo.py).
"""
import torch
import torch.distributed as dist


def band_rows_of(rank, world, height, band_rows=8):
    """Global row indices (bottom row = 0) a rank renders, in the order of its compact buffer."""
    n_bands = (height + band_rows - 1) // band_rows
    return [j for b in range(rank, n_bands, world) for j in range(b * band_rows, min(height, (b + 1) * band_rows))]


def max_band_rows(world, height, band_rows=8):
    n_bands = (height + band_rows - 1) // band_rows
    return ((n_bands + world - 1) // world) * band_rows


def strata_range(rank, world, n_strata):
    """[begin, end) of the strata a rank accumulates; an empty share is (n, n) because
    sample_end <= 0 means "all" in rt_params."""
    lo, hi = (n_strata * rank) // world, (n_strata * (rank + 1)) // world
    return (lo, hi) if hi > lo else (n_strata, n_strata)


class BandGatherer:
    """Gathers every rank's compact band buffer on rank 0 and scatters the rows into the frame."""

    def __init__(self, rank, world, height, width, device, band_rows=8, dtype=torch.float64, channels=3):
        self.rank, self.world, self.H, self.W, self.C = rank, world, height, width, channels
        self.slot_rows = max_band_rows(world, height, band_rows)
        self.mine = torch.zeros(self.slot_rows * width * channels, dtype=dtype, device=device)
        self.frame = torch.empty(height, width, channels, dtype=dtype, device=device) if rank == 0 else None
        self.slots = [torch.empty_like(self.mine) for _ in range(world)] if (rank == 0 and world > 1) else None
        self.index = None
        if rank == 0:
            self.index = [torch.tensor(band_rows_of(q, world, height, band_rows), dtype=torch.long, device=device)
                          for q in range(world)]

    def gather(self):
        """self.mine must hold this rank's rows.  Returns the frame on rank 0, None elsewhere."""
        if self.world == 1:
            n = self.index[0].numel()
            self.frame.index_copy_(0, self.index[0], self.mine[: n * self.W * self.C].view(n, self.W, self.C))
            return self.frame
        dist.gather(self.mine, self.slots, dst=0)
        if self.rank != 0:
            return None
        for q in range(self.world):
            n = self.index[q].numel()
            self.frame.index_copy_(0, self.index[q], self.slots[q][: n * self.W * self.C].view(n, self.W, self.C))
        return self.frame


class FramePipeline:
    """Whitted frames of a multi-GPU job, pipelined: every rank renders its bands of frame k into one of
    `depth` device buffers, converts them to PPM bytes (rt_quantize_device: an eighth of the fp64 data), and
    the gather of those bytes to rank 0 runs asynchronously (RCCL's own stream, xGMI) while frame k+1 is
    being rendered.  complete(k) makes the current stream wait for frame k's gather and scatters its rows
    into `frame` (rank 0: uint8 [H, W, 3], row 0 = bottom row)."""

    def __init__(self, rank, world, height, width, device, band_rows=8, depth=2):
        self.rank, self.world, self.H, self.W, self.depth = rank, world, height, width, depth
        self.slot_rows = max_band_rows(world, height, band_rows)
        n = self.slot_rows * width * 3
        self.rgb = [torch.zeros(n, dtype=torch.float64, device=device) for _ in range(depth)]
        self.u8 = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.pending = [None] * depth
        self.frame = None
        self.slots = None
        self.index = None
        if rank == 0 and world > 1:
            # one receive buffer per frame in flight, [world, slot_rows, W*3]; row H of `_frame` is a dump for
            # the padding rows of ranks whose share is short, so that ONE index_copy_ assembles a frame
            self._frame = torch.empty(height + 1, width * 3, dtype=torch.uint8, device=device)
            self.frame = self._frame[:height].view(height, width, 3)
            self._recv = [torch.empty(world, self.slot_rows, width * 3, dtype=torch.uint8, device=device) for _ in range(depth)]
            self.slots = [[self._recv[i][q].view(-1) for q in range(world)] for i in range(depth)]
            idx = torch.full((world, self.slot_rows), height, dtype=torch.long)
            for q in range(world):
                rows = band_rows_of(q, world, height, band_rows)
                idx[q, : len(rows)] = torch.tensor(rows, dtype=torch.long)
            self.index = idx.view(-1).to(device)

    def target(self, k):
        """(fp64 render target, uint8 quantization target) of frame k."""
        return self.rgb[k % self.depth], self.u8[k % self.depth]

    def submit(self, k):
        """Call after frame k's render and quantization were enqueued on the current stream."""
        i = k % self.depth
        if self.world > 1:
            self.pending[i] = dist.gather(self.u8[i], self.slots[i] if self.rank == 0 else None, dst=0, async_op=True)

    def complete(self, k):
        """Frame k is assembled on rank 0 once the current stream reaches this point."""
        i = k % self.depth
        if self.world == 1:  # one rank's compact buffer IS the frame (bands in order)
            self.frame = self.u8[i][: self.H * self.W * 3].view(self.H, self.W, 3)
            return self.frame
        work, self.pending[i] = self.pending[i], None
        if work is not None:
            work.wait()
        if self.rank != 0:
            return None
        self._frame.index_copy_(0, self.index, self._recv[i].view(self.world * self.slot_rows, self.W * 3))
        return self.frame


class AccumulatorPipeline:
    """Path-trace passes of a multi-GPU job, pipelined the same way: pass k's partial radiance sums are
    reduced (SUM) to rank 0 asynchronously while pass k+1 runs."""

    def __init__(self, world, n_values, device, depth=2):
        self.world, self.depth = world, depth
        self.accum = [torch.empty(n_values, dtype=torch.float64, device=device) for _ in range(depth)]
        self.pending = [None] * depth

    def target(self, k):
        return self.accum[k % self.depth]

    def submit(self, k):
        i = k % self.depth
        if self.world > 1:
            self.pending[i] = dist.reduce(self.accum[i], dst=0, op=dist.ReduceOp.SUM, async_op=True)

    def complete(self, k):
        i = k % self.depth
        work, self.pending[i] = self.pending[i], None
        if work is not None:
            work.wait()
        return self.accum[i]


def reduce_accumulators(accum, world):
    """Sum of the per-rank radiance accumulators on rank 0 (in place)."""
    if world > 1:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    return accum
