"""Partition of one frame over ranks and its recombination with torch.distributed.

Whitted / binary: 8-row bands dealt round-robin (rt_params.band_*), disjoint outputs -> gather.
Path trace: strata ranges (rt_params.sample_begin/sample_end), partial radiance sums -> reduce-scatter(SUM)
BEFORE the non-linear stages (division by pixelSamples, tone map, exposure, gamma, clamp:
/root/reference/src/RayTracer.cpp:301-315), which every rank applies to its slice -> gather.  The same functions run over RCCL on GPUs (bench.py)
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


_concurrent = {}  # device -> streams already found to run side by side


def concurrent_streams(n, device, candidates=32, spin_cycles=2_000_000):
    """n torch streams on `device` whose kernels really run side by side.

    HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and two streams
    that share a queue serialize; which streams share one is not visible through the API and is not
    simply creation order.  So this measures it: a candidate joins the set if spin kernels on the set
    plus the candidate take the time of one spin, not two.  Raises if fewer than n can be found."""
    if n <= 1:
        return [torch.cuda.current_stream(device)] if n == 1 else []
    key = torch.device(device).index
    if len(_concurrent.get(key, [])) >= n:
        return _concurrent[key][:n]
    pool = [torch.cuda.Stream(device=device) for _ in range(candidates)]  # torch hands out its pool of 32 round-robin

    def spin_ms(streams):
        torch.cuda.synchronize(device)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        first = torch.cuda.current_stream(device)
        for s in streams:
            s.wait_stream(first)
            with torch.cuda.stream(s):
                torch.cuda._sleep(spin_cycles)
        for s in streams:
            first.wait_stream(s)
        b.record()
        torch.cuda.synchronize(device)
        return a.elapsed_time(b)

    spin_ms(pool[:1])  # warm-up
    one = min(spin_ms(pool[:1]) for _ in range(2))
    chosen = [pool[0]]
    for c in pool[1:]:
        if len(chosen) == n:
            break
        if min(spin_ms(chosen + [c]) for _ in range(2)) < 1.5 * one:
            chosen.append(c)
    if len(chosen) > len(_concurrent.get(key, [])):
        _concurrent[key] = chosen
    if len(chosen) < n:
        raise RuntimeError(f"only {len(chosen)} of {n} streams run concurrently on {device} (GPU_MAX_HW_QUEUES?)")
    return chosen


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
    `depth` slots (device buffers), converts them to PPM bytes (rt_quantize_device: an eighth of the fp64
    data), and the gather of those bytes to rank 0 runs asynchronously (RCCL's own stream, xGMI) while later
    frames are being rendered.  complete(k) makes the slot's stream wait for frame k's gather and scatters
    its rows into that slot's frame (rank 0: uint8 [H, W, 3], row 0 = bottom row).

    `streams` (optional, one per slot): frames in flight on different streams -- with one render context
    per slot (rt_set_workgroups) the frames share the chip instead of queueing; stream(k) is the one to
    make current (torch.cuda.stream) around frame k's enqueue / submit / complete.  Without it every
    slot uses the current stream."""

    def __init__(self, rank, world, height, width, device, band_rows=8, depth=2, streams=None):
        self.rank, self.world, self.H, self.W, self.depth = rank, world, height, width, depth
        assert streams is None or len(streams) == depth
        self.streams = streams
        self.slot_rows = max_band_rows(world, height, band_rows)
        n = self.slot_rows * width * 3
        self.rgb = [torch.zeros(n, dtype=torch.float64, device=device) for _ in range(depth)]
        self.u8 = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.pending = [None] * depth
        self.frame = None
        self.slots = None
        self.index = None
        if rank == 0 and world > 1:
            # per slot: one receive buffer [world, slot_rows, W*3] and one assembled frame; row H of a frame
            # is a dump for the padding rows of ranks whose share is short, so that ONE index_copy_ assembles it
            self._frame = [torch.empty(height + 1, width * 3, dtype=torch.uint8, device=device) for _ in range(depth)]
            self._recv = [torch.empty(world, self.slot_rows, width * 3, dtype=torch.uint8, device=device) for _ in range(depth)]
            self.slots = [[self._recv[i][q].view(-1) for q in range(world)] for i in range(depth)]
            idx = torch.full((world, self.slot_rows), height, dtype=torch.long)
            for q in range(world):
                rows = band_rows_of(q, world, height, band_rows)
                idx[q, : len(rows)] = torch.tensor(rows, dtype=torch.long)
            self.index = idx.view(-1).to(device)

    def stream(self, k):
        """The stream frame k's work belongs on (None: the current one)."""
        return self.streams[k % self.depth] if self.streams else None

    def target(self, k):
        """(fp64 render target, uint8 quantization target) of frame k."""
        return self.rgb[k % self.depth], self.u8[k % self.depth]

    def submit(self, k):
        """Call after frame k's render and quantization were enqueued on the current stream."""
        i = k % self.depth
        if self.world > 1:
            self.pending[i] = dist.gather(self.u8[i], self.slots[i] if self.rank == 0 else None, dst=0, async_op=True)

    def complete(self, k):
        """Frame k is assembled on rank 0 once the current stream reaches this point.  Must be called
        before the slot is used again (frame k + depth)."""
        i = k % self.depth
        if self.world == 1:  # one rank's compact buffer IS the frame (bands in order)
            self.frame = self.u8[i][: self.H * self.W * 3].view(self.H, self.W, 3)
            return self.frame
        work, self.pending[i] = self.pending[i], None
        if work is not None:
            work.wait()
        if self.rank != 0:
            return None
        self._frame[i].index_copy_(0, self.index, self._recv[i].view(self.world * self.slot_rows, self.W * 3))
        self.frame = self._frame[i][: self.H].view(self.H, self.W, 3)
        return self.frame


class AccumulatorPipeline:
    """Path-trace passes of a multi-GPU job as SURVEY.md 8(e) lays them out: every rank accumulates the radiance sums
    of its strata for the whole frame; pass k's sums are combined by a reduce-scatter (SUM) that leaves each rank with
    the complete sums of ITS slice of the pixels (1/world of them, padded to equal slices), each rank applies the
    non-linear stages to its slice (`finish`: / pixelSamples, tone map, exposure, gamma, clamp -- RayTracer.cpp:301-315
    -- and, if the caller wants PPM bytes, the byte conversion), and the finished slices are gathered on rank 0.  The
    reduce-scatter runs asynchronously while pass k + 1 is rendered.  Per GPU and pass (world - 1) / world of one fp64
    frame crosses the links in the reduce-scatter and 1 / world of a finished frame in the gather, whatever world is.

    target(k)            the [slice_px * world * 3] fp64 buffer pass k accumulates into (the renderer writes the
                         leading n_pixels * 3 values; the padding stays 0)
    submit(k)            call after pass k's accumulation was enqueued on the current stream
    complete(k, finish)  finish(sums, n_px) -> tensor: `sums` is this rank's reduced slice ([slice_px * 3] fp64, to be
                         resolved in place or into a tensor of any dtype with slice_px * 3 elements), n_px = slice_px.
                         Returns the assembled frame ([n_pixels * 3], finish's dtype) on rank 0, None elsewhere.
                         finish=None gathers the sums themselves."""

    def __init__(self, rank, world, n_pixels, device, depth=2):
        self.rank, self.world, self.depth, self.n_pixels = rank, world, depth, n_pixels
        self.slice_px = (n_pixels + world - 1) // world
        n = self.slice_px * world * 3
        self.accum = [torch.zeros(n, dtype=torch.float64, device=device) for _ in range(depth)]
        self.mine = [torch.zeros(self.slice_px * 3, dtype=torch.float64, device=device) for _ in range(depth)] if world > 1 else None
        self.pending = [None] * depth
        self._recv = {}  # dtype -> per-slot receive buffers on rank 0

    def target(self, k):
        return self.accum[k % self.depth]

    def submit(self, k):
        i = k % self.depth
        if self.world > 1:
            self.pending[i] = dist.reduce_scatter_tensor(self.mine[i], self.accum[i], op=dist.ReduceOp.SUM, async_op=True)

    def complete(self, k, finish=None):
        i = k % self.depth
        if self.world == 1:
            out = finish(self.accum[i], self.slice_px) if finish else self.accum[i]
            return out[: self.n_pixels * 3]
        work, self.pending[i] = self.pending[i], None
        if work is not None:
            work.wait()
        out = finish(self.mine[i], self.slice_px) if finish else self.mine[i]
        assert out.numel() == self.slice_px * 3
        recv = None
        if self.rank == 0:
            if out.dtype not in self._recv:
                self._recv[out.dtype] = [torch.empty(self.world, self.slice_px * 3, dtype=out.dtype, device=out.device)
                                         for _ in range(self.depth)]
            recv = self._recv[out.dtype][i]
        dist.gather(out.view(-1), [recv[q] for q in range(self.world)] if self.rank == 0 else None, dst=0)
        return recv.view(-1)[: self.n_pixels * 3] if self.rank == 0 else None


def reduce_accumulators(accum, world):
    """Sum of the per-rank radiance accumulators on rank 0 (in place): the plain form, for callers that want the sums
    themselves in one place (tests, single passes)."""
    if world > 1:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    return accum
