"""Row-stripe partition of one frame over ranks, and the single gather that joins it.

Pixels are independent (the RNG is keyed by the absolute pixel index), so any partition
reproduces the single-GPU frame byte for byte; there is no exchange during rendering.
Stripes are contiguous bands of MEMORY rows (the bottom-up framebuffer of
Raytracer.cpp:64), so concatenating the ranks' bands in rank order IS the final image.

The only collective is the gather of the packed ARGB bands to rank 0.  With equal bands it
is one `dist.gather` received in place; with cost-balanced (unequal) bands it is one
`dist.gather` of the tallest band's height from every rank (views of the frames: no copy on
the sending side, buffers made once — GatherBuffers) and an unpack on rank 0, or one grouped
isend/irecv batch — on RCCL a single ncclGroupStart/End of point-to-point transfers (how
ncclGather itself is built), each peer using its own xGMI link to rank 0, received in place
into rank 0's framebuffer.

Plumbing over torch.distributed; no compute here.
"""
from typing import List, Optional, Sequence, Tuple


def partition_rows(height: int, world: int, row_cost: Optional[Sequence[float]] = None,
                   align: int = 1) -> List[Tuple[int, int]]:
    """Split memory rows [0,height) into `world` contiguous bands.

    Without costs: first `height % world` ranks take one extra row.  With `row_cost`
    (len == height, cost of memory row i): bands of (nearly) equal total cost, each at
    least one row, boundaries rounded to `align` rows where possible.
    """
    if world < 1 or height < world:
        raise ValueError("need 1 <= world <= height")
    if row_cost is None:
        q, r = divmod(height, world)
        out, a = [], 0
        for k in range(world):
            b = a + q + (1 if k < r else 0)
            out.append((a, b))
            a = b
        return out
    if len(row_cost) != height:
        raise ValueError("row_cost must have one entry per memory row")
    total = float(sum(row_cost))
    prefix = [0.0]
    for c in row_cost:
        prefix.append(prefix[-1] + max(float(c), 0.0))
    bounds = [0]
    for k in range(1, world):
        target = total * k / world
        # first row index whose prefix cost reaches the target
        lo, hi = bounds[-1] + 1, height - (world - k)
        i = lo
        while i < hi and prefix[i] < target:
            i += 1
        if align > 1:
            j = int(round(i / align)) * align
            if lo <= j <= hi:
                i = j
        bounds.append(min(max(i, lo), hi))
    bounds.append(height)
    return [(bounds[k], bounds[k + 1]) for k in range(world)]


class GatherBuffers:
    """What the padded gather needs besides the frame, allocated ONCE per (frame, bands) and reused by every step
    (round 3 allocated and zeroed them inside every timed step).

    Every rank sends `rows` = the tallest band's height: a VIEW of its own frame that starts at its band (or ends at the
    frame's last row when the band lies too close to it) — no copy, no padding buffer, nothing to zero: the rows behind
    the band are whatever the frame holds there and are never read.  Rank `dst` receives into `stage` (world x rows x W)
    and copies every other rank's band out of it into the frame.
    """

    def __init__(self, frame, bands, rank, world, dst=0):
        import torch

        self.rows = max(b - a for a, b in bands)
        height = frame.shape[0]
        # where rank k's send view starts, and where its band lies inside it
        self.start = [min(a, height - self.rows) for a, _ in bands]
        self.offset = [a - s for (a, _), s in zip(bands, self.start)]
        self.send = frame[self.start[rank]: self.start[rank] + self.rows]
        self.stage = [torch.empty((self.rows, frame.shape[1]), dtype=frame.dtype, device=frame.device) for _ in range(world)] if rank == dst else None
        self.key = (frame.data_ptr(), tuple(tuple(b) for b in bands), rank, world, dst)


def gather_bands(frame, bands: Sequence[Tuple[int, int]], rank: int, world: int, dist, dst: int = 0,
                 method: str = "p2p", buffers: Optional[GatherBuffers] = None):
    """Join the ranks' bands on rank `dst`.

    frame: torch tensor [H, W] (uint32 viewed as int32), same shape on every rank; rank k has
    rendered rows bands[k].  After the call rank `dst`'s frame holds every band.  One
    collective: dist.gather for equal bands; for unequal bands either one batched isend/irecv
    received in place (method "p2p") or one dist.gather of `rows` = the tallest band's rows from
    every rank followed by an unpack on `dst` (method "padded"; pass `buffers` = GatherBuffers(...)
    made once, or they are made per call).
    """
    if world == 1:
        return
    if method == "padded" and len({b - a for a, b in bands}) > 1:
        key = (frame.data_ptr(), tuple(tuple(b) for b in bands), rank, world, dst)
        if buffers is None or buffers.key != key:
            buffers = GatherBuffers(frame, bands, rank, world, dst)
        if rank == dst:
            dist.gather(buffers.send, gather_list=buffers.stage, dst=dst)
            for k, (x, y) in enumerate(bands):
                if k != dst:
                    o = buffers.offset[k]
                    frame[x:y].copy_(buffers.stage[k][o: o + (y - x)])
        else:
            dist.gather(buffers.send, gather_list=None, dst=dst)
        return
    sizes = {b - a for a, b in bands}
    if len(sizes) == 1:
        a, b = bands[rank]
        if rank == dst:
            outs = [frame[x:y] for x, y in bands]  # views into the final image: received in place
            dist.gather(frame[a:b], gather_list=outs, dst=dst)
        else:
            dist.gather(frame[a:b], gather_list=None, dst=dst)
        return
    ops = []
    if rank == dst:
        for k, (a, b) in enumerate(bands):
            if k != dst:
                ops.append(dist.P2POp(dist.irecv, frame[a:b], k))
    else:
        a, b = bands[rank]
        ops.append(dist.P2POp(dist.isend, frame[a:b], dst))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
