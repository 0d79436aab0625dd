"""Static sharding of candidate sites over the GPUs of one node and the final result gather.

The reference is single-device (PileupModel/predict.py:208); every site is independent in
encode, feature reduction and forward, so rank r of R owns the contiguous range
[r*N/R, (r+1)*N/R) of the position-sorted site list, weights are replicated, and the only
exchange is one rooted gather of the per-site results (RCCL over xGMI when the backend is
"nccl", gloo in the CPU tests).  Result order = rank order = position order, so the merge is a
concatenation.
"""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(n_items, rank, world, align=1):
    """Contiguous, balanced, order-preserving: [lo, hi) of rank; sizes differ by at most one.  align > 1: the cuts fall on
    multiples of align (the items are dealt in units of align, the last unit may be short) - with align = the predict loop's
    batch size every rank owns whole batches and formats its rows without knowing anything of the other ranks'."""
    if world <= 0 or not (0 <= rank < world) or n_items < 0 or align < 1:
        raise ValueError("bad shard arguments")
    units = -(-n_items // align)
    base, rem = divmod(units, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return min(lo * align, n_items), min(hi * align, n_items)


def shard_columns(n_cols, rank, world, halo=16):
    """Column ranges for the encode stage: the owned range plus a halo of `halo` columns each side
    (re-computed, not exchanged) so that every window centred in the owned range is complete."""
    lo, hi = shard_range(n_cols, rank, world)
    return max(0, lo - halo), min(n_cols, hi + halo), lo, hi


def _rooted_gather(local, counts, root, group):
    """Rooted gather of contiguous row blocks straight into place: the root posts one receive per peer into ITS slice of one
    preallocated [sum(counts), ...] buffer, every other rank posts one send; the operations go out as one batch (a grouped
    ncclSend / ncclRecv on RCCL, so every peer uses its own xGMI link to the root; plain point-to-point on gloo).  No
    padding to the longest shard, no list of world-many staging buffers, no concatenation."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    assert local.shape[0] == counts[rank], (tuple(local.shape), counts[rank])
    local = local.contiguous()
    # A rank with an empty block posts nothing below.  On the NCCL / RCCL backend batch_isend_irecv must not be the FIRST
    # operation of a group on only a subset of its ranks (the communicator is created lazily by a call every rank makes), so
    # when some rank sits out, every rank - they all hold the same `counts` - first meets in one tiny all_reduce.
    if any(c == 0 for r, c in enumerate(counts) if r != root):
        token = torch.zeros(1, dtype=torch.int32, device=local.device)
        dist.all_reduce(token, group=group)
    ops, out = [], None
    if rank == root:
        out = torch.empty((sum(counts),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        off = 0
        for r, c in enumerate(counts):
            if r == root:
                out[off:off + c].copy_(local)
            elif c > 0:
                ops.append(dist.P2POp(dist.irecv, out[off:off + c], peer(r), group))
            off += c
    elif counts[rank] > 0:
        ops.append(dist.P2POp(dist.isend, local, peer(root), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def gather_results(local, n_total, root=0, group=None):
    """Gathers per-site result rows (a tensor [n_local, ...]) of the shard_range partition of `n_total` sites to `root` in rank
    order.  Returns the [n_total, ...] tensor on root, None elsewhere.  One batch of point-to-point transfers into the
    root's preallocated buffer, no ring all-reduce."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    counts = [hi - lo for lo, hi in (shard_range(n_total, r, world) for r in range(world))]
    return _rooted_gather(local, counts, root, group)


def gather_varlen(local, root=0, group=None):
    """Rooted gather of per-rank row blocks of DIFFERENT lengths ([n_r, k] tensors, e.g. the candidate sites a rank found in
    its column shard) in rank order: the sizes travel first (one small all_gather), then the blocks as in gather_results.
    Returns the concatenation on root, None elsewhere; the input itself without an initialised process group."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    return _rooted_gather(local, [int(s.item()) for s in sizes], root, group)


def site_offsets(n_local, device="cpu", group=None):
    """(first, n_total): where this rank's n_local rows start in the rank-ordered list of every rank's rows, and the length of that
    list (one small all_gather)."""
    import torch
    import torch.distributed as dist
    n = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(dist.get_world_size(group))]
    dist.all_gather(sizes, n, group=group)
    counts = [int(s_.item()) for s_ in sizes]
    return sum(counts[:dist.get_rank(group)]), sum(counts)


def batch_heads(gt_arg, first, n_total, batch_size, device="cpu", group=None):
    """uint8 [ceil(n_total / batch_size), 10]: the first ten genotype argmax values of every batch of `batch_size` rows of the
    rank-ordered list - all a pileup.vcf row takes from the other rows of its batch (PileupModel/predict.py:102-125 index
    gt_output[ti], ti < 10).  gt_arg: this rank's values, rows [first, first + len) of the list.  Every rank fills in the
    entries it owns (10 of every batch_size rows) and one small all_reduce (10 B per batch) adds the tables up; with it a
    rank formats its rows exactly as the single process would (host.vcf_format_batches(first=, n_total=, heads=)) and the
    TEXT travels to the root instead of the calls."""
    import numpy as np
    import torch
    import torch.distributed as dist
    nb = max(1, -(-int(n_total) // int(batch_size)))            # (never an empty collective)
    heads = np.zeros((nb, 10), np.uint8)
    n = len(gt_arg)
    if n:
        kb = np.arange(first // batch_size, (first + n - 1) // batch_size + 1, dtype=np.int64)       # the batches with rows here
        g = kb[:, None] * batch_size + np.arange(min(10, int(batch_size)), dtype=np.int64)[None, :]  # their first ten rows, global
        own = (g >= first) & (g < first + n)
        heads[(g // batch_size)[own], (g % batch_size)[own]] = np.asarray(gt_arg, np.uint8)[g[own] - first]
    t = torch.from_numpy(heads).to(device)
    dist.all_reduce(t, group=group)
    return t.cpu().numpy()


def gather_text(text, device="cpu", root=0, group=None):
    """Rooted gather of every rank's bytes in rank order (gather_varlen over uint8): a memoryview of the concatenation on root,
    None elsewhere."""
    import numpy as np
    import torch
    arr = np.frombuffer(text, np.uint8) if len(text) else np.zeros(0, np.uint8)
    out = gather_varlen(torch.from_numpy(arr.copy() if not arr.flags.writeable else arr).to(device), root, group)
    return None if out is None else memoryview(out.cpu().numpy())


def gather_results_abi(ctx, local, n_total, root=0, stream=None):
    """The same rooted gather through the library's own RCCL entry (nsnp_comm_init + nsnp_gather_results, include/nanosnp.h) instead of
    a torch.distributed collective: grouped ncclSend / ncclRecv straight into place on the root, no padded copies.  The 128-byte
    communicator id travels over the existing process group once per context.  EXPERIMENTAL: validated on hardware at world size 1
    only (the development pool has one-GPU boxes); bench.py uses it with --gather rccl-abi."""
    import math
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if not hasattr(ctx, "_comm"):
        ids = [ctx.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        ctx.comm_init(ids[0], rank, world)
    row_bytes = local.element_size() * math.prod(local.shape[1:])
    counts = [(hi - lo) * row_bytes for lo, hi in (shard_range(n_total, r, world) for r in range(world))]
    out = ctx.gather_bytes(local.contiguous(), counts, root=root, stream=stream)
    if rank != root:
        return None
    return out.view(local.dtype).view((n_total,) + tuple(local.shape[1:]))
