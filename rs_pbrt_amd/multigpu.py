"""Multi-GPU decomposition of one frame: the Morton-ordered 16x16 tile list of
BlockQueue (src/blockqueue/mod.rs:23-52) is cut into chunks of TILE_CHUNK tiles that are dealt to
the ranks round-robin (TILE_CHUNK = 1: tile by tile — measured on one GPU rendering the 8 shards in turn, tools/c5_shard_balance.py:
chunks of 64 / 16 / 4 / 1 tiles give max / mean shard times of 1.16 / 1.08 / 1.02 / 1.01 on C2 and 1.14 / 1.09 / 1.05 / 1.01 on the
C5 stand-in, at the same mean); every rank renders its tiles into a full-frame film (zero elsewhere) and the
films are SUMMED onto rank 0 — a sum, not a gather, because neighbouring tiles' pixel bounds
overlap (film.rs:321-330).  One process per GPU; torch.distributed is only the transport
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests)."""
TILE_CHUNK = 1


def shard_for_rank(rank, world_size, tile_chunk=TILE_CHUNK):
    """(shard_index, shard_count, tile_chunk) for rspt_render_desc."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return (rank, world_size, tile_chunk)


def reduce_film(film, dst=0):
    """Sum the per-rank film tensors onto `dst` (in place).  film: torch tensor, any device."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def shards_after_failures(rank, world_size, dead, tile_chunk=TILE_CHUNK):
    """Losing GPUs mid-frame ("per-GPU failure => re-queue its tiles on survivors (tiles are idempotent)").
    The frame keeps its decomposition into `world_size` shards; the shards of the ranks in `dead` are handed to the survivors round
    robin, in rank order.  Returns the list of (shard_index, shard_count, tile_chunk) that `rank` renders — its own first — or [] for
    a dead rank.  Every survivor computes the same assignment from (world_size, dead) alone; the films of all listed shards are
    summed (on the rank, then by the usual reduce over the survivors' communicator): shards are disjoint sets of tiles, so the sum
    is the frame whatever the assignment."""
    dead = sorted(set(int(d) for d in dead))
    if any(not 0 <= d < world_size for d in dead):
        raise ValueError("dead rank outside world of %d" % world_size)
    alive = [r for r in range(world_size) if r not in dead]
    if not alive:
        raise ValueError("no surviving rank")
    if rank in dead:
        return []
    mine = [shard_for_rank(rank, world_size, tile_chunk)]
    for k, d in enumerate(dead):
        if alive[k % len(alive)] == rank:
            mine.append((d, world_size, tile_chunk))
    return mine
