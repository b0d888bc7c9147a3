"""Sharded brute force across the GPUs of one node (one process per GPU): `ShardedSearch` cuts
the base into row shards; `GridShardedSearch` cuts rows x queries (the default of bench.py).

The base set is cut into contiguous row ranges: rank r owns rows
[r*n/G, (r+1)*n/G).  Every rank scans its shard for all queries (no data-path collective),
the fixed-size per-shard results [m][k] of (score, global id) are exchanged with ONE
all-gather per array (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box), and
every rank merges the G sorted lists per query with the reference's (score, id) order.
Because shard ranges are contiguous and ids are global, the merged result is bit-identical
to the unsharded one.

This module only orchestrates: `local_search` and `merge` are the C-ABI calls
(expann_search_device / expann_merge_topk_device) in production.  Tests inject CPU
stand-ins to exercise the orchestration over gloo without a GPU; the product has no CPU path.
"""


def shard_range(n, rank, world):
    """Contiguous row range [lo, hi) of `rank` out of `world` shards of an n-row base."""
    return rank * n // world, (rank + 1) * n // world


def ceil_shard_range(n, rank, world):
    """SURVEY 8e's partition: rank r holds rows [r * ceil(n/G), min(n, (r+1) * ceil(n/G))) -- the one
    expann_sharded_build (csrc/expann_sharded.hip) cuts too.  Trailing shards may be empty."""
    per = (n + world - 1) // world
    return min(n, rank * per), min(n, (rank + 1) * per)


class ShardedSearch:
    def __init__(self, dist, world, local_search, merge, alloc_gather):
        """dist: torch.distributed (or None when world == 1);
        local_search(queries, k) -> (ids[m,k], dists[m,k]) tensors of this shard (global ids);
        merge(all_ids[G,m,k], all_dists[G,m,k]) -> (ids[m,k], dists[m,k]);
        alloc_gather(t) -> tensor shaped [G * t.shape[0], *t.shape[1:]] on t's device (the
        concatenated-along-dim-0 layout every backend accepts for all_gather_into_tensor)."""
        self.dist, self.world = dist, world
        self.local_search, self.merge, self.alloc_gather = local_search, merge, alloc_gather

    def search(self, queries, k):
        ids, dists = self.local_search(queries, k)
        if self.world == 1:
            return ids, dists
        all_ids = self.alloc_gather(ids)
        all_d = self.alloc_gather(dists)
        self.dist.all_gather_into_tensor(all_ids, ids)
        self.dist.all_gather_into_tensor(all_d, dists)
        G = self.world
        return self.merge(all_ids.view(G, *ids.shape), all_d.view(G, *dists.shape))


def shard_grid(world, row_shards=None):
    """(row_shards, query_groups) of a `world`-rank job.  Default: pure row sharding (SURVEY 8e,
    BASELINE's north_star): `world` row shards, every rank searches every query.  With
    row_shards = R < world the remaining factor splits the QUERIES: per-query costs (conversion,
    threshold pass, selection, result exchange) then shrink with the group count instead of being
    repeated on every rank, and the all-gather of per-shard top-k runs between R ranks."""
    if row_shards is None:
        row_shards = world
    if world % row_shards:
        raise ValueError(f"row_shards {row_shards} does not divide world {world}")
    return row_shards, world // row_shards


def chunk_bytes(rows, k):
    """bytes of one rank's exchange chunk: [ids: rows*k uint64 | dists: rows*k fp32], padded to 16."""
    return (rows * k * 12 + 15) // 16 * 16


def unpack_chunk(chunk, rows, k):
    """(ids[rows,k] int64 view, dists[rows,k] float32 view) of a uint8 chunk tensor."""
    import torch
    ids = chunk[:rows * k * 8].view(torch.int64).view(rows, k)
    dists = chunk[rows * k * 8:rows * k * 12].view(torch.float32).view(rows, k)
    return ids, dists


class GridShardedSearch:
    """rank r = query group (r // R) x row shard (r % R), R = row shards.

    search(): (1) every rank scans ITS rows for ITS query slice (no collective) and leaves the
    [slice][k] ids and distances in ONE chunk `[ids | dists]`; (2) the R ranks of a query group
    all-gather their chunks (one collective) and merge them (expann_merge_topk_strided_device
    reads the lists where they landed); (3) the merged chunks are all-gathered across the query
    groups (one collective), so every rank ends with the full [m][k] result.  Slices are padded
    to ceil(m / groups) queries for the fixed-size collectives."""

    def __init__(self, dist, world, rank, row_shards, local_search, merge, alloc):
        """local_search(queries_slice, k, chunk): this rank's rows, results (global ids) into the
        uint8 chunk (rows beyond the slice keep their padding); merge(gathered, n_lists, rows, k,
        out_chunk): `gathered` holds n_lists chunks back to back; alloc(name, nbytes, like) ->
        cached uint8 tensor on like's device."""
        self.dist, self.world, self.rank = dist, world, rank
        self.R, self.Q = shard_grid(world, row_shards)
        self.row_idx, self.qgroup = rank % self.R, rank // self.R
        self.local_search, self.merge, self.alloc = local_search, merge, alloc
        self.row_group = self.col_group = None
        if dist is not None and world > 1:
            # every rank creates every group, in the same order (torch.distributed contract)
            for g in range(self.Q):
                grp = dist.new_group([g * self.R + i for i in range(self.R)]) if self.R > 1 else None
                if g == self.qgroup:
                    self.row_group = grp
            for i in range(self.R):
                grp = dist.new_group([g * self.R + i for g in range(self.Q)]) if self.Q > 1 else None
                if i == self.row_idx:
                    self.col_group = grp

    def query_slice(self, m):
        return shard_range(m, self.qgroup, self.Q)

    def search(self, queries, k):
        import torch
        m = queries.shape[0]
        lo, hi = self.query_slice(m)
        pad = (m + self.Q - 1) // self.Q
        cb = chunk_bytes(pad, k)
        mine = self.alloc("mine", cb, queries)
        self.local_search(queries[lo:hi], k, mine)
        if self.R > 1:
            gathered = self.alloc("row_gather", self.R * cb, queries)
            self.dist.all_gather_into_tensor(gathered, mine, group=self.row_group)
            merged = self.alloc("merged", cb, queries)
            self.merge(gathered, self.R, pad, k, merged)
            mine = merged
        if self.Q == 1:
            ids, dists = unpack_chunk(mine, pad, k)
            return ids[:m], dists[:m]
        full = self.alloc("full", self.Q * cb, queries)
        self.dist.all_gather_into_tensor(full, mine, group=self.col_group)
        parts = [shard_range(m, g, self.Q) for g in range(self.Q)]
        views = [unpack_chunk(full[g * cb:(g + 1) * cb], pad, k) for g in range(self.Q)]
        ids = torch.cat([v[0][:b - a] for v, (a, b) in zip(views, parts)], 0)
        dists = torch.cat([v[1][:b - a] for v, (a, b) in zip(views, parts)], 0)
        return ids, dists


def slice_range(m, j, world):
    """query slice j of exchange pattern 2 (csrc/expann_sharded.hip, expann_sharded_slice):
    [min(m, j * ceil(m/G)), min(m, (j+1) * ceil(m/G)))"""
    per = (m + world - 1) // world
    return min(m, j * per), min(m, (j + 1) * per)


class SliceShardedSearch:
    """Pure row sharding with exchange pattern 2 through torch.distributed -- the Python mirror of what
    expann_sharded_search_device does behind the C ABI: (1) every rank scans ITS rows for ALL queries;
    (2) an all-to-all sends rank j the slice-j rows of every rank's [m][k] result (list g of my slice comes
    from rank g); (3) every rank merges its slice -- m/G queries, G lists; (4) the merged slices are
    all-gathered (ragged: the last slices may be short or empty), so every rank ends with the full result.
    Per rank 2 (G-1)/G x 12 m k bytes arrive instead of the (G-1) x 12 m k of one all-gather of whole
    chunks, and each rank merges m/G queries instead of m."""

    def __init__(self, dist, world, rank, local_search, merge_slices, alloc):
        """local_search(queries, k, chunk): this rank's rows, all queries, results (global ids) into the
        uint8 chunk [ids m*k | dists m*k]; merge_slices(lists_ids[G*per, k], lists_d[G*per, k], n_lists,
        per, cnt, k, out_ids[cnt, k], out_d[cnt, k]): list g starts at row g*per; alloc(name, nbytes, like)."""
        self.dist, self.world, self.rank = dist, world, rank
        self.local_search, self.merge_slices, self.alloc = local_search, merge_slices, alloc

    def _exchange(self, pairs):
        """pairs[j] = (tensor sent to rank j, tensor received from rank j); the own part is a copy"""
        # (gloo moves host memory only: device tensors are staged through the host in stream order --
        # the rehearsal on a one-GPU box; RCCL sends and receives device memory directly)
        stage = self.dist.get_backend() == "gloo"
        ops, landed = [], []
        for j, (snd, rcv) in enumerate(pairs):
            if j == self.rank:
                if rcv.numel():
                    rcv.copy_(snd)
                continue
            if snd.numel():
                ops.append(self.dist.P2POp(self.dist.isend, snd.cpu() if stage and snd.is_cuda else snd.contiguous(), j))
            if rcv.numel():
                if stage and rcv.is_cuda:
                    import torch
                    host = torch.empty(rcv.shape, dtype=rcv.dtype)
                    landed.append((rcv, host))
                    ops.append(self.dist.P2POp(self.dist.irecv, host, j))
                else:
                    ops.append(self.dist.P2POp(self.dist.irecv, rcv, j))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for rcv, host in landed:
            rcv.copy_(host)

    def search(self, queries, k):
        m, G, r = queries.shape[0], self.world, self.rank
        mine = self.alloc("mine", chunk_bytes(m, k), queries)
        self.local_search(queries, k, mine)
        ids, dists = unpack_chunk(mine, m, k)
        if G == 1:
            return ids, dists
        per = (m + G - 1) // G
        lo, hi = slice_range(m, r, G)
        cnt = hi - lo
        lists = self.alloc("lists", chunk_bytes(G * per, k), queries)
        l_ids, l_d = unpack_chunk(lists, G * per, k)
        rng = [slice_range(m, j, G) for j in range(G)]
        for src, dst in ((ids, l_ids), (dists, l_d)):
            self._exchange([(src[a:b], dst[j * per:j * per + cnt]) for j, (a, b) in enumerate(rng)])
        merged = self.alloc("merged", chunk_bytes(per, k), queries)
        m_ids, m_d = unpack_chunk(merged, per, k)
        if cnt:
            self.merge_slices(l_ids, l_d, G, per, cnt, k, m_ids[:cnt], m_d[:cnt])
        full = self.alloc("full", chunk_bytes(m, k), queries)
        f_ids, f_d = unpack_chunk(full, m, k)
        for src, dst in ((m_ids, f_ids), (m_d, f_d)):
            self._exchange([(src[:cnt], dst[a:b]) for (a, b) in rng])
        return f_ids, f_d
