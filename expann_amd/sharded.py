"""Row-sharded brute force across the GPUs of one node (one process per GPU).

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
