"""One process per GPU for the all-vs-all entry points (the query-sharded alternative).

The reference's scripts are single-process (FAISS threads inside one process).  Started as they
are -- ``python -m knn_for_homology_amd.cath.search`` -- the mirrors run on one GPU and nothing here imports torch.  Started
under ``python -m torch.distributed.run --nproc-per-node N -m knn_for_homology_amd.cath.search`` (or from a program that has
initialised ``torch.distributed`` itself) the flat searches of ``cath.search``, ``pfam.proteins_search``, ``pfam.search`` and
``pfam.slices.slices_search`` are spread over the N ranks: the database is replicated (CATH 59 MB, Pfam 819 MB), rank r
answers its contiguous slice of the queries with the single-GPU kernels, the slices are concatenated with one
all-gather of D and one of I (RCCL), every rank returns the full arrays and rank 0 alone writes files and prints.  Every
query is answered by one rank: the result is the single-GPU result bit for bit, whatever N.
"""
import os
import sys

from . import faiss


def launched_group():
    """(rank, world): (0, 1) for a plain ``python`` run -- without touching torch."""
    td = sys.modules.get("torch.distributed")
    if td is not None and td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size()
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or "RANK" not in os.environ:
        return 0, 1
    from .sharded import launched_group as _join
    return _join()


def flat_index(d: int, metric):
    """``faiss.IndexFlat(d, metric)`` on one GPU; under a multi-rank launch a ``QueryShardedFlatIndex`` with the same
    ``train / add / normalize_rows / search / search_self / reconstruct_into`` surface."""
    if launched_group()[1] == 1:
        return faiss.IndexFlat(d, metric)
    from .sharded import QueryShardedFlatIndex
    return QueryShardedFlatIndex(d, metric)


def writer() -> bool:
    """True on the rank that writes the output files and prints (rank 0; the only rank of a plain run)."""
    return launched_group()[0] == 0


def barrier():
    if launched_group()[1] > 1:
        import torch.distributed as dist
        dist.barrier()


def same_everywhere(obj):
    """rank 0's value of ``obj`` on every rank (decisions taken from the file system, e.g. "which outputs exist")"""
    if launched_group()[1] == 1:
        return obj
    import torch.distributed as dist
    box = [obj]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def rank0_only(work):
    """``work()`` on rank 0 alone (HNSW and LSH do not shard: SURVEY.md 8(e), replicas only); the other ranks wait for
    its OUTCOME, not at a bare barrier: rank 0 broadcasts "done" or its error text, and a failure raises on every rank
    instead of leaving the others in a collective until its timeout.  Returns work()'s value on rank 0, None elsewhere."""
    rank, world = launched_group()
    if world == 1:
        return work()
    result, failure, text = None, None, None
    if rank == 0:
        try:
            result = work()
        except BaseException as e:  # noqa: BLE001 -- the peers must hear about it whatever it is
            failure, text = e, f"{type(e).__name__}: {e}"
    text = same_everywhere(text)
    if failure is not None:
        raise failure
    if text is not None:
        raise RuntimeError(f"rank 0 failed: {text}")
    return result
