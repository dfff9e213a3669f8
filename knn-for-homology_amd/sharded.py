"""Row-sharded flat index: one process per GPU, one RCCL all-gather per search.

The reference has no multi-device code; this is the MI355X-native
addition the north star asks for.  Rank r owns database rows [lo_r, hi_r) -- contiguous
slices in insertion order, so global ids are ``row_offset + local row``.  A search

  1. scans the local shard for ALL queries with the same fused kernel as the single-GPU
     path, producing k packed keys per query (uint64: order-preserving score bits << 32 |
     global row id; ascending key order == best first, ties -> lower global id),
  2. exchanges them with one ``all_gather_into_tensor`` ([world, nq, k] x 8 bytes --
     nq=32, k=100: 25.6 KB per rank, latency-bound over xGMI),
  3. merges the ``world`` sorted lists per query on every rank (same merge kernel that
     combines the per-chunk lists inside one GPU), so the result does not depend on the
     number of shards.

``submit`` enqueues a search and returns at once; searches submitted back to back alternate
between two lanes of the backend (the index and a read-only view of it, each with its own
stream and scratch memory) and overlap on the GPU: the seed sample, the merges and the
all-gather of one search run beside the scan of the other.  ``search_dev`` = submit + wait.

torch is used for device memory, streams and torch.distributed only.
"""
import ctypes
import datetime
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from . import faiss as _faiss


def launched_group():
    """(rank, world) of the process group the entry points spread their all-vs-all searches over.

    A caller that has initialised ``torch.distributed`` itself is taken at its word.  Otherwise, under
    ``python -m torch.distributed.run`` (RANK / WORLD_SIZE / LOCAL_RANK in the environment, WORLD_SIZE > 1) this is
    where the launch becomes one process per GPU: the rank takes GPU LOCAL_RANK (library and torch) and joins the default
    group over RCCL (``nccl``).  KNN355_REHEARSE_ONE_GPU=1 puts every rank on GPU 0 with gloo -- a functional rehearsal
    on a one-GPU box.  A plain ``python -m ...`` run is (0, 1): nothing is initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or "RANK" not in os.environ:
        return 0, 1
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = os.environ.get("KNN355_REHEARSE_ONE_GPU") == "1"
    device = 0 if rehearse else local
    torch.cuda.set_device(device)
    _lib.check(_lib.lib().knn_init(device))
    # (a rank that waits for rank 0's HNSW build or for a slow peer must outlive the 10-minute default)
    timeout = datetime.timedelta(seconds=float(os.environ.get("KNN355_DIST_TIMEOUT_S", "7200")))
    if rehearse:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timeout)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device), timeout=timeout)
    return rank, world


def shard_bounds(n_total: int, world: int, rank: int):
    """Contiguous row range [lo, hi) of ``rank``: ceil(n/world) rows per shard."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


class HipShardBackend:
    """Local shard scan + key merge on the rank's GPU through the C ABI.

    Two *lanes*: the index itself and a read-only view of it (``knn_flat_view``: same rows,
    own scratch memory), each with its own HIP stream.  Consecutive searches alternate
    between them, so the small launches at either end of one search (seed sample, merges,
    the all-gather) run beside the other search's scan instead of leaving the GPU idle."""

    def __init__(self, d, metric):
        self.index = _faiss.IndexFlat(d, metric)
        self.metric = metric
        self.device = torch.device("cuda", int(_lib.lib().knn_device_of(self.index._h)))
        self._lanes = None
        self._turn = 0

    def _invalidate(self):
        self._lanes = None  # views see the rows present when they were made

    def next_lane(self):
        """(index handle object, torch stream) of the lane whose turn it is."""
        if self._lanes is None:
            # different priorities = different hardware queues: two streams of one priority can land on the same
            # queue (the runtime multiplexes its streams over a few), and then the lanes run strictly one after
            # the other -- measured: no overlap at all, two lanes = one lane
            prio = [int(v) for v in os.environ.get("KNN355_LANE_PRIORITIES", "0,-1").split(",")]
            self._lanes = [(self.index if i == 0 else self.index.view(), torch.cuda.Stream(self.device, priority=pr)) for i, pr in enumerate(prio)]
        lane = self._lanes[self._turn % len(self._lanes)]
        self._turn = (self._turn + 1) % len(self._lanes)
        return lane

    def reserve(self, n):
        _lib.check(_lib.lib().knn_flat_reserve(self.index._h, n))

    def add_dev(self, x: torch.Tensor):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] == self.index.d
        self._invalidate()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.lib().knn_flat_add_dev(self.index._h, x.data_ptr(), x.shape[0], ctypes.c_void_p(stream)))

    def add(self, x: np.ndarray):
        self._invalidate()
        self.index.add(x)

    @property
    def ntotal(self):
        return self.index.ntotal

    # the three device steps; `index` picks the lane, the launches go to torch's CURRENT stream
    def search_keys(self, q: torch.Tensor, k: int, id_base: int, index=None, out=None) -> torch.Tensor:
        """k packed keys per query; ``out``: a contiguous [nq, k] int64 tensor to write them into (the front of the
        all-gather's send buffer)"""
        nq = q.shape[0]
        keys = out if out is not None else torch.empty((nq, k), dtype=torch.int64, device=q.device)
        assert keys.is_contiguous() and keys.shape == (nq, k) and keys.dtype == torch.int64
        cur = torch.cuda.current_stream(q.device)
        stream = cur.cuda_stream
        if stream == 0:
            cur.synchronize()  # (see search(): the NULL stream means the library's own stream)
        _lib.check(_lib.lib().knn_flat_search_keys_dev((index or self.index)._h, q.data_ptr(), nq, k, id_base, keys.data_ptr(),
                                                       ctypes.c_void_p(stream)))
        return keys

    def search(self, q: torch.Tensor, k: int, index=None):
        nq = q.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        cur = torch.cuda.current_stream(q.device)
        stream = cur.cuda_stream
        if stream == 0:
            # torch's default stream IS the NULL stream, and the C ABI reads NULL as "use the library's own
            # (non-blocking) stream and synchronise it": nothing orders that stream behind torch kernels still
            # producing q (a normalisation, a gather, the copy inside .contiguous()).  Drain them first; the call
            # below is synchronous anyway (and may then use the statistical seed and its verification).
            cur.synchronize()
        _lib.check(_lib.lib().knn_flat_search_dev((index or self.index)._h, q.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(),
                                                  ctypes.c_void_p(stream)))
        return D, I

    def search_self(self, k: int, row0: int, nrows: int):
        """rows [row0, row0 + nrows) of the index as queries against all of it: host (D, I), nothing uploaded"""
        return self.index.search_self(k, row0, nrows)

    def normalize_rows(self):
        self.index.normalize_rows()

    def merge(self, gathered: torch.Tensor, nlists: int, nq: int, k: int, index=None):
        D = torch.empty((nq, k), dtype=torch.float32, device=gathered.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=gathered.device)
        cur = torch.cuda.current_stream(gathered.device)
        stream = cur.cuda_stream
        if stream == 0:
            cur.synchronize()
        # the lane's own handle: its merge scratch is private to the lane's stream
        _lib.check(_lib.lib().knn_merge_keys_dev((index or self.index)._h, gathered.data_ptr(), nlists,
                                                 nq, k, D.data_ptr(), I.data_ptr(), ctypes.c_void_p(stream)))
        return D, I


class ShardSearchError(RuntimeError):
    """A sharded search failed on at least one rank; raised on EVERY rank (``failed_ranks``: who)."""

    def __init__(self, failed_ranks, local=None):
        self.failed_ranks = list(failed_ranks)
        msg = f"sharded search failed on rank(s) {self.failed_ranks}"
        if local is not None:
            msg += f"; this rank: {type(local).__name__}: {local}"
        super().__init__(msg)


class PendingSearch:
    """Result of ``ShardedFlatIndex.submit``: ``result()`` makes the caller's current stream
    wait for the search and hands out (D, I).

    A search over several ranks carries a status row through its all-gather (see
    ``ShardedFlatIndex._search_on_current_stream``): ``result()`` reads it (k x 8 bytes to the host, so the call waits
    for the search) and raises ``ShardSearchError`` on every rank if any rank's local scan failed.  ``result(check=False)``
    only orders the streams; ``check()`` can be called later, from any stream: it always waits for the search's own event
    first.  The status row has k slots: ``failed_ranks`` lists at most k ranks (the lowest ones)."""

    def __init__(self, D, I, event=None, status=None, local_error=None):
        self._D, self._I, self._event = D, I, event
        self._status_event = event  # (kept until the status row has been read: result() may run on another stream than check())
        self._status, self._local_error = status, local_error

    def check(self):
        if self._status is not None:
            if self._status_event is not None and self._status.is_cuda:
                # whatever stream is current now was not necessarily ordered behind the lane's stream by result()
                torch.cuda.current_stream(self._status.device).wait_event(self._status_event)
            st = self._status.cpu()
            self._status = None
            self._status_event = None
            failed = [int(r) for r in st.tolist() if r >= 0]
            if failed or self._local_error is not None:
                raise ShardSearchError(failed, self._local_error) from self._local_error

    def result(self, check=True):
        if self._event is not None:
            cur = torch.cuda.current_stream(self._D.device)
            cur.wait_event(self._event)
            self._D.record_stream(cur)
            self._I.record_stream(cur)
            if self._status is not None:
                self._status.record_stream(cur)
            self._event = None
        if check:
            self.check()
        return self._D, self._I


class ShardedFlatIndex:
    """faiss.IndexFlat whose rows are split over the ranks of a process group."""

    def __init__(self, d, metric=_faiss.METRIC_L2, rank=None, world=None, row_offset=0, group=None, backend=None):
        self.d, self.metric_type, self.group = d, metric, group
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world = rank, world
        self.row_offset = int(row_offset)
        # exercise the keys -> all-gather -> merge path even with one rank (tests)
        self.force_collective = os.environ.get("KNN355_FORCE_COLLECTIVE", "0") == "1"
        self.backend = backend if backend is not None else DEFAULT_BACKEND(d, metric)
        # a list here makes every search append a (start, end) pair of timing events recorded on the lane's
        # stream around its all-gather (bench.py reports their mean)
        self.collective_events = None
        self.step_events = None  # a list here makes submit() record a timing event behind every search
        self.unchecked = []  # searches of search_dev(check=False) whose status rows nobody has read yet

    @property
    def local(self):
        return self.backend.index

    @property
    def ntotal_local(self):
        return self.backend.ntotal

    def reserve(self, n):
        self.backend.reserve(n)

    def add_dev(self, x):
        self.backend.add_dev(x)

    def add(self, x):
        """Adds THIS rank's rows (global ids row_offset + insertion order)."""
        self.backend.add(x)

    def _search_on_current_stream(self, q, k, index=None):
        """-> (D, I, status, local_error).  Several ranks: the all-gather's send buffer is [nq + 1, k] keys -- the rank's
        k best keys per query and one STATUS row, all padding on a healthy rank.  A rank whose local scan raises (out of
        memory, a stale view, a bad argument -- anything the C ABI reports) still enters the collective, with "no rows"
        for keys and its rank number as the one key of its status row: its peers neither hang in the all-gather nor
        return a result that silently lacks a shard.  The merge treats the status row as one more query, so its output
        row lists the failed ranks (ids, ascending; -1 = none): ``status``.  The same rule as the library's own
        ``knn_sharded_search_dev`` (enter the collective with padding, report afterwards)."""
        kw = {"index": index} if index is not None else {}
        if self.world == 1 and not self.force_collective:
            D, I = self.backend.search(q, k, **kw)
            return D, I, None, None
        nq = q.shape[0]
        dev = getattr(self.backend, "device", q.device)
        send = torch.full((nq + 1, k), -1, dtype=torch.int64, device=dev)  # (-1 = KEY_PAD: no key)
        local_error = None
        try:
            self.backend.search_keys(q, k, self.row_offset, out=send[:nq], **kw)
        except Exception as e:  # noqa: BLE001 -- whatever it is, the peers are about to enter the collective
            local_error = e
            send[:nq].fill_(-1)
            send[nq, 0] = self.rank  # (score word 0, id = rank: sorts in front of any padding)
        # rank-major concatenation along dim 0 == [world, nq + 1, k]
        gathered = torch.empty((self.world * (nq + 1), k), dtype=torch.int64, device=send.device)
        timed = self.collective_events is not None and send.is_cuda
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        dist.all_gather_into_tensor(gathered, send, group=self.group)
        if timed:
            ev1.record()
            self.collective_events.append((ev0, ev1))
        D, I = self.backend.merge(gathered.view(self.world, nq + 1, k), self.world, nq + 1, k, **kw)
        return D[:nq], I[:nq], I[nq], local_error

    def submit(self, q, k) -> PendingSearch:
        """Enqueues one search of q ([nq, d] float32, identical on every rank) and returns at once.
        Searches submitted back to back alternate between the backend's two lanes and overlap on
        the GPU; every rank must submit the same searches in the same order (the all-gathers
        pair up by order).  The C ABI treats a NULL stream as "use the library's stream and
        synchronise", and torch's default stream IS the NULL stream: the lanes' own streams are
        also what keeps the host from blocking on every call."""
        k = int(k)
        lanes = getattr(self.backend, "next_lane", None)
        if lanes is None:  # a backend without device streams (the CPU test backend)
            D, I, status, err = self._search_on_current_stream(q, k)
            return PendingSearch(D, I, None, status, err)
        index, side = lanes()
        side.wait_stream(torch.cuda.current_stream(q.device))
        q.record_stream(side)
        with torch.cuda.stream(side):
            D, I, status, err = self._search_on_current_stream(q, k, index)
            done = side.record_event()
            if self.step_events is not None:  # (bench.py: a timing event behind every search, for per-step durations)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(side)
                self.step_events.append(ev)
        return PendingSearch(D, I, done, status, err)

    def search_dev(self, q, k, check=True):
        """q: [nq, d] float32 tensor, identical on every rank.  Returns (D, I) tensors holding the
        global result on every rank, ordered after the search on the caller's current stream.  Over several ranks the
        call waits for the search's status row (``PendingSearch.result``) and raises ``ShardSearchError`` on every rank
        if any rank failed.  ``check=False`` leaves the status row unread and the host free to enqueue the next search:
        the pending search is kept in ``self.unchecked`` and a caller that pipelines K searches calls ``check_pending()``
        once behind them (what bench.py does with ``submit``)."""
        if (check and self.world == 1 and not self.force_collective and self.row_offset == 0 and q.shape[0] > 128
                and hasattr(self.backend, "next_lane")):
            # One rank, a batch of several query tiles, a caller that waits for the result anyway: the synchronous entry.  It
            # can read a statistical seed's verification flag before it returns, so it may take that seed and with it the
            # 256 x 256 tile -- the lanes are asynchronous and never do (10 M rows x 1024 queries: 0.886 against 0.838 of the
            # fp32 MFMA peak).
            return self.backend.search(q, int(k))
        pnd = self.submit(q, k)
        if not check:
            self.unchecked.append(pnd)
        return pnd.result(check=check)

    def check_pending(self):
        """Reads the status rows of every ``search_dev(..., check=False)`` since the last call; raises
        ``ShardSearchError`` (on every rank) for the first search that failed anywhere."""
        pend, self.unchecked = self.unchecked, []
        for pnd in pend:
            pnd.check()

    def search(self, x: np.ndarray, k):
        _faiss._check_matrix(x, self.d)
        dev = getattr(self.backend, "device", torch.device("cpu"))
        D, I = self.search_dev(torch.from_numpy(x).to(dev), k)
        if D.is_cuda:
            torch.cuda.current_stream(D.device).synchronize()
        return D.cpu().numpy(), I.cpu().numpy()


DEFAULT_BACKEND = HipShardBackend  # (what a sharded index is built on when none is passed; the CPU tests put their double here)


class QueryShardedFlatIndex:
    """The other way to spread an all-vs-all over the GPUs ("alternative"): the database is
    REPLICATED -- CATH (59 MB) and Pfam (819 MB) fit any one GPU many times over -- and the queries are split:
    rank r answers queries [lo_r, hi_r) = ``shard_bounds(nq, world, r)`` against all rows.  No collective on the
    data path at all; every query's result is computed by one rank with the single-GPU kernels, so it is the
    single-GPU result bit for bit.  ``gather=True`` concatenates the slices on every rank with one all-gather of
    D and one of I (padded to ceil(nq/world) rows per rank); the default leaves each rank with its own slice,
    which is what a caller that writes per-rank output files wants (2.4 GB of results at Pfam size, k=1000)."""

    def __init__(self, d, metric=_faiss.METRIC_L2, rank=None, world=None, group=None, backend=None):
        self.d, self.metric_type, self.group = d, metric, group
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world = rank, world
        self.backend = backend if backend is not None else DEFAULT_BACKEND(d, metric)
        self.unchecked = []  # (status, local error) of gathered searches whose outcome nobody has read yet

    @property
    def ntotal(self):
        return self.backend.ntotal

    @property
    def replica(self):
        """this rank's copy of the database as a plain ``faiss.IndexFlat`` (``write_index``, ``reconstruct``)"""
        return self.backend.index

    def reserve(self, n):
        self.backend.reserve(n)

    def train(self, x):
        """no-op, as for IndexFlat"""

    def add(self, x):
        """Adds rows to THIS rank's replica: every rank adds the same rows in the same order."""
        self.backend.add(x)

    def reconstruct_into(self, x):
        self.backend.index.reconstruct_into(x)

    def add_dev(self, x):
        self.backend.add_dev(x)

    def query_bounds(self, nq):
        return shard_bounds(nq, self.world, self.rank)

    def search_dev(self, q, k, gather=False, check=True):
        """q: [nq, d] float32 tensor, identical on every rank.  Returns (D, I) of this rank's query slice
        (``query_bounds(nq)``), or of all nq queries on every rank with ``gather=True``.  A gathered search carries every
        rank's outcome as one extra row of the id gather (no collective and no host round trip of its own in front of the
        gathers); ``check=True`` reads those rows behind the gathers and raises ``ShardSearchError`` on every rank if any
        rank's slice failed, ``check=False`` leaves them in ``self.unchecked`` for ``check_pending()``."""
        k = int(k)
        nq = q.shape[0]
        lo, hi = self.query_bounds(nq)
        local_error = None
        D = torch.empty((0, k), dtype=torch.float32, device=q.device)
        I = torch.empty((0, k), dtype=torch.int64, device=q.device)
        if hi > lo:
            # (this rank's slice is a piece of the caller's batch: the L2 formula FAISS would pick for all nq queries)
            set_batch = getattr(getattr(self.backend, "index", None), "set_batch", None)
            if set_batch:
                set_batch(nq)
            try:
                D, I = self.backend.search(q[lo:hi].contiguous(), k)
            except Exception as e:  # noqa: BLE001
                if not gather or self.world == 1:
                    raise
                local_error = e  # (the peers are about to enter the gathers: this rank enters them too, with its flag set)
                D = torch.zeros((hi - lo, k), dtype=torch.float32, device=q.device)
                I = torch.full((hi - lo, k), -1, dtype=torch.int64, device=q.device)
            finally:
                if set_batch:
                    set_batch(0)
        if not gather or self.world == 1:
            return D, I
        Dg, Ig, status = self._gather_with_status(D, I, nq, k, q.device, local_error)
        self.unchecked.append((status, local_error))
        if check:
            self.check_pending()
        return Dg, Ig

    def _gather_with_status(self, D, I, n, k, dev, local_error):
        """Two all-gathers, rank-major: D padded to ceil(n / world) rows per rank, I to one row more -- the STATUS row, whose
        first slot is 1 on a rank whose slice failed (every rank then knows, from the gather it was going to do anyway).
        -> (D [n, k], I [n, k], status [world])."""
        per = (n + self.world - 1) // self.world
        m = D.shape[0]
        Dp = torch.zeros((per, k), dtype=torch.float32, device=dev)
        Ip = torch.full((per + 1, k), -1, dtype=torch.int64, device=dev)
        Dp[:m] = D
        Ip[:m] = I
        Ip[per, 0] = 0 if local_error is None else 1
        Dg = torch.empty((self.world * per, k), dtype=torch.float32, device=dev)
        Ig = torch.empty((self.world * (per + 1), k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(Dg, Dp, group=self.group)
        dist.all_gather_into_tensor(Ig, Ip, group=self.group)
        Ig = Ig.view(self.world, per + 1, k)
        status = Ig[:, per, 0]
        # (contiguous slices of ceil(n / world) queries: rank-major concatenation is query order, once the status rows are out)
        return Dg[:n], Ig[:, :per].reshape(self.world * per, k)[:n], status

    def check_pending(self):
        """Reads the status rows of the gathered searches since the last call (one small device-to-host copy each); a
        failure anywhere raises ``ShardSearchError`` on EVERY rank."""
        pend, self.unchecked = self.unchecked, []
        for status, local_error in pend:
            failed = [r for r, f in enumerate(status.cpu().tolist()) if f]
            if failed:
                raise ShardSearchError(failed, local_error) from local_error

    def normalize_rows(self):
        """L2-normalises the replica's rows in HBM (every rank: the same rows, the same bits)"""
        self.backend.normalize_rows()

    def search_self(self, k, gather=True):
        """Every row of the (replicated) database against all rows: rank r answers rows ``query_bounds(ntotal)`` --
        its slice of the queries is already in its HBM, nothing is uploaded -- and with ``gather=True`` every rank
        returns all ntotal rows of (D, I) as host arrays (two all-gathers), else its own slice."""
        k = int(k)
        n = self.ntotal
        lo, hi = self.query_bounds(n)
        set_batch = getattr(getattr(self.backend, "index", None), "set_batch", None)
        if set_batch:
            set_batch(n)  # (the slice is a piece of an n-query batch: FAISS's choice of the L2 formula)
        local_error = None
        D, I = np.empty((0, k), np.float32), np.empty((0, k), np.int64)
        try:
            if hi > lo:
                D, I = self.backend.search_self(k, lo, hi - lo)
        except Exception as e:  # noqa: BLE001
            if not gather or self.world == 1:
                raise
            local_error = e
        finally:
            if set_batch:
                set_batch(0)
        if not gather or self.world == 1:
            return D, I
        # gloo gathers host tensors, RCCL device tensors
        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = getattr(self.backend, "device", torch.device("cpu")) if on_gpu else torch.device("cpu")
        if local_error is not None:
            D, I = np.zeros((hi - lo, k), np.float32), np.full((hi - lo, k), -1, np.int64)
        Dg, Ig, status = self._gather_with_status(torch.from_numpy(D).to(dev), torch.from_numpy(I).to(dev), n, k, dev, local_error)
        if on_gpu:
            torch.cuda.current_stream(dev).synchronize()
        self.unchecked.append((status, local_error))
        self.check_pending()
        return Dg.cpu().numpy(), Ig.cpu().numpy()

    def search(self, x: np.ndarray, k, gather=True):
        _faiss._check_matrix(x, self.d)
        dev = getattr(self.backend, "device", torch.device("cpu"))
        D, I = self.search_dev(torch.from_numpy(x).to(dev), k, gather=gather)
        if D.is_cuda:
            torch.cuda.current_stream(D.device).synchronize()
        return D.cpu().numpy(), I.cpu().numpy()
