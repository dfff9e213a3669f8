"""GPU: process / thread behaviour the reference relies on (SURVEY 7.3 H6, 8(b) threading).
cath/compare_seqvec_layer.py:58-64 calls cath.search.search from two forked worker processes;
nothing may touch HIP before the fork, and every worker must get its own context."""
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_forked_workers_like_compare_seqvec_layer():
    script = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(ROOT)!r})
        import numpy as np
        from concurrent.futures import ProcessPoolExecutor
        import multiprocessing as mp
        # importing the package (and loading the library) must not create any HIP state
        from knn_for_homology_amd import _lib
        from knn_for_homology_amd.cath.search import search
        _lib.lib()
        from oracle import knn_oracle as ko

        def work(seed):
            rng = np.random.default_rng(seed)
            x = rng.standard_normal((700, 256), dtype=np.float32)
            hits, scores = search(x, hits=10)
            return seed, hits, scores

        if __name__ == "__main__":
            mp.set_start_method("fork")
            with ProcessPoolExecutor(max_workers=2) as pool:
                results = list(pool.map(work, [1, 2, 3, 4]))
            orc = ko.oracle()
            for seed, hits, scores in results:
                rng = np.random.default_rng(seed)
                x = rng.standard_normal((700, 256), dtype=np.float32)
                orc.normalize_l2(x)
                D, I = orc.flat_search(x, x, 11, 0)
                assert np.array_equal(hits, I[:, 1:]) and np.array_equal(scores, D[:, 1:])
            print("FORK_OK")
    """)
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert "FORK_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_two_threads_two_handles(gpu_faiss, oracle):
    """Calls on different handles may run concurrently (ctypes releases the GIL)."""
    import threading
    rng = np.random.default_rng(9)
    data = [(rng.standard_normal((6000, 128), dtype=np.float32), rng.standard_normal((90, 128), dtype=np.float32), m)
            for m in (0, 1, 0, 1)]
    out = [None] * len(data)

    def run(i):
        xb, xq, m = data[i]
        idx = gpu_faiss.IndexFlat(128, m)
        idx.add(xb)
        for _ in range(5):
            out[i] = idx.search(xq, 20)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(data))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for (xb, xq, m), (D, I) in zip(data, out):
        Do, Io = oracle.flat_search(xb, xq, 20, m)
        assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))


def test_same_handle_from_threads_is_serialised(gpu_faiss, oracle):
    import threading
    rng = np.random.default_rng(10)
    xb = rng.standard_normal((5000, 64), dtype=np.float32)
    idx = gpu_faiss.IndexFlat(64, 1)
    idx.add(xb)
    qs = [rng.standard_normal((40, 64), dtype=np.float32) for _ in range(6)]
    out = [None] * 6

    def run(i):
        out[i] = idx.search(qs[i], 15)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for q, (D, I) in zip(qs, out):
        Do, Io = oracle.flat_search(xb, q, 15, 1)
        assert np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))
