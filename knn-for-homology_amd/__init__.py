"""knn355 -- MI355X-native flat / HNSW kNN search behind the reference's own entry points.

Layout (mirrors the reference modules that sit on the kNN hot path):

  faiss                      the slice of the ``faiss`` module the reference calls
  cath.search                cath/search.py           (search, search_and_save)
  pfam.proteins_search       pfam/proteins_search.py  (main: flat | hnsw | lsh)
  pfam.search                pfam/search.py           (load_embeddings, search_flat, search_index)
  pfam.slices.slices_search  pfam/slices/slices_search.py
  seqvec_search.main         seqvec_search/main.py    (faiss_search, evaluate_faiss, evaluate)
  seqvec_search.create_index seqvec_search/create_index.py (main)
  sharded                    row-sharded multi-GPU index (one process per GPU, RCCL all-gather)

All arithmetic runs in ``libknn355.so`` (HIP, gfx950) through ctypes; there is no
CPU fallback -- importing works anywhere, computing without the library or
without a GPU raises.
"""
__version__ = "0.1.0"
