"""MMseqs2 bridge, writer half only (seqvec_search/mmseqs/_write_prefilter_db.py).  Running
MMseqs2 itself (align/search, result parsing) needs the external binary and stays with the
reference."""
from ._write_prefilter_db import make_id_map, write_prefilter_db  # noqa: F401
