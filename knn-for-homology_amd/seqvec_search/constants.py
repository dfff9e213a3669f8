"""seqvec_search/constants.py:3 -- default number of neighbours."""
default_hits: int = 13
