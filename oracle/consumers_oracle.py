"""CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the consumers of (hits, scores): pure-Python
loops restating the reference, small inputs only.

  remove_self_hit         pfam/proteins.py:85-122
  evaluate                seqvec_search/main.py:64-82
  compute_auc1            pfam/proteins_shared.py:139-157
  compute_is_correct      cath/cath.py:76-84
  compute_tps_comulative  seqvec_search/tp_cumulative.py:15-34
  write_prefilter_db      seqvec_search/mmseqs/_write_prefilter_db.py:52-97
"""
from collections import Counter

import numpy as np


def remove_self_hit(hits, scores, self_ids):
    hits, scores = hits.copy(), scores.copy()
    bogus = 0
    for r in np.argwhere(hits[:, 0] != self_ids)[:, 0]:
        sid = self_ids[r]
        row = list(hits[r])
        if sid in row:
            index = row.index(sid)
        else:
            index = len(row) - 1
            bogus += 1
        hits[r, 0], hits[r, 1:index + 1] = hits[r, index].copy(), hits[r, 0:index].copy()
        scores[r, 0], scores[r, 1:index + 1] = scores[r, index].copy(), scores[r, 0:index].copy()
    return hits[:, 1:], scores[:, 1:], bogus


def evaluate(ids_to_family, train_ids, test_ids, results):
    family_sizes = dict(Counter(ids_to_family[i] for i in train_ids))
    auc1s, tps = [], []
    for key, row in enumerate(results):
        name = test_ids[key]
        matches = [train_ids[i] for i in row]
        correct = ids_to_family[name]
        tp = sum(ids_to_family[i] == correct for i in matches)
        auc1 = 0
        for i in matches:
            if ids_to_family[i] == correct:
                auc1 += 1
            else:
                break
        auc1s.append(auc1 / family_sizes[correct])
        tps.append(tp / family_sizes[correct])
    return auc1s, tps


def compute_auc1(hits, homologous_proteins, queries, target_ids):
    out = []
    for index, row in enumerate(hits):
        all_correct = homologous_proteins[queries[index]]
        auc1 = 0
        for hit in row:
            if target_ids[hit] in all_correct:
                auc1 += 1
            else:
                break
        out.append(auc1 / max(len(all_correct), 1))
    return np.asarray(out)


def compute_is_correct(results, mapping_array):
    return np.asarray([(mapping_array[q] == mapping_array[res]).T for q, res in zip(range(len(results)), results)])


def compute_tps_comulative(ids_to_family, train_ids, test_ids, results):
    family_sizes = dict(Counter(ids_to_family[i] for i in train_ids))
    is_correct, tp_counts = [], []
    for key, row in enumerate(results):
        is_correct.append(np.asarray([ids_to_family[train_ids[h]] for h in row]) == ids_to_family[test_ids[key]])
        tp_counts.append(family_sizes[ids_to_family[test_ids[key]]])
    is_correct = np.asarray(is_correct)
    tp_counts = np.asarray(tp_counts)
    expanded = tp_counts.repeat(is_correct.shape[1]).reshape(is_correct.shape)
    return (is_correct.cumsum(axis=1) / expanded).mean(axis=0)


def write_prefilter_db(hits, queries, scores, test_map, train_map, clip=True, numpy2=None):
    """Returns (data bytes, index bytes) as the reference writes them.  Clipped branch: double
    arithmetic under numpy 1.x (the reference pins 1.22.2: value-based casting of the 10**30
    bounds), float32 under numpy >= 2 (NEP 50); numpy2=None follows the running numpy."""
    data, index = bytearray(), bytearray()
    offset = 0
    if numpy2 is None:
        numpy2 = int(np.__version__.split(".")[0]) >= 2
    if clip and numpy2:
        scores_int = np.clip(scores.astype(np.float32), np.float32(-1e30), np.float32(1e30)) * np.float32(100)
    elif clip:
        scores_int = np.clip(scores.astype(np.float64), -(10 ** 30), 10 ** 30) * 100
    else:
        scores_int = scores * np.float32(100)
    for query, hit_entry, score_entry in zip(queries, hits, scores_int):
        length = 0
        for hit, score in zip(hit_entry, score_entry):
            if hit == -1:
                continue
            line = f"{train_map[hit]}\t{int(score)}\t0\n".encode()
            length += len(line)
            data += line
        data += b"\0"
        length += 1
        index += f"{test_map[query]}\t{offset}\t{length}\n".encode()
        offset += length
    return bytes(data), bytes(index)
