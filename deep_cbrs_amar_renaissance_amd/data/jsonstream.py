"""Streaming readers for the two embedding file formats of the reference.

The reference loads them whole (`json.load` / `pd.read_json`, `/root/reference/src/data/loaders.py:85-105`); the files are
190 MB - 5.5 GB of JSON text (`embeddings/*.dvc`), i.e. tens of GB of Python float objects.  Here the text is read in
chunks and each row goes straight into a float32 array:

* KGE (OpenKE export):   ``{"ent_embeddings": [[f, f, ...], [f, ...], ...], <other keys ignored>}``
* BERT:                   ``[{"ID_OpenKE": 12, "embedding" | "profile_embedding": [f, ...], <other keys ignored>}, ...]``

Host-side plumbing of the boundary; no arithmetic.
"""
import json

import numpy as np

CHUNK = 16 << 20


def _chunks(fp, chunk):
    while True:
        block = fp.read(chunk)
        if not block:
            return
        yield block


def stream_ent_embeddings(filepath, key='ent_embeddings', chunk=CHUNK):
    """float32 [n_rows, D] of the array of arrays stored under `key`."""
    rows, buf, state = [], '', 'seek'          # seek: before the key; outer: inside [[...], ...]; done
    needle = '"' + key + '"'
    with open(filepath) as fp:
        for block in _chunks(fp, chunk):
            buf += block
            if state == 'seek':
                at = buf.find(needle)
                if at < 0:
                    buf = buf[-len(needle):]                      # the key may straddle two chunks
                    continue
                rest = buf[at + len(needle):]
                opening = rest.find('[')
                if opening < 0:
                    buf = buf[at:]
                    continue
                buf, state = rest[opening + 1:], 'outer'
            if state == 'outer':
                pos = 0
                while True:
                    start = buf.find('[', pos)
                    closing = buf.find(']', pos)
                    if closing >= 0 and (start < 0 or closing < start):
                        state = 'done'                             # the outer array closes before another row opens
                        break
                    if start < 0:
                        break
                    stop = buf.find(']', start)
                    if stop < 0:
                        break                                      # the row continues in the next chunk
                    rows.append(np.array(buf[start + 1:stop].split(','), dtype=np.float32))
                    pos = stop + 1
                buf = buf[pos:]
            if state == 'done':
                break
    if state == 'seek':
        raise KeyError("no '{}' array in {}".format(key, filepath))
    if state != 'done':
        raise ValueError("truncated '{}' array in {}".format(key, filepath))
    if not rows:
        return np.zeros((0, 0), dtype=np.float32)
    width = len(rows[0])
    if any(len(r) != width for r in rows):
        raise ValueError("ragged rows under '{}' in {}".format(key, filepath))
    return np.stack(rows)


def stream_bert_records(filepath, column, id_column='ID_OpenKE', chunk=CHUNK):
    """(ids int64 [n], float32 [n, D]) of a JSON array of records, one record decoded at a time."""
    decoder = json.JSONDecoder()
    ids, rows, buf, started, finished = [], [], '', False, False
    with open(filepath) as fp:
        for block in _chunks(fp, chunk):
            buf += block
            pos = 0
            while True:
                while pos < len(buf) and buf[pos] in ' \t\r\n,':
                    pos += 1
                if pos >= len(buf):
                    break
                if not started:
                    if buf[pos] != '[':
                        raise ValueError("{}: a JSON array of records expected".format(filepath))
                    started, pos = True, pos + 1
                    continue
                if buf[pos] == ']':
                    finished = True
                    break
                try:
                    record, end = decoder.raw_decode(buf, pos)
                except json.JSONDecodeError:
                    break                                          # the record continues in the next chunk
                ids.append(int(record[id_column]))
                rows.append(np.asarray(record[column], dtype=np.float32))
                pos = end
            buf = buf[pos:]
            if finished:
                break
    if not finished:
        raise ValueError("truncated record array in {}".format(filepath))
    table = np.stack(rows) if rows else np.zeros((0, 0), dtype=np.float32)
    return np.asarray(ids, dtype=np.int64), table
