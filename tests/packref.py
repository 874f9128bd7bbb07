"""numpy statement of the compact match-record format of SURVEY 8(e) / include/orbmatch.h (orbm_pack_records): the checker
of the device packer in the GPU tests and the stand-in for it in the CPU (gloo) tests.  Test infrastructure only."""
import numpy as np

SUMMARY_DTYPE = np.dtype([("pair_id", "<u4"), ("n", "<u2"), ("pad", "<u2")])
PAD_ID = 0xFFFFFFFF


def pack(pair_id, counts, match, threshold):
    """-> (summaries, lists, (accepted pairs, list entries))"""
    p = len(counts)
    pid = np.arange(p, dtype=np.uint32) if pair_id is None else np.asarray(pair_id, np.uint32)
    n = np.where(pid == PAD_ID, 0, np.asarray(counts)).astype(np.int64)
    summ = np.zeros(p, SUMMARY_DTYPE)
    summ["pair_id"] = pid
    summ["n"] = np.clip(n, 0, 65535)
    lists, accepted = [], 0
    for k in range(p):
        if n[k] >= threshold and n[k] > 0:
            rows = np.nonzero(match[k] >= 0)[0][: n[k]]
            lists.append(rows.astype(np.uint32) | (match[k][rows].astype(np.uint32) << 16))
            accepted += 1
    lists = np.concatenate(lists) if lists else np.zeros(0, np.uint32)
    return summ, lists, (accepted, len(lists))
