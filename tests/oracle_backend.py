"""Test-only backend: lets the host bookkeeping of IVInterpolator (packing, join, gather,
assembly) run on CPU by answering the two device calls from the oracle.  Lives under tests/
on purpose: the product package has no CPU route."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ivs_oracle as O  # noqa: E402


class OracleBackend:
    def interp1d_batch(self, xk, yk, knot_off, q_off, total_q, code):
        return O.interp1d_batch(xk, yk, knot_off, None, q_off, code)

    def ffill_index_batch(self, src_pos, src_off, valid, q_off, total_q):
        n_cols = valid.shape[0]
        idx = np.full((n_cols, total_q), -1, np.int32)
        for s in range(len(src_off) - 1):
            a, b = int(src_off[s]), int(src_off[s + 1])
            m = int(q_off[s + 1] - q_off[s])
            last = np.searchsorted(src_pos[a:b], np.arange(m), side="right") - 1      # last source row <= i
            for c in range(n_cols):
                v = valid[c, a:b].astype(bool)
                prev_valid = np.where(v, np.arange(b - a), -1)
                prev_valid = np.maximum.accumulate(prev_valid)                          # last valid row <= j
                r = np.where(last >= 0, prev_valid[np.clip(last, 0, None)], -1)
                idx[c, q_off[s]:q_off[s + 1]] = np.where(r >= 0, r + a, -1)
        return idx
