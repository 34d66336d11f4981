"""Test helper: replays a packed sweep stream in float64
log space, checking the schedule's invariants on the way.  Used to validate the
host packer on CPU, independently of the HIP kernels."""
import numpy as np

from nfst_amd import _lib


def replay(lat, b, direction, theta, extra=None):
    """Returns log values (alpha for 'fwd', beta for 'bwd') per row of lattice b."""
    m = lat.meta_host[b]
    n_rows = int(m[_lib.META_N_ROWS])
    if direction == "fwd":
        s = lat.fwd_stream.cpu().numpy().view(np.uint32)
        perm = lat.fwd_perm.cpu().numpy()
        off, steps, start = int(m[_lib.META_FWD_OFF]), int(m[_lib.META_FWD_STEPS]), 0
        words = int(m[_lib.META_FWD_WORDS])
    else:
        s = lat.bwd_stream.cpu().numpy().view(np.uint32)
        perm = lat.bwd_perm.cpu().numpy()
        off, steps, start = int(m[_lib.META_BWD_OFF]), int(m[_lib.META_BWD_STEPS]), int(m[_lib.META_SINK])
        words = int(m[_lib.META_BWD_WORDS])
    src = lat.arc_src.cpu().numpy(); dst = lat.arc_dst.cpu().numpy(); lab = lat.arc_label.cpu().numpy()
    dp_off = int(m[_lib.META_DP_OFF])
    val = np.full(n_rows, -np.inf)
    done = np.zeros(n_rows, bool)
    val[start] = 0.0
    done[start] = True
    seen_arcs = []
    base = off
    arc_base = 0
    for _ in range(steps):
        h0, na = int(s[off]), int(s[off + 1])
        ns, kl, accum = h0 & 0xFFFF, (h0 >> 16) & 0xF, (h0 >> 20) & 1
        assert 0 <= kl <= 6 and ns >= 1
        st = s[off + 2: off + 2 + ns + 1]
        rec = s[off + 3 + ns: off + 3 + ns + na]
        assert int(st[ns] >> 16) == na and int(st[ns] & 0xFFFF) == 0xFFFF
        new = {}
        for i in range(ns):
            sid, a0, a1 = int(st[i] & 0xFFFF), int(st[i] >> 16), int(st[i + 1] >> 16)
            assert a0 <= a1 <= na and sid < n_rows
            terms = []
            for a in range(a0, a1):
                other, l = int(rec[a] & 0xFFFF), int(rec[a] >> 16)
                ca = int(perm[dp_off + arc_base + a])
                # the record agrees with the canonical arc it stands for
                assert lab[ca] == l
                if direction == "fwd":
                    assert src[ca] == other and dst[ca] == sid
                else:
                    assert dst[ca] == other and src[ca] == sid
                assert done[other], "dependency not finished before use"
                x = float(theta[l]) + (float(extra[ca]) if extra is not None else 0.0)
                terms.append(x + val[other])
                seen_arcs.append(ca)
            t = np.array(terms)
            v = -np.inf if len(t) == 0 or np.all(np.isneginf(t)) else t.max() + np.log(np.exp(t - t.max()).sum())
            if accum:
                assert sid in new or done[sid] or True
                v = np.logaddexp(v, val[sid])
            new[sid] = v
            val[sid] = v  # accumulate steps read their own earlier partial
        for sid in new:
            done[sid] = True
        off += 2 + ns + 1 + na
        arc_base += na
    assert off - base == words
    n_dp = int(m[_lib.META_N_DP])
    assert arc_base == n_dp
    assert sorted(seen_arcs) == sorted(int(a) for a in perm[dp_off: dp_off + n_dp])
    assert len(set(seen_arcs)) == n_dp  # every DP arc exactly once
    return val
