"""Test helper: replays a packed tile program in float64 log
space, checking the schedule's invariants on the way.  Used to validate the host
packer on CPU, independently of the HIP kernels."""
import numpy as np

from nfst_amd import _lib


def replay(lat, b, direction, theta, extra=None):
    """Returns log values (alpha for 'fwd', beta for 'bwd') per row of lattice b."""
    m = lat.meta_host[b]
    n_rows = int(m[_lib.META_N_ROWS])
    V = lat.vocab
    if direction == "fwd":
        s = lat.fwd_stream.cpu().numpy().view(np.uint32)
        perm = lat.fwd_perm.cpu().numpy()
        off, tiles, U, slot0, start = (int(m[_lib.META_FWD_OFF]), int(m[_lib.META_FWD_TILES]), int(m[_lib.META_FWD_U]) & 0xFF,
                                       int(m[_lib.META_FWD_SLOT_OFF]), 0)
    else:
        s = lat.bwd_stream.cpu().numpy().view(np.uint32)
        perm = lat.bwd_perm.cpu().numpy()
        off, tiles, U, slot0, start = (int(m[_lib.META_BWD_OFF]), int(m[_lib.META_BWD_TILES]), int(m[_lib.META_BWD_U]) & 0xFF,
                                       int(m[_lib.META_BWD_SLOT_OFF]), int(m[_lib.META_SINK]))
    wide = (int(m[_lib.META_FWD_U if direction == "fwd" else _lib.META_BWD_U]) >> 8) & 1
    compact = U == 8  # format code 8: 16 bytes per lane = control word + four 24-bit records
    if compact:
        U = 4
    assert U in (1, 2, 4) and off % 64 == 0
    ST = 256 if compact else 64 * (1 + U)
    src = lat.arc_src.cpu().numpy(); dst = lat.arc_dst.cpu().numpy(); lab = lat.arc_label.cpu().numpy()
    n_lds = max(n_rows, int(lat.max_rows))  # + scratch rows of partial groups
    val = np.full(n_lds, -np.inf)
    done = np.zeros(n_lds, bool)
    val[start] = 0.0
    done[start] = True
    seen = []
    owner = {}  # scratch row -> the state whose arcs its partial group sums
    for T in range(tiles):
        base = off + T * ST
        if compact:
            t4 = s[base: base + 256].reshape(64, 4).astype(np.uint64)
            ctl = t4[:, 0].astype(np.uint32)
            r24 = np.stack([t4[:, 1] & 0xFFFFFF, ((t4[:, 1] >> 24) | (t4[:, 2] << 8)) & 0xFFFFFF,
                            ((t4[:, 2] >> 16) | (t4[:, 3] << 16)) & 0xFFFFFF, (t4[:, 3] >> 8) & 0xFFFFFF], axis=1)
            rec = (((r24 & 0x1FFF) << 3) | ((r24 >> 13) << 16)).astype(np.uint32)  # as 32-bit records
        else:
            ctl = s[base: base + 64]
            rec = s[base + 64: base + ST].reshape(64, U)
        pm = perm[slot0 + T * 64 * U: slot0 + (T + 1) * 64 * U].reshape(64, U)
        gmax = int(ctl[0] >> 23) & 7
        assert np.all(((ctl >> 23) & 7) == gmax)
        writes = {}
        lane = 0
        while lane < 64:
            c = int(ctl[lane])
            g = (c >> 20) & 7
            size = 1 << g
            assert g <= gmax and lane % size == 0 and g <= (6 if wide else 3)
            leader = (c >> 31) & 1
            assert (c & 7) == 0 and ((c >> 16) & 15) == 0
            sid = (c & 0xFFFF) >> 3
            terms = []
            carries = 0
            for r in range(size):
                cr = int(ctl[lane + r])
                assert ((cr >> 20) & 7) == g and ((cr & 0xFFFF) >> 3) == sid
                assert ((cr >> 31) & 1) == (leader if r == 0 else 0)
                for j in range(U):
                    ca = int(pm[lane + r, j])
                    assert int(rec[lane + r, j]) & 7 == 0
                    other, l = (int(rec[lane + r, j]) & 0xFFFF) >> 3, int(rec[lane + r, j]) >> 16
                    if ca < 0:
                        if l == V + 1 and other == sid:  # carry record of a continuation piece: weight one
                            assert r == 0 and j == 0 and (c >> 30) & 1
                            assert done[sid], "continuation piece before the state's first piece"
                            terms.append(val[sid])
                            carries += 1
                        elif l == V + 1:  # combine piece: the sum a partial group left in a scratch row
                            assert other >= n_rows and done[other] and sid < n_rows and owner.pop(other) == sid
                            terms.append(val[other])
                        else:
                            assert l == V  # the null label
                        continue
                    assert leader or r > 0 or True
                    assert lab[ca] == l
                    mine = int(dst[ca] if direction == "fwd" else src[ca])
                    assert int(src[ca] if direction == "fwd" else dst[ca]) == other
                    if sid < n_rows:
                        assert mine == sid
                    else:  # partial group: all its arcs belong to one state, remembered for the combine piece
                        assert owner.setdefault(sid, mine) == mine
                    assert done[other], "operand state not finished by an earlier tile"
                    x = float(theta[l]) + (float(extra[ca]) if extra is not None else 0.0)
                    terms.append(x + val[other])
                    seen.append(ca)
            if leader:
                t = np.array(terms)
                v = -np.inf if len(t) == 0 or np.all(np.isneginf(t)) else t.max() + np.log(np.exp(t - t.max()).sum())
                assert carries == ((c >> 30) & 1)
                assert sid not in writes
                writes[sid] = v
            else:
                assert not terms  # idle lanes carry no arcs
            lane += size
        for sid, v in writes.items():
            val[sid] = v
            done[sid] = True
    n_dp = int(m[_lib.META_N_DP])
    assert len(seen) == n_dp and len(set(seen)) == n_dp  # every DP arc exactly once
    return val[:n_rows]
