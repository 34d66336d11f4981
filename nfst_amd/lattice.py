"""LatticeBatch: a batch of nFST lattices packed once, shared by all K samples.

Host-side mirror of ``FSAGRUScorer.set_masks`` / ``set_k``
(/root/reference/src/modules/scorers.py:877-918).  The reference keeps the dense
``emission``/``transition`` tables ``[B, S+1, V]`` and materialises K copies of
them; here the tables are scanned once into canonical CSR arrays plus two
level-scheduled arc streams (include/nfst_hip.h) that the
HIP kernels consume.  torch is used for device memory only.
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


_warned_d2h = False


def _host(a, dtype) -> np.ndarray:
    """The packer runs on the host.  Tables that already live on the GPU (the reference's trainer moves
    the collated batch there before ``set_masks``) are copied back first -- 5 MB per 2k-state lattice
    at V = 256, far more than the step itself costs; said once, not done silently: pack in the
    DataLoader workers instead (``io.load_packed`` / ``io.collate_packed``, section 8f-1)."""
    global _warned_d2h
    if isinstance(a, torch.Tensor):
        if a.is_cuda and not _warned_d2h:
            _warned_d2h = True
            warnings.warn(f"nfst_amd: dense lattice tables on {a.device} are copied to the host for packing "
                          f"({a.numel() * a.element_size() / 1e6:.1f} MB for this tensor); keep the tables on the host, or pack "
                          "per example in the data loader (nfst_amd.io.load_packed / collate_packed)", stacklevel=3)
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


def _view(ptr, n, ctype, dtype) -> np.ndarray:
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True)


class ChunkProgram:
    """The chunked programs of a batch of deep, narrow lattices (``nfst_chunks``, include/nfst_hip.h): per lattice and
    direction the states in topological order, cut into chunks that are swept at the same time.  Built on the host by
    ``LatticeBatch.build_chunks`` (or by ``LatticeBatch.to`` when the cost model says the batch is one for this flavour);
    the device copy owns the scratch every launch on the batch uses (one launch at a time per batch)."""
    _FIELDS = ("meta", "tab", "stream", "pos", "label")
    _HEADER = ("n_lattices", "threads", "lds_bytes", "n_tab", "n_stream", "n_pos", "t_units", "total_rows", "total_arcs")

    def __init__(self, header: dict, tensors: dict):
        self._h, self._t = dict(header), dict(tensors)
        self.meta_host = self._t["meta"].detach().cpu().numpy().reshape(-1, 2, _lib.CHK_META_WORDS).copy()
        self.ws = None
        self._struct = None

    @classmethod
    def build(cls, lat: "LatticeBatch", threads: int = 0, lds_bytes: int = 0, force: bool = False,
              max_chunks: int = 0, n_threads: int = 0) -> "Optional[ChunkProgram]":
        if lat.device.type != "cpu":
            raise ValueError("chunked programs are cut from the host copy of a batch")
        opts = _lib.ChunkOpts(int(threads), int(lds_bytes), 1 if force else 0, int(max_chunks), int(n_threads), 0)
        handle = C.c_void_p()
        check(lib.nfst_pack_chunks(C.byref(lat.c_struct()), C.byref(opts), C.byref(handle)), "nfst_pack_chunks")
        if not handle:
            return None  # not a batch for this flavour
        try:
            v = _lib.Chunks()
            check(lib.nfst_chunks_view(handle, C.byref(v)), "nfst_chunks_view")
            arrs = {"meta": _view(v.meta, v.n_lattices * 2 * _lib.CHK_META_WORDS, C.c_int32, np.int32),
                    "tab": _view(v.tab, v.n_tab * 4, C.c_int32, np.int32),
                    "stream": _view(v.stream, v.n_stream, C.c_int32, np.int32),
                    "pos": _view(v.pos, v.n_pos, C.c_int32, np.int32),
                    "label": _view(v.label, v.n_stream, C.c_int16, np.int16)}
            header = {k: int(getattr(v, k)) for k in cls._HEADER}
        finally:
            lib.nfst_chunks_free(handle)
        return cls(header, {k: torch.from_numpy(a) for k, a in arrs.items()})

    def to(self, device, non_blocking: bool = False) -> "ChunkProgram":
        device = torch.device(device)
        out = ChunkProgram.__new__(ChunkProgram)
        out._h, out.meta_host, out._struct, out.ws = dict(self._h), self.meta_host, None, None
        out._t = {k: v.to(device, non_blocking=non_blocking) for k, v in self._t.items()}
        if device.type == "cuda":
            # (zeroed once: the flags at its end are compared with the launch counter, which starts at one)
            out.ws = torch.zeros(int(lib.nfst_chunks_ws_bytes(C.byref(out.c_struct()))), dtype=torch.uint8, device=device)
            out._struct = None
        return out

    def pin_memory(self) -> "ChunkProgram":
        out = ChunkProgram.__new__(ChunkProgram)
        out._h, out.meta_host, out._struct, out.ws = dict(self._h), self.meta_host, None, None
        out._t = {k: v.pin_memory() for k, v in self._t.items()}
        return out

    def flagged(self) -> np.ndarray:
        """[B] bool: lattices the last launch handed back to the general kernels (numbers beyond the float64 range of
        pass 1).  Synchronises; for tests and diagnostics."""
        h = self._h
        off = h["n_stream"] * 16 + h["t_units"] * 512 + h["total_rows"] * 32 + h["n_lattices"] * 16
        flags = self.ws[off:off + 4 * h["n_lattices"]].cpu().numpy().view(np.int32)
        return flags == int(self.c_struct().launches)

    def c_struct(self) -> _lib.Chunks:
        if self._struct is None:
            s = _lib.Chunks()
            for k, v in self._h.items():
                setattr(s, k, v)
            for k in self._FIELDS:
                setattr(s, k, self._t[k].data_ptr())
            if self.ws is not None:
                s.ws, s.ws_bytes = self.ws.data_ptr(), self.ws.numel()
            self._struct = s
        return self._struct


class LatticeBatch:
    """Packed lattices.  Tensors live on ``self.device``; ``meta`` is also kept on
    the host (numpy) because shapes and offsets are needed to size outputs."""

    _FIELDS = ("meta", "row_ptr", "arc_src", "arc_dst", "arc_label", "arc_w", "fwd_stream", "bwd_stream",
               "fwd_perm", "bwd_perm", "arc_sd", "arc_l16")
    _HEADER = ("n_lattices", "vocab", "max_rows", "max_tiles", "weighted", "reserved0", "total_rows", "total_arcs",
               "total_dp_arcs", "fwd_words", "bwd_words", "fwd_slots", "bwd_slots")

    def __init__(self, header: dict, tensors: dict):
        self._h = dict(header)
        self._t = dict(tensors)
        self.meta_host = self._t["meta"].detach().cpu().numpy().reshape(-1, _lib.META_WORDS).copy()
        self._struct = None
        self.chunks = None          # ChunkProgram of a batch of deep, narrow lattices (build_chunks / to)
        self._chunks_tried = False

    # ---------------------------------------------------------------- construction
    @staticmethod
    def _opts(n_threads=0, slots_per_lane=0, group_mode=0, no_compact=False):
        return _lib.PackOpts(int(n_threads), int(slots_per_lane), int(group_mode), 1 if no_compact else 0)

    @classmethod
    def _from_handle(cls, handle, device) -> "LatticeBatch":
        try:
            v = _lib.Batch()
            check(lib.nfst_packed_view(handle, C.byref(v)), "nfst_packed_view")
            B = v.n_lattices
            arrs = {
                "meta": _view(v.meta, B * _lib.META_WORDS, C.c_int32, np.int32),
                "row_ptr": _view(v.row_ptr, v.total_rows + B, C.c_int32, np.int32),
                "arc_src": _view(v.arc_src, v.total_arcs, C.c_int32, np.int32),
                "arc_dst": _view(v.arc_dst, v.total_arcs, C.c_int32, np.int32),
                "arc_label": _view(v.arc_label, v.total_arcs, C.c_int32, np.int32),
                "arc_w": _view(v.arc_w, v.total_arcs, C.c_float, np.float32) if v.weighted else None,
                "fwd_stream": _view(v.fwd_stream, v.fwd_words, C.c_int32, np.int32),
                "bwd_stream": _view(v.bwd_stream, v.bwd_words, C.c_int32, np.int32),
                "fwd_perm": _view(v.fwd_perm, v.fwd_slots, C.c_int32, np.int32),
                "bwd_perm": _view(v.bwd_perm, v.bwd_slots, C.c_int32, np.int32),
                "arc_sd": _view(v.arc_sd, v.total_arcs + 8, C.c_int32, np.int32),
                "arc_l16": _view(v.arc_l16, v.total_arcs + 8, C.c_int16, np.int16),
            }
            header = {k: int(getattr(v, k)) for k in cls._HEADER}
        finally:
            lib.nfst_packed_free(handle)
        tensors = {k: (None if a is None else torch.from_numpy(a)) for k, a in arrs.items()}
        out = cls(header, tensors)
        return out.to(device) if device is not None else out

    @classmethod
    def from_dense(cls, emission, transition, device=None, **pack_opts) -> "LatticeBatch":
        """emission ``[B, S+1, V]`` bool (or float log weights, -inf = no arc) and
        transition ``[B, S+1, V]`` int64, as ``set_masks`` receives them
        (scorers.py:877-885; collated as in util/dataset_reader.py:175-186)."""
        if device is None and isinstance(transition, torch.Tensor) and transition.is_cuda:
            device = transition.device
        if isinstance(transition, torch.Tensor) and transition.is_cuda and isinstance(emission, torch.Tensor) and emission.is_cuda:
            # tables that already live on the GPU (the reference's trainer moves the collated batch there before
            # set_masks, lightning.py:417) are packed there; only what the device packer does not take goes back to the host
            try:
                return cls.from_dense_device(emission, transition, **pack_opts).to(device)
            except _lib.NfstError as e:
                if e.code != -6:  # NFST_ERR_LIMIT: beyond the device packer (wide vocabulary, > 16384 states or pieces)
                    raise
        is_float = (emission.dtype.is_floating_point if isinstance(emission, torch.Tensor)
                    else np.issubdtype(np.asarray(emission).dtype, np.floating))
        em = _host(emission, np.float32 if is_float else np.bool_)
        tr = _host(transition, np.int64)
        if em.ndim != 3 or tr.shape != em.shape:
            raise ValueError("emission and transition must both be [B, S+1, V]")
        B, R, V = tr.shape
        opts = cls._opts(**pack_opts)
        handle = C.c_void_p()
        bad = C.c_int32(-1)
        rc = lib.nfst_pack_dense(em.ctypes.data, 1 if is_float else 0, tr.ctypes.data, B, R, V, C.byref(opts),
                                 C.byref(handle), C.byref(bad))
        check(rc, "nfst_pack_dense", bad.value)
        return cls._from_handle(handle, device)

    @classmethod
    def from_arcs(cls, n_rows, arc_off, src, label, dst, vocab: int, arc_w=None, device=None,
                  **pack_opts) -> "LatticeBatch":
        """Arc lists sorted by (src, label); lattice b owns arcs [arc_off[b], arc_off[b+1])."""
        n_rows = _host(n_rows, np.int32)
        arc_off = _host(arc_off, np.int64)
        src, label, dst = _host(src, np.int32), _host(label, np.int32), _host(dst, np.int32)
        w = None if arc_w is None else _host(arc_w, np.float32)
        B = n_rows.shape[0]
        if arc_off.shape[0] != B + 1 or not (src.shape == label.shape == dst.shape) or arc_off[-1] != src.shape[0]:
            raise ValueError("inconsistent arc list shapes")
        opts = cls._opts(**pack_opts)
        handle = C.c_void_p()
        bad = C.c_int32(-1)
        rc = lib.nfst_pack_arcs(n_rows.ctypes.data, arc_off.ctypes.data, src.ctypes.data, label.ctypes.data,
                                dst.ctypes.data, None if w is None else w.ctypes.data, B, int(vocab),
                                C.byref(opts), C.byref(handle), C.byref(bad))
        check(rc, "nfst_pack_arcs", bad.value)
        return cls._from_handle(handle, device)

    # ---------------------------------------------------------------- the packer on the device
    @classmethod
    def from_arcs_device(cls, n_rows, arc_off, src, label, dst, vocab: int, arc_w=None, device=None, **pack_opts) -> "LatticeBatch":
        """``from_arcs`` with the packer running on the GPU (``nfst_pack_device_plan`` / ``_emit``): arc lists sorted by
        (src, label) -- 12 bytes per arc, what a loader uploads instead of 47-byte-per-arc packed batches or 5 MB dense
        tables.  ``n_rows`` and ``arc_off`` are small host arrays; ``src / label / dst (/ arc_w)`` may be host arrays
        (uploaded here) or tensors already on the device.  The result is bit-identical to ``from_arcs``.  Raises
        ``NfstError(NFST_ERR_LIMIT)`` for what only the host packer takes (vocab + 2 > 2048, > 16384 reachable states
        or pieces of one sweep)."""
        n_rows = np.ascontiguousarray(_host(n_rows, np.int32))
        arc_off = np.ascontiguousarray(_host(arc_off, np.int64))
        B = int(n_rows.shape[0])
        if arc_off.shape[0] != B + 1:
            raise ValueError("inconsistent arc list shapes")
        if device is None:
            device = src.device if isinstance(src, torch.Tensor) and src.is_cuda else torch.device("cuda")
        dev = torch.device(device)

        def up(x, dt):
            if isinstance(x, torch.Tensor):
                return x.to(device=dev, dtype=dt).contiguous()
            return torch.from_numpy(np.ascontiguousarray(x, dtype={torch.int32: np.int32, torch.float32: np.float32, torch.int64: np.int64}[dt])).to(dev)

        src_d, label_d, dst_d = up(src, torch.int32), up(label, torch.int32), up(dst, torch.int32)
        w_d = None if arc_w is None else up(arc_w, torch.float32)
        total_arcs = int(arc_off[-1])
        if not (src_d.numel() == label_d.numel() == dst_d.numel() == total_arcs) or (w_d is not None and w_d.numel() != total_arcs):
            raise ValueError("inconsistent arc list shapes")
        row_off = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(n_rows, out=row_off[1:])
        total_rows = int(row_off[-1])
        small = torch.from_numpy(np.concatenate([row_off, arc_off, n_rows.astype(np.int64)])).to(dev)  # one upload
        row_off_d, arc_off_d = small[:B + 1], small[B + 1:2 * B + 2]
        n_rows_d = small[2 * B + 2:].to(torch.int32)
        opts = cls._opts(**pack_opts)
        stream = torch.cuda.current_stream(dev).cuda_stream
        arcs = _lib.ArcsDevice(n_rows_d.data_ptr(), row_off_d.data_ptr(), arc_off_d.data_ptr(), src_d.data_ptr(), label_d.data_ptr(),
                               dst_d.data_ptr(), None if w_d is None else w_d.data_ptr(), total_rows, total_arcs, B, int(vocab))
        ws_bytes = int(lib.nfst_pack_device_ws_bytes(B, total_rows, total_arcs))
        check(min(ws_bytes, 0), "nfst_pack_device_ws_bytes")
        ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.int32, device=dev)
        plan = torch.empty(B * (_lib.META_WORDS + 2), dtype=torch.int32, device=dev)  # meta | status | scratch rows
        meta_d, status_d, scratch_d = plan[:B * _lib.META_WORDS], plan[B * _lib.META_WORDS:B * (_lib.META_WORDS + 1)], plan[B * (_lib.META_WORDS + 1):]
        plan.zero_()
        check(lib.nfst_pack_device_plan(C.byref(arcs), C.byref(opts), ws.data_ptr(), ws_bytes, meta_d.data_ptr(), status_d.data_ptr(),
                                        scratch_d.data_ptr(), stream), "nfst_pack_device_plan")
        plan_h = plan.cpu().numpy()  # the one read-back: 18 words per lattice
        meta_h = np.ascontiguousarray(plan_h[:B * _lib.META_WORDS])
        status_h = np.ascontiguousarray(plan_h[B * _lib.META_WORDS:B * (_lib.META_WORDS + 1)])
        scratch_h = np.ascontiguousarray(plan_h[B * (_lib.META_WORDS + 1):])
        header = _lib.Batch()
        bad = C.c_int32(-1)
        check(lib.nfst_pack_device_layout(meta_h.ctypes.data, status_h.ctypes.data, scratch_h.ctypes.data, B, int(vocab),
                                          0 if w_d is None else 1, C.byref(header), C.byref(bad)), "nfst_pack_device_plan", bad.value)
        h = {k: int(getattr(header, k)) for k in cls._HEADER}
        sizes = cls._sizes(h)
        tensors = {k: torch.empty(n, dtype=cls._DTYPES[k], device=dev) for k, n in sizes.items() if k != "meta"}
        tensors["meta"] = torch.from_numpy(meta_h).to(dev)
        tensors.setdefault("arc_w", None)
        for k in cls._FIELDS:
            t = tensors[k]
            setattr(header, k, None if t is None or t.numel() == 0 else t.data_ptr())
        check(lib.nfst_pack_device_emit(C.byref(arcs), C.byref(opts), ws.data_ptr(), ws_bytes, tensors["meta"].data_ptr(), status_d.data_ptr(),
                                        C.byref(header), stream), "nfst_pack_device_emit")
        out = cls.__new__(cls)
        out._h, out._t, out._struct = h, tensors, None
        out.chunks, out._chunks_tried = None, True  # (packed on the device: the general kernels)
        out.meta_host = meta_h.reshape(-1, _lib.META_WORDS).copy()
        out._keep = (ws, src_d, label_d, dst_d, w_d, small, n_rows_d, plan)  # alive until the emit launch has run
        return out

    @classmethod
    def from_dense_device(cls, emission: torch.Tensor, transition: torch.Tensor, **pack_opts) -> "LatticeBatch":
        """``from_dense`` for tables that live on the GPU: reachable rows -> arc lists (``nfst_dense_to_arcs_count`` /
        ``_write``), then the device packer.  Two small read-backs (arc counts, the plan), no table leaves the GPU."""
        if not (emission.is_cuda and transition.is_cuda) or emission.dim() != 3 or transition.shape != emission.shape:
            raise ValueError("emission and transition must both be [B, S+1, V] tensors on the GPU")
        dev = transition.device
        B, R, V = transition.shape
        is_float = emission.dtype.is_floating_point
        em = emission.to(torch.float32).contiguous() if is_float else (emission if emission.dtype == torch.bool else emission != 0).contiguous()
        tr = transition.to(torch.int64).contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        reach = torch.empty(B * R, dtype=torch.uint8, device=dev)
        row_cnt = torch.empty(B * R, dtype=torch.int32, device=dev)
        cs = torch.empty(2 * B, dtype=torch.int32, device=dev)
        check(lib.nfst_dense_to_arcs_count(em.data_ptr(), 1 if is_float else 0, tr.data_ptr(), B, R, V, reach.data_ptr(), row_cnt.data_ptr(),
                                           cs.data_ptr(), cs[B:].data_ptr(), stream), "nfst_dense_to_arcs_count")
        cs_h = cs.cpu().numpy()
        if cs_h[B:].any():
            bad = int(np.nonzero(cs_h[B:])[0][0])
            raise _lib.NfstError(int(cs_h[B + bad]), "nfst_dense_to_arcs_count", bad)
        arc_off = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(cs_h[:B], out=arc_off[1:])
        A = int(arc_off[-1])
        arc_off_d = torch.from_numpy(arc_off).to(dev)
        src = torch.empty(A, dtype=torch.int32, device=dev)
        label, dst = torch.empty_like(src), torch.empty_like(src)
        w = torch.empty(A, dtype=torch.float32, device=dev) if is_float else None
        check(lib.nfst_dense_to_arcs_write(em.data_ptr(), 1 if is_float else 0, tr.data_ptr(), B, R, V, reach.data_ptr(), row_cnt.data_ptr(),
                                           arc_off_d.data_ptr(), src.data_ptr(), label.data_ptr(), dst.data_ptr(),
                                           None if w is None else w.data_ptr(), stream), "nfst_dense_to_arcs_write")
        return cls.from_arcs_device(np.full(B, R, dtype=np.int32), arc_off, src, label, dst, V, arc_w=w, device=dev, **pack_opts)

    @classmethod
    def from_synth(cls, lattices: Sequence, device=None, **pack_opts) -> "LatticeBatch":
        from . import synth
        n_rows, arc_off, src, label, dst, w = synth.batch_arcs(lattices)
        return cls.from_arcs(n_rows, arc_off, src, label, dst, lattices[0].vocab, arc_w=w, device=device, **pack_opts)

    _DTYPES = {"meta": torch.int32, "row_ptr": torch.int32, "arc_src": torch.int32, "arc_dst": torch.int32, "arc_label": torch.int32,
               "arc_w": torch.float32, "fwd_stream": torch.int32, "bwd_stream": torch.int32, "fwd_perm": torch.int32,
               "bwd_perm": torch.int32, "arc_sd": torch.int32, "arc_l16": torch.int16}

    @classmethod
    def _sizes(cls, h: dict) -> dict:
        """number of elements of every array of a batch with header ``h``"""
        B, A = h["n_lattices"], h["total_arcs"]
        n = {"meta": B * _lib.META_WORDS, "row_ptr": h["total_rows"] + B, "arc_src": A, "arc_dst": A, "arc_label": A,
             "fwd_stream": h["fwd_words"], "bwd_stream": h["bwd_words"], "fwd_perm": h["fwd_slots"], "bwd_perm": h["bwd_slots"],
             "arc_sd": A + 8, "arc_l16": A + 8}
        if h["weighted"]:
            n["arc_w"] = A
        return n

    @classmethod
    def concat(cls, batches: Sequence["LatticeBatch"], device=None, arena: "Optional[HostArena]" = None,
               n_threads: int = 0) -> "LatticeBatch":
        """One batch from already packed ones, without running the packer again: pack every
        example once (in a DataLoader worker, or cache it next to the ``.npz``) and build the
        step's batch by concatenation -- ``nfst_concat_packed``: memcpy + offset fix-ups in C, threads over
        the parts (the reference's collate pads and stacks the dense tables,
        util/dataset_reader.py:175-186).  The parts must agree in vocabulary and in being weighted;
        the result is bit-identical to packing the lattices as one batch.  ``arena``: write into the
        grow-only page-locked buffers of a ``HostArena`` (valid until the arena is used again)."""
        batches = list(batches)
        if not batches:
            raise ValueError("nothing to concatenate")
        cpu = [b if b.device.type == "cpu" else b.to("cpu") for b in batches]
        parts = (_lib.Batch * len(cpu))(*[b.c_struct() for b in cpu])
        total = _lib.Batch()
        rc = lib.nfst_concat_sizes(parts, len(cpu), C.byref(total))
        if rc == -1:
            raise ValueError("batches differ in vocabulary or in being weighted")
        check(rc, "nfst_concat_sizes")
        header = {k: int(getattr(total, k)) for k in cls._HEADER}
        sizes = cls._sizes(header)
        if arena is not None:
            tensors = arena.take(sizes, cls._DTYPES)
        else:
            tensors = {k: torch.empty(n, dtype=cls._DTYPES[k]) for k, n in sizes.items()}
        tensors.setdefault("arc_w", None)
        for k in cls._FIELDS:
            t = tensors[k]
            setattr(total, k, None if t is None or t.numel() == 0 else t.data_ptr())
        check(lib.nfst_concat_packed(parts, len(cpu), C.byref(total), int(n_threads)), "nfst_concat_packed")
        out = cls(header, tensors)
        return out.to(device) if device is not None else out

    def validate(self) -> None:
        """``nfst_validate_batch``: every offset, count, state / label / arc id that a kernel turns into an
        address without looking is inside the batch's arrays (host copy; O(words of the batch)).  Raises
        ``NfstError``.  ``load`` calls it: a truncated, stale or damaged sidecar is refused on the host
        instead of faulting on the GPU."""
        b = self if self.device.type == "cpu" else self.to("cpu")
        bad = C.c_int32(-1)
        check(lib.nfst_validate_batch(C.byref(b.c_struct()), C.byref(bad)), "nfst_validate_batch", bad.value)

    # ---------------------------------------------------------------- sidecar files
    # One flat file: [magic][abi, n header words, n arrays, reserved][header][per array: present, item size, elements,
    # byte offset, CRC-32C][arrays, 64-byte aligned].  Read back with one read(): no decompression, the arrays are
    # views of the buffer until they are gathered into a step's batch (``concat`` into a page-locked arena).
    _MAGIC = b"NFSTPK1\0"

    def save(self, fname: str) -> None:
        """Write the packed arrays next to the example's ``.npz``: the packer then runs once per example,
        offline or in a DataLoader worker, and a step's batch is ``concat`` of loaded sidecars.
        Every array carries a CRC-32C that ``load`` verifies."""
        arrs = {k: (None if v is None else np.ascontiguousarray(v.detach().cpu().numpy())) for k, v in self._t.items()}
        head = np.zeros(4 + len(self._HEADER) + 5 * len(self._FIELDS), dtype=np.int64)
        head[0:4] = (lib.nfst_abi_version(), len(self._HEADER), len(self._FIELDS), 0)
        head[4:4 + len(self._HEADER)] = [self._h[k] for k in self._HEADER]
        pos = (len(self._MAGIC) + head.nbytes + 63) & ~63
        for i, k in enumerate(self._FIELDS):
            a = arrs[k]
            e = 4 + len(self._HEADER) + 5 * i
            if a is None:
                continue
            crc = lib.nfst_crc32c(a.ctypes.data, a.nbytes, 0) if a.nbytes else 0
            head[e:e + 5] = (1, a.itemsize, a.size, pos, crc)
            pos = (pos + a.nbytes + 63) & ~63
        with open(fname, "wb") as f:
            f.write(self._MAGIC)
            f.write(head.tobytes())
            for i, k in enumerate(self._FIELDS):
                e = 4 + len(self._HEADER) + 5 * i
                if head[e]:
                    f.seek(int(head[e + 3]))
                    f.write(arrs[k].tobytes())
            f.truncate(pos)

    @classmethod
    def load(cls, fname: str, device=None, verify: bool = True, validate: bool = True, buffer: Optional[np.ndarray] = None) -> "LatticeBatch":
        """Read a sidecar written by ``save`` (one read; the arrays are views of the buffer).  ``verify`` (default): the CRC-32C of
        every array; ``validate`` (default): ``nfst_validate_batch`` -- a truncated, stale or damaged file raises
        ``ValueError`` here and never reaches a kernel.  Both off (header and size checks only) is for files this
        process wrote itself moments ago.  ``buffer``: a uint8 array the file is read into (the batch's arrays are then views
        of it: valid until the buffer is used again)."""
        try:
            # one read into private memory (a memory map's page faults cost more than the copy: 2.7 ms against 0.3 ms
            # per 0.9 MB file on the GPU boxes of this pool)
            if buffer is None:
                raw = np.fromfile(fname, dtype=np.uint8)
            else:  # a caller-owned byte buffer that is reused from file to file (io.PackedReader): no fresh pages to fault in
                with open(fname, "rb", buffering=0) as f:
                    n = f.readinto(memoryview(buffer))
                    if n == buffer.size and f.read(1):
                        raise ValueError("buffer too small")
                raw = buffer[:n]
        except (ValueError, OSError) as e:
            raise ValueError(f"{fname} is not a packed lattice file ({e})")
        nm = len(cls._MAGIC)
        fixed = nm + 8 * (4 + len(cls._HEADER) + 5 * len(cls._FIELDS))
        if raw.size < fixed or bytes(raw[:nm]) != cls._MAGIC:
            raise ValueError(f"{fname} is not a packed lattice file")
        head = raw[nm:fixed].view(np.int64)
        if int(head[0]) != lib.nfst_abi_version():
            raise ValueError(f"{fname} was packed for ABI {int(head[0])}, the library is ABI {lib.nfst_abi_version()}: pack it again")
        if int(head[1]) != len(cls._HEADER) or int(head[2]) != len(cls._FIELDS):
            raise ValueError(f"{fname}: unknown layout")
        header = {k: int(v) for k, v in zip(cls._HEADER, head[4:4 + len(cls._HEADER)])}
        if min(header.values()) < 0 or header["n_lattices"] <= 0:
            raise ValueError(f"{fname}: bad header")
        want = cls._sizes(header)
        tensors = {}
        for i, k in enumerate(cls._FIELDS):
            present, item, n, off, crc = (int(x) for x in head[4 + len(cls._HEADER) + 5 * i:][:5])
            if not present:
                if k in want:
                    raise ValueError(f"{fname}: array {k} is missing")
                tensors[k] = None
                continue
            dt = cls._DTYPES[k]
            if k not in want or n != want[k] or item != torch.empty(0, dtype=dt).element_size() or off < fixed or off % 64 or off + n * item > raw.size:
                raise ValueError(f"{fname}: array {k} does not match the header")
            a = raw[off:off + n * item]
            if verify and n and lib.nfst_crc32c(a.ctypes.data, a.nbytes, 0) != (crc & 0xFFFFFFFF):
                raise ValueError(f"{fname}: checksum of array {k} does not match (damaged or truncated file)")
            tensors[k] = torch.from_numpy(a.view({4: np.int32, 2: np.int16}[item] if dt != torch.float32 else np.float32))
        out = cls(header, tensors)
        if validate:
            try:
                out.validate()
            except _lib.NfstError as e:
                raise ValueError(f"{fname}: {e}")
        return out.to(device) if device is not None else out

    # ---------------------------------------------------------------- placement
    def to(self, device, non_blocking: bool = False, auto_chunks: bool = True) -> "LatticeBatch":
        device = torch.device(device)
        t = {k: (None if v is None else v.to(device, non_blocking=non_blocking)) for k, v in self._t.items()}
        out = LatticeBatch.__new__(LatticeBatch)
        out._h, out._t, out._struct = dict(self._h), t, None
        out.meta_host = self.meta_host  # shapes and offsets stay on the host: no device round trip
        # deep, narrow lattices: the chunked programs are cut on the way to the device (once per host batch; a quick no
        # for every other shape) and travel with the batch
        if device.type == "cuda" and self.device.type == "cpu" and not self.__dict__.get("_chunks_tried", False) and auto_chunks:
            try:
                self.build_chunks()
            except _lib.NfstError:  # (a batch the cutter refuses runs the general kernels)
                self.chunks, self._chunks_tried = None, True
        ck = self.__dict__.get("chunks")
        out.chunks = None if ck is None else ck.to(device, non_blocking=non_blocking)
        out._chunks_tried = self.__dict__.get("_chunks_tried", False)
        return out

    def build_chunks(self, force: bool = False, **opts) -> bool:
        """Cut the chunked programs of this batch: ``nfst_pack_chunks`` (host work, on the canonical arrays).  Without ``force`` only when the cost model
        of the two flavours says the chunked sweeps are faster (deep, narrow lattices); True when the batch has them now."""
        if self.device.type == "cuda":  # (packed on the device, or moved there without programs: cut from a host copy)
            ck = ChunkProgram.build(self.to("cpu", auto_chunks=False), force=force, **opts)
            self.chunks = None if ck is None else ck.to(self.device)
        else:
            self.chunks = ChunkProgram.build(self, force=force, **opts)
        self._chunks_tried = True
        self._struct = None
        return self.chunks is not None

    def pin_memory(self) -> "LatticeBatch":
        """Page-locked host copy: what ``to(device, non_blocking=True)`` needs to be asynchronous."""
        t = {k: (None if v is None else v.pin_memory()) for k, v in self._t.items()}
        out = LatticeBatch.__new__(LatticeBatch)
        out._h, out._t, out._struct, out.meta_host = dict(self._h), t, None, self.meta_host
        ck = self.__dict__.get("chunks")
        out.chunks = None if ck is None else ck.pin_memory()
        out._chunks_tried = self.__dict__.get("_chunks_tried", False)
        return out

    @property
    def device(self) -> torch.device:
        return self._t["meta"].device

    # ---------------------------------------------------------------- accessors
    def __getattr__(self, name):
        if name in ("_h", "_t", "chunks", "_chunks_tried"):
            raise AttributeError(name)
        if name in self._h:
            return self._h[name]
        if name in self._t:
            return self._t[name]
        raise AttributeError(name)

    @property
    def n_rows(self) -> np.ndarray:
        return self.meta_host[:, _lib.META_N_ROWS]

    @property
    def row_off(self) -> np.ndarray:
        return self.meta_host[:, _lib.META_ROW_OFF]

    @property
    def arc_off(self) -> np.ndarray:
        return self.meta_host[:, _lib.META_ARC_OFF]

    @property
    def n_arcs(self) -> np.ndarray:
        return self.meta_host[:, _lib.META_N_ARCS]

    @property
    def n_dp_arcs(self) -> np.ndarray:
        return self.meta_host[:, _lib.META_N_DP]

    @property
    def depth(self) -> np.ndarray:
        return self.meta_host[:, _lib.META_DEPTH]

    @property
    def sink(self) -> np.ndarray:
        return self.meta_host[:, _lib.META_SINK]

    @property
    def uniform_rows(self) -> bool:
        return bool(np.all(self.n_rows == self.n_rows[0]))

    def rows_view(self, x: torch.Tensor) -> torch.Tensor:
        """[total_rows, ...] -> [B, S+1, ...] when every lattice has the same row count."""
        if not self.uniform_rows:
            raise ValueError("lattices have different row counts; index with row_off instead")
        return x.reshape(self.n_lattices, int(self.n_rows[0]), *x.shape[1:])

    def arc_lattice(self) -> torch.Tensor:
        """int64 [total_arcs]: the lattice each canonical arc belongs to."""
        reps = torch.from_numpy(self.n_arcs.astype(np.int64))
        return torch.repeat_interleave(torch.arange(self.n_lattices), reps).to(self.device)

    # ---------------------------------------------------------------- C view
    def c_struct(self) -> _lib.Batch:
        """nfst_batch whose pointers are this batch's tensors (device memory)."""
        if self._struct is None:
            s = _lib.Batch()
            for k, v in self._h.items():
                setattr(s, k, v)
            for k in self._FIELDS:
                t = self._t[k]
                setattr(s, k, None if t is None or t.numel() == 0 else t.data_ptr())
            ck = self.__dict__.get("chunks")
            if ck is not None and ck.ws is not None:
                s.chunks = C.addressof(ck.c_struct())
            self._struct = s
        return self._struct

    def lds_bytes(self) -> int:
        return int(lib.nfst_lds_bytes(C.byref(self.c_struct())))

    def algorithmic_bytes(self, mode: str = "forward_backward") -> int:
        """log-Z only = 8 B/arc + 8 B/state; full
        forward-backward with arc posteriors = 32 B/arc + 24 B/state."""
        arcs = int(self.n_dp_arcs.sum())
        states = int(self.meta_host[:, _lib.META_N_REACH].sum())
        if mode == "backward":
            return 8 * arcs + 8 * states
        return 32 * arcs + 24 * states



class HostArena:
    """Grow-only host buffers, one per array of a packed batch, page-locked once (``pin=True``) and reused for every
    batch staged through them: ``concat(..., arena=...)`` gathers a step's batch straight into page-locked memory, and
    ``to(device, non_blocking=True)`` from there runs at the PCIe rate.  (Pinning per batch -- what
    ``tensor.pin_memory()`` does -- costs more than the copy itself: 0.47 GB/s measured in round 2.)  A batch taken
    from an arena is valid until the arena is used again."""

    def __init__(self, pin: bool = True, slack: float = 1.25):
        self.pin, self.slack, self.buf = bool(pin) and torch.cuda.is_available(), slack, {}

    def take(self, sizes: dict, dtypes: dict) -> dict:
        out = {}
        for k, n in sizes.items():
            b = self.buf.get(k)
            if b is None or b.numel() < n:
                b = torch.empty(max(int(n * self.slack), 64), dtype=dtypes[k], pin_memory=self.pin)
                self.buf[k] = b
            out[k] = b[:n]
        return out
