"""ctypes binding of ``libnfst_hip.so`` (the C ABI declared in include/nfst_hip.h).

The library is the product: there is no CPU fallback.  If it cannot be loaded
the import of this module fails with an explicit error.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- first, so that torch's bundled HIP runtime is the one the process uses

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library.  Only a tuning session (NFST_TUNING=1 in the environment) may put an experimental build of
# the same sources in its place through NFST_LIB (python -m nfst_amd.build --variant TAG -D...: A/B measurements
# inside one GPU box call); without NFST_TUNING the variable is ignored.
LIB_PATH = os.path.join(_HERE, "lib", "libnfst_hip.so")
if os.environ.get("NFST_TUNING") == "1" and os.environ.get("NFST_LIB"):
    LIB_PATH = os.environ["NFST_LIB"]

META_WORDS = 16
CHK_META_WORDS = 8
CHK_C, CHK_F, CHK_R, CHK_NPOS, CHK_TAB_OFF, CHK_STREAM_OFF, CHK_POS_OFF, CHK_T_OFF = range(8)
(META_ROW_OFF, META_N_ROWS, META_ARC_OFF, META_N_ARCS, META_FWD_OFF, META_FWD_TILES, META_BWD_OFF,
 META_BWD_TILES, META_SINK, META_N_REACH, META_DEPTH, META_N_DP, META_FWD_U, META_BWD_U, META_FWD_SLOT_OFF,
 META_BWD_SLOT_OFF) = range(16)

OK = 0
ERR_LENGTH = -9


class NfstError(RuntimeError):
    def __init__(self, code: int, where: str, lattice: int = -1):
        self.code = code
        self.lattice = lattice
        msg = lib.nfst_strerror(code).decode()
        if lattice >= 0:
            msg += f" (lattice {lattice})"
        super().__init__(f"{where}: {msg} [code {code}]")


class Batch(C.Structure):
    _fields_ = [
        ("n_lattices", C.c_int32), ("vocab", C.c_int32), ("max_rows", C.c_int32), ("max_tiles", C.c_int32),
        ("weighted", C.c_int32), ("reserved0", C.c_int32),
        ("total_rows", C.c_int64), ("total_arcs", C.c_int64), ("total_dp_arcs", C.c_int64),
        ("fwd_words", C.c_int64), ("bwd_words", C.c_int64), ("fwd_slots", C.c_int64), ("bwd_slots", C.c_int64),
        ("meta", C.c_void_p), ("row_ptr", C.c_void_p), ("arc_src", C.c_void_p), ("arc_dst", C.c_void_p),
        ("arc_label", C.c_void_p), ("arc_w", C.c_void_p), ("fwd_stream", C.c_void_p), ("bwd_stream", C.c_void_p),
        ("fwd_perm", C.c_void_p), ("bwd_perm", C.c_void_p), ("arc_sd", C.c_void_p), ("arc_l16", C.c_void_p),
        ("chunks", C.c_void_p), ("only", C.c_void_p), ("only_tag", C.c_int32), ("reserved2", C.c_int32),
    ]


class Chunks(C.Structure):  # nfst_chunks: the chunked programs of a batch of deep, narrow lattices
    _fields_ = [
        ("n_lattices", C.c_int32), ("threads", C.c_int32), ("lds_bytes", C.c_int32), ("launches", C.c_int32),
        ("n_tab", C.c_int64), ("n_stream", C.c_int64), ("n_pos", C.c_int64), ("t_units", C.c_int64),
        ("total_rows", C.c_int64), ("total_arcs", C.c_int64),
        ("meta", C.c_void_p), ("tab", C.c_void_p), ("stream", C.c_void_p), ("pos", C.c_void_p), ("label", C.c_void_p),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
    ]


class ChunkOpts(C.Structure):
    _fields_ = [("threads", C.c_int32), ("lds_bytes", C.c_int32), ("force", C.c_int32), ("max_chunks", C.c_int32),
                ("n_threads", C.c_int32), ("reserved", C.c_int32)]


class Scores(C.Structure):
    _fields_ = [("theta", C.c_void_p), ("theta_stride", C.c_int64), ("arc_scores", C.c_void_p), ("reserved_ws", C.c_void_p),
                ("reserved_flag", C.c_int64)]


class StepExtras(C.Structure):
    _fields_ = [("value_state", C.c_void_p), ("accumulated", C.c_void_p), ("vocab_use", C.c_void_p),
                ("insertion_mark", C.c_int32), ("insert_threshold", C.c_int32), ("insert_penalty", C.c_float),
                ("length_threshold", C.c_int32), ("length_penalty", C.c_float), ("length", C.c_int32),
                ("not_pad", C.c_void_p)]


class ArcsDevice(C.Structure):
    _fields_ = [("n_rows", C.c_void_p), ("row_off", C.c_void_p), ("arc_off", C.c_void_p), ("src", C.c_void_p), ("label", C.c_void_p),
                ("dst", C.c_void_p), ("arc_w", C.c_void_p), ("total_rows", C.c_int64), ("total_arcs", C.c_int64),
                ("n_lattices", C.c_int32), ("vocab", C.c_int32)]


class PackOpts(C.Structure):
    _fields_ = [("n_threads", C.c_int32), ("slots_per_lane", C.c_int32), ("group_mode", C.c_int32),
                ("reserved1", C.c_int32)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m nfst_amd.build` (hipcc --offload-arch=gfx950). "
            "nfst_amd has no CPU fallback."
        )
    l = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
    BP, SP = C.POINTER(Batch), C.POINTER(Scores)
    sig = {
        "nfst_strerror": (C.c_char_p, [C.c_int]),
        "nfst_abi_version": (C.c_int, []),
        "nfst_sizeof": (C.c_int, [C.c_char_p]),
        "nfst_device_available": (C.c_int, []),
        "nfst_tuning_set": (C.c_int, [C.c_char_p, C.c_int]),
        "nfst_pack_dense": (C.c_int, [vp, C.c_int, vp, i32, i32, i32, C.POINTER(PackOpts), C.POINTER(vp), C.POINTER(i32)]),
        "nfst_pack_arcs": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, C.POINTER(PackOpts), C.POINTER(vp), C.POINTER(i32)]),
        "nfst_packed_view": (C.c_int, [vp, BP]),
        "nfst_dense_to_arcs_count": (C.c_int, [vp, C.c_int, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
        "nfst_dense_to_arcs_write": (C.c_int, [vp, C.c_int, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "nfst_pack_device_ws_bytes": (i64, [i32, i64, i64]),
        "nfst_pack_device_plan": (C.c_int, [C.POINTER(ArcsDevice), C.POINTER(PackOpts), vp, i64, vp, vp, vp, vp]),
        "nfst_pack_device_layout": (C.c_int, [vp, vp, vp, i32, i32, i32, BP, C.POINTER(i32)]),
        "nfst_pack_device_emit": (C.c_int, [C.POINTER(ArcsDevice), C.POINTER(PackOpts), vp, i64, vp, vp, BP, vp]),
        "nfst_validate_batch": (C.c_int, [BP, C.POINTER(i32)]),
        "nfst_crc32c": (C.c_uint32, [vp, i64, C.c_uint32]),
        "nfst_concat_sizes": (C.c_int, [BP, i32, BP]),
        "nfst_concat_packed": (C.c_int, [BP, i32, BP, i32]),
        "nfst_packed_free": (None, [vp]),
        "nfst_pack_chunks": (C.c_int, [BP, C.POINTER(ChunkOpts), C.POINTER(vp)]),
        "nfst_chunks_view": (C.c_int, [vp, C.POINTER(Chunks)]),
        "nfst_chunks_free": (None, [vp]),
        "nfst_chunks_ws_bytes": (i64, [C.POINTER(Chunks)]),
        "nfst_lds_bytes": (i64, [BP]),
        "nfst_backward": (C.c_int, [BP, SP, vp, vp, vp, vp, vp]),
        "nfst_forward_backward": (C.c_int, [BP, SP, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp]),
        "nfst_viterbi": (C.c_int, [BP, SP, vp, vp, vp, vp, i32, i32, vp]),
        "nfst_sample_paths": (C.c_int, [BP, SP, vp, vp, i32, i32, vp, u64, i32, vp, vp, vp, vp, vp, vp]),
        "nfst_score_paths": (C.c_int, [BP, SP, vp, i32, i32, vp, vp, vp]),
        "nfst_step": (C.c_int, [BP, vp, vp, vp, i32, vp]),
        "nfst_emission_mask": (C.c_int, [BP, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
        "nfst_beta_logits": (C.c_int, [BP, vp, vp, vp, i32, vp]),
        "nfst_proposal_step": (C.c_int, [BP, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, C.POINTER(StepExtras), vp, vp,
                                         vp, vp, vp, i32, vp]),
        "nfst_proposal_step_backward": (C.c_int, [BP, vp, vp, vp, vp, vp, vp, i32, f32, vp, vp, i32, vp]),
        "nfst_neural_ws_floats": (i64, [BP, i32]),
        "nfst_backward_neural": (C.c_int, [BP, vp, vp, vp, i32, vp, vp, vp, vp]),
        "nfst_neural_grad_ws_floats": (i64, [BP, i32]),
        "nfst_backward_neural_grad": (C.c_int, [BP, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "nfst_gather_label_scores": (C.c_int, [BP, SP, vp, vp]),
        "nfst_path_logprob": (C.c_int, [vp, vp, i64, i32, i32, i32, i32, i32, i32, f32, i32, f32, i32, vp, vp]),
        "nfst_path_logprob_backward": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, f32, i32, f32, i32, vp, vp]),
        "nfst_iwae": (C.c_int, [vp, vp, i32, i32, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(l, name)  # AttributeError here = the library does not match the header
        fn.restype = res
        fn.argtypes = args
    return l, list(sig)


def header_abi_version() -> int:
    """NFST_ABI_VERSION of include/nfst_hip.h -- the one place the ABI version is written down."""
    import re
    with open(os.path.join(os.path.dirname(_HERE), "include", "nfst_hip.h")) as f:
        return int(re.search(r"#define\s+NFST_ABI_VERSION\s+(\d+)", f.read()).group(1))


lib, EXPORTS = _load()
ABI_VERSION = header_abi_version()
if lib.nfst_abi_version() != ABI_VERSION:
    raise ImportError(f"libnfst_hip.so has ABI {lib.nfst_abi_version()}, include/nfst_hip.h says {ABI_VERSION}; "
                      "rebuild with `python -m nfst_amd.build --force`")


for _name, _cls in (("nfst_batch", Batch), ("nfst_scores", Scores), ("nfst_chunks", Chunks), ("nfst_chunk_opts", ChunkOpts),
                    ("nfst_pack_opts", PackOpts), ("nfst_step_extras", StepExtras), ("nfst_arcs_device", ArcsDevice)):
    if lib.nfst_sizeof(_name.encode()) != C.sizeof(_cls):  # (the entry points copy whole structs: a drifted declaration corrupts memory)
        raise ImportError(f"struct {_name}: the library has {lib.nfst_sizeof(_name.encode())} bytes, nfst_amd/_lib.py declares {C.sizeof(_cls)}")


def check(code: int, where: str, lattice: int = -1) -> None:
    if code != OK:
        raise NfstError(code, where, lattice)


class tuning:
    """``with tuning(tw=0, precise=0): ...`` -- launcher switches for tests and A/B measurements
    (``nfst_tuning_set``, include/nfst_hip.h); the defaults come back on exit."""
    DEFAULTS = dict(tw=1, fused=1, xcache=1, precise=-1, neu_pack=1, neu_small=1, neu_bf16=1, chunked=1)

    def __init__(self, **kw):
        unknown = set(kw) - set(self.DEFAULTS)
        if unknown:
            raise ValueError(f"unknown tuning switches {sorted(unknown)}")
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            check(lib.nfst_tuning_set(k.encode(), int(v)), "nfst_tuning_set")
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            check(lib.nfst_tuning_set(k.encode(), int(self.DEFAULTS[k])), "nfst_tuning_set")
        return False
