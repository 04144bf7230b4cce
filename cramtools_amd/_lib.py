"""ctypes binding of libcramgpu.so (C ABI: include/cramgpu.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C cramtools_amd/csrc`.  There is no fallback:
a missing library raises ImportError-like RuntimeError, a missing GPU makes every compute call fail with
CRAMGPU_E_NO_DEVICE.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcramgpu.so")
CSRC = os.path.join(_HERE, "csrc")

E_INPUT_LENGTH, E_OUT_TOO_SMALL, E_INSUFFICIENT, E_ORDER, E_CORRUPT = -1, -2, -3, -4, -5
E_ARG, E_HIP, E_NO_DEVICE, E_JAVA_THROWS, E_UNSUPPORTED = -6, -7, -8, -9, -10
ARENA_ALIGN = 16
RF_UNMAPPED, RF_NULL_QUALS, RF_NO_BASES, RF_UNKNOWN_BASES, RF_FORCE_QS, RF_SKIP = 1, 2, 4, 8, 16, 32

# the reference's exception messages; %-forms are filled by the caller that holds N / X
MESSAGES = {
    E_INPUT_LENGTH: "Incorrect input length.",
    E_OUT_TOO_SMALL: "Output buffer too small to fit %d bytes.",
    E_INSUFFICIENT: "Insufficient buffer size.",
    E_ORDER: "Unknown rANS order: %s",
    E_CORRUPT: "Malformed rANS stream",
    E_ARG: "Bad argument",
    E_HIP: "HIP runtime error",
    E_NO_DEVICE: "No MI355X device visible (libcramgpu has no CPU path)",
    E_JAVA_THROWS: "The reference would throw on this record",
    E_UNSUPPORTED: "Lossy-model term needs the coverage / mismatch tracks, which libcramgpu does not build",
}


class CramGpuError(RuntimeError):
    """`arg` fills the N / X of the reference's message (RANS#uncompress "…to fit N bytes.", RANS.ORDER.fromInt
    "Unknown rANS order: X"); without it the placeholder stays generic."""

    def __init__(self, code, detail="", arg=None):
        self.code = code
        msg = MESSAGES.get(code, "error %d" % code)
        if "%" in msg:
            msg = msg % (arg,) if arg is not None else msg.replace(" %d bytes", " the raw bytes").replace(": %s", "")
        super().__init__(msg + (": " + detail if detail else ""))


class Block(C.Structure):
    _fields_ = [("in_off", C.c_int64), ("out_off", C.c_int64), ("in_len", C.c_int32), ("out_cap", C.c_int32),
                ("order", C.c_int32), ("reserved", C.c_int32)]


def _block_dtype():
    import numpy as np
    return np.dtype([("in_off", np.int64), ("out_off", np.int64), ("in_len", np.int32), ("out_cap", np.int32),
                     ("order", np.int32), ("reserved", np.int32)])


BLOCK_DTYPE = _block_dtype()  # cramgpu_block as a numpy record (block tables read back from the device)


class RefWin(C.Structure):
    _fields_ = [("ref_off", C.c_int64), ("ref_len", C.c_int64), ("ref_start0", C.c_int64)]


class Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_cont", C.c_int32), ("n_bases", C.c_int64), ("max_read_len", C.c_int32),
                ("reserved", C.c_int32), ("bases", C.c_void_p), ("quals", C.c_void_p), ("base_off", C.c_void_p),
                ("read_len", C.c_void_p), ("aln_start", C.c_void_p), ("rflags", C.c_void_p), ("mapq", C.c_void_p),
                ("cigar", C.c_void_p), ("cigar_off", C.c_void_p), ("n_cigar", C.c_void_p), ("read_cont", C.c_void_p),
                ("win", C.c_void_p)]


class Features(C.Structure):
    _fields_ = [("op", C.c_void_p), ("pos", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p), ("off", C.c_void_p),
                ("cap", C.c_int64)]


class RecordCols(C.Structure):
    _fields_ = [("bam_flags", C.c_void_p), ("comp_flags", C.c_void_p), ("ref_id", C.c_void_p), ("read_group", C.c_void_p),
                ("mate_flags", C.c_void_p), ("mate_ref", C.c_void_p), ("mate_start", C.c_void_p), ("tlen", C.c_void_p),
                ("next_frag", C.c_void_p), ("tag_list", C.c_void_p), ("cont_multiref", C.c_void_p),
                ("cont_ap_base", C.c_void_p), ("ap_delta", C.c_int32), ("flags", C.c_int32),
                ("names", C.c_void_p), ("name_off", C.c_void_p), ("name_len", C.c_void_p), ("capture_names", C.c_int32),
                ("reserved", C.c_int32)]


class ReadOut(C.Structure):
    _fields_ = [("read_len", C.c_void_p), ("aln_start", C.c_void_p), ("mapq", C.c_void_p), ("base_off", C.c_void_p),
                ("bases", C.c_void_p), ("quals", C.c_void_p), ("bases_cap", C.c_int64), ("seq", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq_cap", C.c_int64), ("rflags", C.c_void_p), ("name_off", C.c_void_p),
                ("name_len", C.c_void_p)]


N_SERIES = 30
WS_ALIAS_QS = 1
SERIES_ID = dict(BF=0, AP=1, FP=2, FC=3, QS=4, DL=5, BA=6, TN=7, NF=8, RL=9, RG=10, MQ=11, RN=12, NP=13, TS=14, FN=15, BS=16,
                 IN=17, TC=18, MF=19, NS=20, CF=21, TV=22, TM=23, TL=24, RI=25, RS=26, SC=27, HC=28, PD=29)


class BamCols(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("ref_id", "aln_start", "aln_end", "mapq", "bam_flags", "mate_flags", "mate_ref",
                                          "mate_start", "tlen", "read_len", "base_off", "bases", "quals", "qual_status",
                                          "n_cigar", "cigar_off", "cigar", "names", "name_off")] + \
               [("name_base", C.c_int64), ("tags", C.c_void_p), ("tag_off", C.c_void_p)]


class BamOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("ref_id", "aln_start", "mapq", "bam_flags", "mate_ref", "mate_start", "tlen",
                                          "read_len", "base_off", "bases", "quals")] + [("bases_cap", C.c_int64)] + \
               [(k, C.c_void_p) for k in ("n_cigar", "cigar_off", "cigar")] + [("cigar_cap", C.c_int64)] + \
               [(k, C.c_void_p) for k in ("aln_span", "rflags", "name_off", "name_len", "tag_off", "tag_len")]


class HostReads(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_cont", C.c_int32), ("n_bases", C.c_int64), ("n_cigar_ops", C.c_int64),
                ("bases", C.c_void_p), ("quals", C.c_void_p), ("read_len", C.c_void_p), ("aln_start", C.c_void_p),
                ("rflags", C.c_void_p), ("mapq", C.c_void_p), ("cigar", C.c_void_p), ("n_cigar", C.c_void_p),
                ("cont_reads", C.c_void_p), ("cont_contig", C.c_void_p)]


class HostReadResults(C.Structure):
    _fields_ = [("feats", Features), ("matrices", C.c_void_p), ("sub_hist", C.c_void_p), ("scores", C.c_void_p),
                ("force", C.c_void_p), ("status", C.c_void_p)]


class HostRecords(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("n_cont", C.c_int32), ("n_seq", C.c_int64), ("read_len", C.c_void_p),
                ("aln_start", C.c_void_p), ("rflags", C.c_void_p), ("cont_reads", C.c_void_p), ("cont_contig", C.c_void_p),
                ("cont_win_start0", C.c_void_p), ("cont_win_len", C.c_void_p), ("feats", Features), ("seq", C.c_void_p),
                ("seq_off", C.c_void_p), ("matrices", C.c_void_p), ("default_qs", C.c_int32), ("reserved", C.c_int32)]


class HostRestoreResults(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("scores", C.c_void_p), ("status_bases", C.c_void_p), ("status_quals", C.c_void_p)]


class Policy(C.Structure):
    _fields_ = [("read_cat", C.c_int32), ("read_param", C.c_int32), ("treatment", C.c_int32), ("n_base_cat", C.c_int32),
                ("base_cat", C.c_int32 * 8), ("base_param", C.c_int32 * 8)]


# every symbol include/cramgpu.h declares: name -> (restype, argtypes)
P, I32, I64, U8 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint8
SYMBOLS = {
    "cramgpu_open": (P, [I32]),
    "cramgpu_close": (None, [P]),
    "cramgpu_last_error": (C.c_char_p, []),
    "cramgpu_device_count": (I32, []),
    "cramgpu_info": (I32, [P, P, P]),
    "cramgpu_rans_bound": (I32, [I32]),
    "cramgpu_rans_compress": (I32, [P, I32, I32, P, I32]),
    "cramgpu_rans_uncompress": (I32, [P, I32, P, I32]),
    "cramgpu_rans_raw_size": (I32, [P, I32]),
    "cramgpu_rans_encode_dev": (I32, [P, P, P, P, I32, P, P, P]),
    "cramgpu_rans_encode_static": (I32, [P, P, P, P, I32, P, P, P]),
    "cramgpu_stream_wait_staged": (I32, [P, P]),
    "cramgpu_rans_encode_series_dev": (I32, [P, P, I64, P, P, P, I32, P, P, I32, P, I64, P, P, P, P, P]),
    "cramgpu_rans_decode_dev": (I32, [P, P, P, P, I32, P, P]),
    "cramgpu_gather_dev": (I32, [P, P, P, P, I32, I32, I32, P, I64, P, P]),
    "cramgpu_rans_encode_batch": (I32, [P, P, I64, P, I64, P, I32, P]),
    "cramgpu_rans_decode_batch": (I32, [P, P, I64, P, I64, P, I32, P]),
    "cramgpu_host_alloc": (P, [I64]),
    "cramgpu_host_free": (None, [P]),
    "cramgpu_rans_submit": (P, [P, I32, P, I64, P, I64, P, I32, P]),
    "cramgpu_rans_submit_packed": (P, [P, P, I64, P, I64, P, I32, P, P]),
    "cramgpu_job_poll": (I32, [P]),
    "cramgpu_job_wait": (I32, [P]),
    "cramgpu_coalescer_open": (P, [P, I32]),
    "cramgpu_coalescer_close": (None, [P]),
    "cramgpu_rans_enqueue": (I64, [P, I32, P, I64, P, I64, P, I32, P, P]),
    "cramgpu_ticket_poll": (I32, [P, I64]),
    "cramgpu_ticket_wait": (I32, [P, I64]),
    "cramgpu_rans_coalesced": (I32, [P, I32, P, I64, P, I64, P, I32, P, P]),
    "cramgpu_coalesce_stats": (I32, [P, P, P, P]),
    "cramgpu_pool_open": (P, [I32, I32]),
    "cramgpu_pool_close": (None, [P]),
    "cramgpu_pool_devices": (I32, [P]),
    "cramgpu_pool_submit": (I32, [P, I64, I32, P, I64, P, I64, P, I32, P, P]),
    "cramgpu_pool_next": (I32, [P, P]),
    "cramgpu_extract_features_dev": (I64, [P, P, P, P, P, P]),
    "cramgpu_substitution_matrix": (None, [P, P]),
    "cramgpu_assign_codes_dev": (I32, [P, I32, P, I32, P, P, P]),
    "cramgpu_matrices_assign_codes_dev": (I32, [P, I32, P, I32, P, P, P, P]),
    "cramgpu_restore_bases_dev": (I32, [P, I32, P, P, P, P, I32, P, P, P, P, I64, P, P, P, P, P, P]),
    "cramgpu_restore_quals_dev": (I32, [P, I32, P, P, P, U8, P, P, P, P]),
    "cramgpu_parse_policies": (I32, [C.c_char_p, P, I32]),
    "cramgpu_lossy_quals_dev": (I32, [P, P, P, P, I32, P, I32, P, P, P, P]),
    "cramgpu_tracks_windows": (I32, [I32, P, P, P, P, P, I32, I64, I32, P, P]),
    "cramgpu_merge_quality_features_dev": (I64, [P, I32, P, P, P, P, P, P, P, P]),
    "cramgpu_write_series_dev": (I32, [P, P, P, P, P, I64, P, P, P, P]),
    "cramgpu_read_series_dev": (I32, [P, I32, I32, P, P, P, P, I64, P, P, P, P, P]),
    "cramgpu_cigars_dev": (I32, [P, I32, P, P, P, P, P, P, P, I64, P, P, P]),
    "cramgpu_restore_mates_dev": (I32, [P, I32, I32, P, P, P, P, P, P, P]),
    "cramgpu_set_mate_info_dev": (I32, [P, I32, I32, P, P, P, P, P, P, P, P]),
    "cramgpu_content_digests_dev": (I32, [P, I32, I32, P, P, P, P, P, P, P]),
    "cramgpu_bam_encode_dev": (I32, [P, I32, P, P, I64, P, P, P]),
    "cramgpu_bam_decode_dev": (I32, [P, I32, P, P, P, P, P]),
    "cramgpu_bam_index": (I64, [P, I64, P, I64]),
    "cramgpu_bam_index_dev": (I32, [P, P, I64, I32, P, I64, P, P, P]),
    "cramgpu_slice_md5_dev": (I32, [P, I32, P, P, P, P, P, P]),
    "cramgpu_gunzip_dev": (I32, [P, P, P, P, I32, P, P]),
    "cramgpu_gzip_dev": (I32, [P, P, P, P, I32, P, I32, P]),
    "cramgpu_gzip_bound": (I64, [I64]),
    "cramgpu_bgzf_index": (I64, [P, I64, P, P, I64]),
    "cramgpu_ref_upload": (I32, [P, I32, P, I64]),
    "cramgpu_ref_drop": (I32, [P, I32]),
    "cramgpu_submit_reads": (P, [P, P, C.c_char_p, I32, P]),
    "cramgpu_submit_restore": (P, [P, P, P]),
}

_lib = None


def build(force=False):
    """Compile libcramgpu.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "cramgpu.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC, "-j8", "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libcramgpu.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                               "cramtools_amd has no CPU fallback")
        # torch ships its own libamdhip64 (same soname as /opt/rocm's).  It has to be in the process BEFORE libcramgpu
        # is loaded so that both resolve to ONE HIP runtime: two runtimes in a process cannot share device pointers,
        # and the second one to initialise sees no GPU.  Hosts without torch (the JNI path) use /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = LIB_PATH
        prof = os.path.join(_HERE, "libcramgpu_prof.so")  # make -C cramtools_amd/csrc prof: in-kernel cycle stamps
        if os.environ.get("CRAMGPU_PROF") == "1" and os.path.exists(prof):
            path = prof
        if os.environ.get("CRAMGPU_LIB"):  # development: another build of the same library (kernel experiments)
            path = os.environ["CRAMGPU_LIB"]
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return (lib().cramgpu_last_error() or b"").decode(errors="replace")


def check(rc):
    if rc < 0:
        raise CramGpuError(int(rc), last_error())
    return rc
