"""CPU-side checks of the drop-in boundary: libcramgpu.so loads and exports exactly what include/cramgpu.h declares."""
import ctypes
import os
import re

from cramtools_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "cramgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cramgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.SYMBOLS) == names  # the Python binding covers the whole header


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_lib.Block) == 32
    assert ctypes.sizeof(_lib.RefWin) == 24
    assert ctypes.sizeof(_lib.Policy) == 80
    assert ctypes.sizeof(_lib.Features) == 48
    assert ctypes.sizeof(_lib.Reads) == 24 + 12 * 8
    assert ctypes.sizeof(_lib.HostReads) == 24 + 10 * 8
    assert ctypes.sizeof(_lib.HostReadResults) == 48 + 5 * 8
    assert ctypes.sizeof(_lib.HostRecords) == 16 + 7 * 8 + 48 + 3 * 8 + 8
    assert ctypes.sizeof(_lib.HostRestoreResults) == 32
    assert ctypes.sizeof(_lib.RecordCols) == 12 * 8 + 8 + 3 * 8 + 8
    assert ctypes.sizeof(_lib.BamCols) == 22 * 8
    assert ctypes.sizeof(_lib.BamOut) == 22 * 8
    assert ctypes.sizeof(_lib.ReadOut) == 13 * 8


def test_host_only_entry_points():
    L = _lib.lib()
    assert L.cramgpu_rans_bound(0) == 198151
    assert L.cramgpu_rans_bound(1000) == int(1.05 * 1000 + 198151)
    buf = (ctypes.c_uint8 * 9)(1, 5, 0, 0, 0, 0x39, 0x30, 0, 0)
    assert L.cramgpu_rans_raw_size(buf, 9) == 12345
    assert L.cramgpu_rans_raw_size(buf, 0) == 0
    assert L.cramgpu_rans_raw_size(buf, 5) == _lib.E_INPUT_LENGTH
