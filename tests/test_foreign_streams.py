"""Streams of a foreign (htslib-style, tables summing to 4096) rANS encoder: the oracle's decoder takes them (CPU), and
the GPU decoder gives the same bytes (-m gpu)."""
import pytest

from foreign_streams import cases


def test_oracle_decodes_foreign_streams(orc):
    for name, data, stream in cases():
        assert orc.unrans(stream) == data, name


def check_library_decodes(xc):
    """Body shared with tests/test_glue_hostsim.py (the same kernels compiled for the CPU)."""
    cs = cases()
    eng = xc.Engine(0)
    try:
        got = eng.uncompress_many([s for _, _, s in cs])
    finally:
        eng.close()
    for (name, data, stream), g in zip(cs, got):
        assert g == data, name
        assert xc.unrans(stream) == data, name


@pytest.mark.gpu
def test_gpu_decodes_foreign_streams():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from cramtools_amd import external_compression as xc
    check_library_decodes(xc)
