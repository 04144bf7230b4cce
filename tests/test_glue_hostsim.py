"""The Python glue (ctypes binding, external_compression.py, records.py, bench.Step) and the BODIES of the GPU parity
tests, exercised on the CPU against the host build of the kernel sources (tests/cpu_sim/libcramgpu_sim.so).

This is test infrastructure: the package never loads that library; here the test swaps it in by hand so that a typo in
an argtypes list or in a GPU test is found without a GPU.  The real parity runs are the `gpu`-marked tests.
"""
import ctypes
import os
import subprocess
import types

import numpy as np
import pytest
import torch

# the long legs (random containers, a whole bench step) take minutes on the thread-per-lane host build
SLOW = pytest.mark.skipif(not os.environ.get("CRAMGPU_SLOW_TESTS"), reason="set CRAMGPU_SLOW_TESTS=1 (several minutes)")

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "cpu_sim")


@pytest.fixture(scope="module")
def hostsim():
    from cramtools_amd import _lib
    subprocess.check_call(["make", "-C", SIM, "-s", "libcramgpu_sim.so"])
    saved = (_lib._lib, _lib.LIB_PATH, torch.cuda.current_stream, torch.cuda.synchronize)
    _lib._lib, _lib.LIB_PATH = None, os.path.join(SIM, "libcramgpu_sim.so")
    _lib.lib()
    torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=None)
    torch.cuda.synchronize = lambda *a, **k: None
    yield
    _lib._lib, _lib.LIB_PATH, torch.cuda.current_stream, torch.cuda.synchronize = saved


def test_rans_gpu_test_bodies(hostsim, kat1, orc, monkeypatch):
    import test_rans_gpu as t
    monkeypatch.setattr(t, "DEV", "cpu")
    from cramtools_amd import external_compression as xc
    eng = xc.Engine(0)
    try:
        t.test_kat1_hex(kat1, xc)
        t.test_empty_short_and_errors(xc, orc)
        t.test_decoder_accepts_oracle_streams(eng, orc)
        t.test_corrupt_streams_do_not_crash(eng, orc)
        t.test_async_jobs(eng, orc)
        t.test_gather_reports_overflow(eng, orc)
        import test_foreign_streams as tf
        tf.check_library_decodes(xc)  # tables summing to 4096, F = 4096 in both spellings
        # device-arena entry points + gather, as bench.py drives them
        datas = [bytes([65 + (i * 7 + k) % 5 for i in range(300 + 17 * k)]) for k in range(6)]
        orders = [k & 1 for k in range(6)]
        blocks, in_bytes, out_bytes = xc.Engine.plan_encode([len(d) for d in datas], orders)
        d_in = torch.zeros(in_bytes, dtype=torch.uint8)
        for i, d in enumerate(datas):
            d_in[blocks[i].in_off:blocks[i].in_off + len(d)] = torch.as_tensor(np.frombuffer(d, np.uint8).copy())
        d_out = torch.zeros(out_bytes, dtype=torch.uint8)
        d_len = torch.zeros(6, dtype=torch.int32)
        d_pos = torch.zeros(6, dtype=torch.int64)
        eng.encode_dev(d_in, d_out, blocks, 6, d_len, d_pos)
        d_pack = torch.zeros(out_bytes, dtype=torch.uint8)
        d_off = torch.zeros(7, dtype=torch.int64)
        eng.gather_dev(d_out, d_pos, d_len, 6, max(b.out_cap for b in blocks[:6]), d_pack, d_off)
        for i, d in enumerate(datas):
            want = orc.rans(d, orders[i])
            assert int(d_len[i]) == len(want)
            assert d_pack[int(d_off[i]):int(d_off[i]) + len(want)].numpy().tobytes() == want
            assert int(d_off[i]) % 16 == 0
        # decode_dev
        streams = [orc.rans(d, o) for d, o in zip(datas, orders)]
        blocks, in_bytes, out_bytes = xc.Engine.plan_decode([len(s) for s in streams], [len(d) for d in datas])
        d_in = torch.zeros(in_bytes, dtype=torch.uint8)
        for i, s in enumerate(streams):
            d_in[blocks[i].in_off:blocks[i].in_off + len(s)] = torch.as_tensor(np.frombuffer(s, np.uint8).copy())
        d_out = torch.zeros(out_bytes, dtype=torch.uint8)
        eng.decode_dev(d_in, d_out, blocks, 6, d_len)
        for i, d in enumerate(datas):
            assert int(d_len[i]) == len(d)
            assert d_out[blocks[i].out_off:blocks[i].out_off + len(d)].numpy().tobytes() == d
    finally:
        eng.close()


def test_coalescer_test_bodies(hostsim, orc, monkeypatch):
    """The coalescing queue (dispatcher / completer threads, tickets) over the host build: tests/test_coalesce_gpu.py."""
    import test_coalesce_gpu as t
    from cramtools_amd import external_compression as xc
    monkeypatch.setattr(t, "N_THREADS", 6)
    monkeypatch.setattr(t, "N_WINDOW", 5)
    monkeypatch.setattr(t, "LONG", 2000)
    eng = xc.Engine(0)
    try:
        t.test_many_threads_one_container_each(eng, xc, orc)
        t.test_enqueue_window_packed(eng, xc, orc)
        t.test_bad_submissions_and_capacity(eng, xc, orc)
    finally:
        eng.close()
    monkeypatch.setenv("HIPSIM_DEVICES", "3")  # the pool's routing and ordering over three "devices"
    from cramtools_amd import sharding
    p = sharding.DevicePool()
    assert p.devices == 3
    p.close()
    t.test_device_pool_orders_results(orc)


def test_records_gpu_test_bodies(hostsim, katrec, orc, monkeypatch):
    import test_records_gpu as t
    from cramtools_amd import external_compression as xc
    from cramtools_amd import records
    monkeypatch.setattr(t, "DEV", "cpu")
    eng = xc.Engine(0)
    try:
        rk = (records, eng)
        t.test_kat2_features_matrix_restore(katrec, rk)
        t.test_kat4_window_edges(katrec, rk)
        t.test_kat3_lossy(katrec, rk)
        t.test_malformed_feature_positions(rk)
    finally:
        eng.close()


def test_host_stripe_test_bodies(hostsim, orc):
    """cramgpu_ref_upload / cramgpu_submit_reads / cramgpu_submit_restore through host arrays (tests/test_host_stripes_gpu.py)."""
    import test_host_stripes_gpu as t
    t.test_host_stripes_encode_and_restore(orc)
    t.test_host_stripes_lossy(orc)


@SLOW
def test_records_gpu_random_bodies(hostsim, orc, monkeypatch):
    import test_records_gpu as t
    from cramtools_amd import external_compression as xc
    from cramtools_amd import records
    monkeypatch.setattr(t, "DEV", "cpu")
    eng = xc.Engine(0)
    try:
        rk = (records, eng)
        t.test_random_containers_vs_oracle(rk, orc, False)
        t.test_multi_container_batch(rk, orc)
        t.test_restore_quals_many_reads(rk, orc)
    finally:
        eng.close()


@SLOW
def test_bench_step_on_host_build(hostsim, orc, monkeypatch):
    """One untimed bench step end to end (K3 + matrices + codes + rANS encode + gather, then the decode leg) on a 2-container workload."""
    import bench
    from cramtools_amd import external_compression as xc
    monkeypatch.setattr(bench, "READS_PER_CONTAINER", 300)
    eng = xc.Engine(0)
    try:
        w = bench.build_workload("cpu", 600, 20000, 150, 0, lambda reads, ref: bench.gpu_insert_bases(eng, reads, ref, 600))
        monkeypatch.setattr(torch.cuda, "Event", lambda **k: None, raising=False)
        step = bench.Step("cpu", eng, w)
        step.run(False)
        lens = step.d_len.numpy()
        pos = step.d_pos.numpy()
        off = step.d_pack_off.numpy()
        d_in = w["d_in"].numpy()
        assert (lens >= 0).all()
        for i in range(w["n_blocks"]):
            b = w["blocks"][i]
            want = orc.rans(d_in[b.in_off:b.in_off + b.in_len].tobytes(), b.order)
            assert step.d_out[int(pos[i]):int(pos[i]) + int(lens[i])].numpy().tobytes() == want, i
            assert step.d_pack[int(off[i]):int(off[i]) + int(lens[i])].numpy().tobytes() == want, i
        # the decode leg (configs[2]): rANS decode + restoreReadBases + restoreQualityScores round-trip
        dec = bench.DecodeStep("cpu", eng, w, step)
        dec.prepare()
        dec.run(False)
        assert "round-trip" in dec.verify()
    finally:
        eng.close()


def test_series_writer_test_bodies(hostsim, orc, monkeypatch):
    """tests/test_series_gpu.py on the host build of csrc/series.hip."""
    import test_series_gpu as t
    monkeypatch.setattr(t, "DEV", "cpu")
    t.test_writer_matches_oracle(orc, 0, [1])
    t.test_writer_matches_oracle(orc, 2, [300, 5, 257])
    t.test_writer_feeds_the_encoder(orc)
    t.test_reader_matches_oracle_and_inverts_writer(orc, 0, [1])
    t.test_reader_matches_oracle_and_inverts_writer(orc, 2, [300, 5, 257])
    t.test_read_names_series(orc, 0, [1], True)
    t.test_read_names_series(orc, 1, [300, 7], False)
    t.test_read_names_series(orc, 2, [260, 5, 257], True)


def test_bam_test_bodies(hostsim, orc, monkeypatch):
    """tests/test_bam_gpu.py on the host build of csrc/bam.hip."""
    import test_bam_gpu as t
    monkeypatch.setattr(t, "DEV", "cpu")
    t.test_assembly_and_encode_match_oracle(orc, 0, [1])
    t.test_assembly_and_encode_match_oracle(orc, 2, [300, 5, 257])
    t.test_decoder_flags_malformed_records(orc)
    t.test_slice_md5_matches_hashlib(orc)
    t.test_set_mate_info_matches_oracle(orc, 1, [2, 3])
    t.test_set_mate_info_matches_oracle(orc, 2, [700, 5, 256, 257])
    t.test_content_digests_match_oracle(orc, 0, [1, 0, 3], 60)
    t.test_content_digests_match_oracle(orc, 1, [700, 256], 300)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    t.test_reference_bam_record_test_inputs()
    t.test_record_boundaries_on_the_device(orc, 0, [1])
    t.test_record_boundaries_on_the_device(orc, 2, [300, 5, 257])


def test_gunzip_test_bodies(hostsim, monkeypatch):
    """tests/test_gunzip_gpu.py on the host build of csrc/inflate.hip."""
    import test_gunzip_gpu as t
    monkeypatch.setattr(t, "DEV", "cpu")
    # the corpus through the lane-per-stream kernel; the wave-per-stream kernel (a barrier of 64 host threads per match
    # here) on its own small streams
    monkeypatch.setenv("CRAMGPU_GUNZIP_LANES", "2")
    t.test_gunzip_matches_zlib("2", monkeypatch)
    t.test_bgzf_blocks()
    t.test_gunzip_failures("2", monkeypatch)
    t.test_wave_kernel_small_streams(monkeypatch)


def test_gzip_test_bodies(hostsim, monkeypatch):
    """tests/test_gzip_gpu.py on the host build of csrc/deflate.hip."""
    import test_gzip_gpu as t
    monkeypatch.setattr(t, "DEV", "cpu")
    monkeypatch.setenv("CRAMGPU_GUNZIP_LANES", "2")  # the members are inflated again by the device kernel
    # (test_members_inflate_to_the_input: the same inputs run under the sanitizers in tests/test_kernels_hostsim.py)
    t.test_bgzf_image()
    t.test_long_series_as_several_members()
    t.test_slot_too_small()

