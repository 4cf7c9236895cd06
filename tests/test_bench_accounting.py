"""bench.py's byte accounting and timing arithmetic, without a GPU: what `roofline.frac` is computed from is part of the
measurement contract, so its definition is pinned here on hand-computable cases.

The fraction counts the bytes a launch READS AND WRITES: x and y once, the stream's per-nnz bytes, row_ptr and code bytes
only of the blocks that read them, the dot operand of the launches whose operand is not their input vector — never the
format's size and never SURVEY §8d's CSR bytes for a compressed stream (round 2's headline did the former)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402


class FakeCsr:
    def __init__(self, mode, n_off, n_pair, nb, nu, dtype="f8"):
        self._fmt, self._blk, self.dtype = (mode, n_off, n_pair), (nb, nu), np.dtype(dtype)

    def stream_format(self):
        return self._fmt

    def wide_blocks(self):
        return self._blk

    def tile_plan(self):
        return getattr(self, "_tiles", (0, 0, 0))


N, NNZ = 50_000_000, 349_100_000          # BASELINE cfg 5


def test_plain_csr_bytes_are_survey_8d_minus_unread_row_ptr():
    s = bench.stream_info(FakeCsr(0, 0, 0, 1000, 0), N, NNZ, 8)
    assert s["bytes_moved_per_launch"] == bench.spmv_bytes(N, NNZ, 8) == NNZ * 12 + (N + 1) * 4 + 2 * N * 8
    s = bench.stream_info(FakeCsr(0, 0, 0, 1000, 900), N, NNZ, 8)     # 90 % of the blocks take their extents from the descriptor
    assert s["bytes_moved_per_launch"] == int(NNZ * 12 + 0.1 * (N + 1) * 4 + 2 * N * 8)
    assert s["format_bytes_per_launch"] == bench.spmv_bytes(N, NNZ, 8)


def test_compressed_streams_count_only_what_is_read():
    pair = bench.stream_info(FakeCsr(2, 7, 7, 390625, 390048), N, NNZ, 8)
    uf = 390048 / 390625
    assert pair["format_bytes_per_launch"] == NNZ * 1 + (N + 1) * 4 + 2 * N * 8                 # 1.35 GB: the FORMAT
    assert pair["bytes_moved_per_launch"] == int((1 - uf) * (NNZ + (N + 1) * 4) + 2 * N * 8)    # 0.80 GB: what a launch moves
    off = bench.stream_info(FakeCsr(1, 7, 0, 781250, 679688), N, NNZ, 8)
    uf = 679688 / 781250
    assert off["bytes_moved_per_launch"] == int(NNZ * 8 + (1 - uf) * (NNZ + (N + 1) * 4) + 2 * N * 8)
    # complex pair codes (cfg 4): one byte per entry, row_ptr, + the row-value slot: one scalar per row beside x and y
    n4, nnz4 = 500_000, 2_497_000
    cp = bench.stream_info(FakeCsr(2, 5, 5, 7813, 0, dtype="c16"), n4, nnz4, 16)
    assert cp["bytes_moved_per_launch"] == cp["format_bytes_per_launch"] == nnz4 + (n4 + 1) * 4 + 3 * n4 * 16
    assert "row" in bench.KERNEL_NAMES[cp["kernel_id"]] and "cplx" in bench.KERNEL_NAMES[cp["kernel_id"]]


def test_roofline_fraction_adds_the_dot_operand_and_keeps_the_extras_apart():
    pair = bench.stream_info(FakeCsr(2, 7, 7, 390625, 390048), N, NNZ, 8)
    t = 278e-6
    r = bench.roofline_of(pair, t, 401, N, NNZ, 8, 200)            # 200 iterations: 401 launches, K2's operand is r0
    moved = pair["bytes_moved_per_launch"] + N * 8 * 200 / 401
    assert abs(r["bytes_moved_per_launch"] - moved) < 1 and abs(r["frac"] - moved / t / 8e12) < 1e-12
    assert 0.44 < r["frac"] < 0.46 and 0.60 < r["frac_format_bytes"] < 0.61      # round 2's "0.60" was the second figure
    assert r["csr_equivalent_GBs"] > 8000 and "NOT a roofline fraction" in r["csr_equivalent_note"]
    csr = bench.stream_info(FakeCsr(0, 0, 0, 781250, 675000), N, NNZ, 8)
    r = bench.roofline_of(csr, 1010e-6, 201, N, NNZ, 8, 100)
    assert r["frac"] > r["frac_survey_8d"] * 0.99 and abs(r["frac_survey_8d"] - bench.spmv_bytes(N, NNZ, 8) / 1010e-6 / 8e12) < 1e-12
    assert "csr_equivalent_GBs" not in r
    r0 = bench.roofline_of(csr, 1010e-6, 201, N, NNZ, 8, 0)          # MINRES: the Lanczos operand is the input vector
    assert r0["bytes_moved_per_launch"] == csr["bytes_moved_per_launch"]


def test_pmc_traffic_reports_staleness_by_source_digest(tmp_path, monkeypatch):
    t, note, stale = bench.pmc_traffic("cfg5_pair")
    assert t and t > 1e9 and stale in (False, True) and "csrc digest" in note
    newest = [q for q in ("r04_pmc_summary.json", "r03_pmc_summary.json") if os.path.exists(os.path.join(ROOT, "profiles", q))][0]
    with open(os.path.join(ROOT, "profiles", newest)) as f:
        rec = json.load(f)["cfg5_pair"]
    assert stale == (rec["csrc_digest"] != bench.csrc_digest())
    monkeypatch.setattr(bench, "csrc_digest", lambda: "0" * 16)
    assert bench.pmc_traffic("cfg5_pair")[2] is True
    assert bench.pmc_traffic("no_such_key") == (None, "no PMC summary found", None)


def test_marginal_timing_arithmetic(monkeypatch):
    calls = []

    def fake_time_solve(torch, dist, solver, precond, rhs, x, steps, warmup, world, profile=True):
        calls.append((steps, warmup))
        return 0.010 + 0.0015 * steps, dict(spmv_ms_total=0.3 * (2 * steps + 1), spmv_launches=2 * steps + 1)     # 10 ms of set-up

    monkeypatch.setattr(bench, "time_solve", fake_time_solve)
    ms, dt, prof, rec = bench.time_marginal(None, None, None, None, None, None, 20, 5, 1)
    assert calls == [(20, 5), (120, 0)] and abs(ms - 1.5) < 1e-9                 # the set-up cancels
    assert abs(rec["timed_region"]["ms_per_step_with_setup"] - 2.0) < 1e-9 and rec["ms_per_step_from"].startswith("marginal")
    assert prof["spmv_launches"] == 241
    calls.clear()
    ms, dt, prof, rec = bench.time_marginal(None, None, None, None, None, None, 200, 20, 1)
    assert calls == [(200, 20)] and abs(ms - (0.010 + 0.3) / 200 * 1e3) < 1e-9 and rec["ms_per_step_from"] == "timed_region"


def test_tile_plans_name_their_kernel_and_count_no_more_bytes():
    """A handle with an LDS-window tile plan is timed on spmv_tile_kernel / spmv_tile_off_kernel: the roofline names it; the
    bytes a launch moves do not grow (a window is read once from HBM however often the CUs reuse it) — for the offset stream
    they are counted WITHOUT any code or row_ptr bytes, which errs on the low side of the fraction."""
    pair = FakeCsr(2, 7, 7, 390625, 390048); pair._tiles = (11880, 380160, 10465)
    s = bench.stream_info(pair, N, NNZ, 8)
    plain = bench.stream_info(FakeCsr(2, 7, 7, 390625, 390048), N, NNZ, 8)
    assert s["bytes_moved_per_launch"] == plain["bytes_moved_per_launch"] and s["kernel_id"] == 3 and s["lds_window_tiles"] == 11880
    assert "spmv_tile_kernel" in bench.roofline_of(s, 213e-6, 201, N, NNZ, 8, 100)["kernel"]
    off = FakeCsr(1, 7, 0, 781250, 675000); off._tiles = (11880, 380160, 10465)
    s = bench.stream_info(off, N, NNZ, 8)
    assert s["bytes_moved_per_launch"] == NNZ * 8 + 2 * N * 8 and s["kernel_id"] == 4
    assert s["bytes_moved_per_launch"] < bench.stream_info(FakeCsr(1, 7, 0, 781250, 675000), N, NNZ, 8)["bytes_moved_per_launch"]
    assert "spmv_tile_off_kernel" in bench.roofline_of(s, 740e-6, 201, N, NNZ, 8, 100)["kernel"]
