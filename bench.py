#!/usr/bin/env python3
"""Benchmark of the sprsolve Krylov hot path on MI355X — BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--workload poisson3d|poisson2d|banded|complex|bench100]

A "step" is one BiCGStab iteration (2 SpMV + the fused vector updates and dot products) on
synthetic data already resident in HBM.  The timed region is one `solve` call with
max_iter = K and tol = 0 — exactly K iterations (src/bicg_stab.rs:122: the unrolled first
iteration + K-1 loop iterations, no early exit), including the solve's set-up (one SpMV, one
axpy, two norms) — bracketed by barrier + device synchronisation; the maximum over ranks is
reported.  One JSON line is printed by rank 0.

Workload (config.workload):
  * default, every N: BASELINE cfg 5 — 7-point 3-D Poisson, 500x500x200 = 50 M rows,
    349.1 M nnz, f64, BiCGStab.  It is the configuration the metric's "1/2/4/8 MI355X"
    scaling is quoted on, it fits one GPU (5.2 GB), and — unlike the 1 M-row cfg 2, whose
    whole working set sits in the 256 MiB Infinity Cache — it is genuinely HBM-bound.
    N > 1: the same system row-partitioned in z-slabs over the ranks (strong scaling), halo
    exchange + dot all-reduce over RCCL.  `--gpus N` started WITHOUT torchrun launches itself:
    the parent (which never touches the GPU) runs `python -m torch.distributed.run` with N ranks,
    relays rank 0's JSON line and exits with the child's code.
  * N = 1 additionally measures (object "also"): the plain-CSR kernel on the same matrix, the same
    pattern with RANDOM values (what "identical random CSR inputs" exercises: no value dictionary,
    9 B/nnz offset-code stream), and BASELINE cfg 2 (1 M-row 2-D Poisson, BiCGStab + Jacobi).
  * `--workload bench100`: BASELINE cfg 1, the reference's own bench (benches/bicgstab.rs:14-37):
    CPU restatement at 4 threads and all cores, fixed iteration count; the GPU number beside it if a GPU is there.

roofline: `achieved` = the bytes the TIMED kernel reads and writes per launch / its mean launch time (HIP events on the
solver's stream inside the timed solve — around a sample of the launches, `--profile-stride`: a launch that carries events
costs ~6 us more; `roofline.launches` of `launches_in_region`; rocprofv3's per-kernel averages agree, profiles/).  "Reads and writes" is literal
(`bytes_moved_per_launch`): x and y once, the dot operand where it is not the input vector (K2's r0), the stream's
per-nnz bytes, and row_ptr / code bytes ONLY of the blocks that read them (uniform blocks of the compressed streams and
equal-length blocks of the plain stream take their extents from the descriptor).  The format's size
(`format_bytes_per_launch`) and SURVEY §8d's CSR formula (`csr_equivalent_GBs`, a speed-up figure) are named extras,
never the fraction.  BASELINE's "CSR SpMV GB/s (% HBM roofline)" is the plain-CSR kernel's line: `roofline_plain_csr`
(and the top-level `roofline` under `--stream csr`).

Timing: W untimed warm-up iterations, then EXACTLY K timed ones between barrier + device synchronisation (max over ranks):
`timed_region`.  With K < 100 that region is mostly the solve's set-up (one SpMV, one axpy, two norms, the unrolled first
iteration), so a second region of K_hi = K + 100 iterations is timed the same way and `ms_per_step` / `value` are the
MARGINAL rate (K_hi - K) / (T(K_hi) - T(K)); both regions are in the line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
STREAM_KNOB = {"auto": -1, "csr": 0, "offsets": 1, "dict": 2}
STREAM_NAMES = {0: "csr", 1: "offset-codes", 2: "pair-codes"}
KERNEL_NAMES = {0: "spmv_kernel<double> (plain CSR stream, wavefront-private LDS products, fused dot epilogue; csrc/spmv.hip)",
                1: "spmv_dict_kernel<double, PAIR=false> (one-byte column-offset codes + 8-byte values; csrc/spmv_dict.hip)",
                2: "spmv_pair2_kernel<DOT> (one-byte (offset, value) pair codes, two rows per lane, uniform blocks from a scalar "
                   "pattern; csrc/spmv_dict.hip)",
                4: "spmv_tile_off_kernel<DOT> (offset codes + 8-byte values; runs of 4096 rows of one stencil pattern: x from a window staged in "
                   "LDS + per-row-pair far loads, the values streamed through the wavefront's LDS slice; the other 64-row blocks by the "
                   "per-block walk of the same launch; csrc/spmv_dict.hip)",
                5: "spmv_chain_kernel<DOT> (pair codes; plane-streaming chains: a workgroup walks a column of 2048-row tiles plane by plane with the "
                   "x windows of three consecutive tiles in LDS — the +-plane operands come from the neighbouring tiles' windows, no far load; "
                   "the other 128-row blocks by the per-block walk of the same launch; csrc/spmv_chain.hip)",
                6: "spmv_dict_kernel<cplx, PAIR=true> (complex scalars: one-byte (offset, value) pair codes of the off-diagonal entries + one row "
                   "value (the diagonal) per row; csrc/spmv_dict.hip)",
                3: "spmv_tile_kernel<DOT> (pair codes; runs of 4096 rows of one stencil pattern multiplied from an x window staged in "
                   "LDS + per-row-pair far loads, tiles dealt to the XCDs by the far period; the other 128-row blocks by the "
                   "per-block walk of the same launch; csrc/spmv_dict.hip)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="poisson3d", choices=["poisson3d", "poisson2d", "banded", "complex", "bench100"])
    ap.add_argument("--grid", default="500x500x200", help="poisson3d grid nx x ny x nz")
    ap.add_argument("--values", default="poisson", choices=["poisson", "random"],
                    help="poisson3d values: the constant-coefficient operator, or U(-1,1) off-diagonals with a dominant diagonal")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=24.0,
                    help="budget of the CPU baseline (both thread legs together); samples keep all rows and cut iterations")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the bootstrap (gloo + SPRS_BENCH_DEVICE + SPRS_RCCL_LIB rehearse "
                         "the N>1 leg with several ranks on one GPU)")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"],
                    help="N>1 SpMV input exchange: sparse halo (default) or north_star's literal full all-gather of x")
    ap.add_argument("--stream", default="auto", choices=list(STREAM_KNOB),
                    help="SpMV stream: plain CSR (12 B/nnz), one-byte column-offset codes, offset + value codes; "
                         "auto = the most compact one the matrix qualifies for (csrc/spmv_dict.hip)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="library tuning knob (sprs_ctx_set), e.g. --set spmv_grid=2048; experiments only")
    ap.add_argument("--profile-stride", type=int, default=PROFILE_STRIDE,
                    help="poisson3d, one GPU: HIP events around one pair of consecutive SpMV launches in this many (1 = around every launch)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed (RCCL) code path even with one rank — rehearsal of the N>1 leg on a 1-GPU box")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ self-launch
def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This process has not touched the
    GPU (no torch import, no HIP call), the ranks are ordinary children (never an exec).

    No port is chosen here: the launcher's c10d rendezvous binds port 0 ITSELF (atomically) and the ranks share that store
    (torch elastic's agent store), so there is no window between "find a free port" and "bind it" for another process to
    take it (round 3's EADDRINUSE).  Should the agent still die on a socket error before any rank ran, ONE fresh child is
    started — a new rendezvous, a new port."""
    import uuid
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("MASTER_PORT", "MASTER_ADDR", "RANK", "LOCAL_RANK", "WORLD_SIZE", "TORCH_DISABLE_SHARE_RDZV_TCP_STORE"):
        env.pop(k, None)
    p = None
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--rdzv-backend", "c10d", "--rdzv-endpoint", "127.0.0.1:0", "--rdzv-id", uuid.uuid4().hex,
               "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(p.stderr)
        sock_err = any(w in p.stderr for w in ("EADDRINUSE", "Address already in use", "address already in use"))
        if p.returncode == 0 or not sock_err or "bench.py rank" in p.stderr or attempt == 1:
            break
        print("bench.py: the launcher lost a socket before any rank started; starting one fresh child", file=sys.stderr)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    if p.returncode != 0:
        sys.exit(p.returncode)
    if line is None:
        print("bench.py: the ranks exited 0 but rank 0 printed no JSON line", file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


# ------------------------------------------------------------------------------------------ helpers
def spmv_bytes(n, nnz, s):
    """nnz*(s+4) + (n+1)*4 + n*s (x) + n*s (y)."""
    return nnz * (s + 4) + (n + 1) * 4 + 2 * n * s


def stream_info(A, n, nnz, s):
    """What the SpMV of handle A streams (csrc/spmv.hip, csrc/spmv_dict.hip): the format's size per launch (x and y
    counted once, like spmv_bytes: plain CSR nnz*(s+4); offset codes nnz*(s+1); pair codes nnz*1, + row_ptr) and the
    bytes a launch actually READS AND WRITES (`bytes_moved_per_launch`, the dot operand excluded — roofline_of adds it):
    blocks that take their row extents / code pattern from the descriptor read neither row_ptr nor code bytes."""
    mode, n_off, n_pair = A.stream_format()
    per_nnz = {0: s + 4, 1: s + 1, 2: 1}[mode]
    nb, nu = A.wide_blocks()          # blocks the kernel walks / blocks that read no row_ptr (and no code bytes)
    uf = nu / nb if nb else 0.0
    code_b = 0 if mode == 0 else 1
    moved = int(nnz * (per_nnz - code_b) + (1.0 - uf) * (nnz * code_b + (n + 1) * 4) + 2 * n * s)
    tiles = A.tile_plan() if hasattr(A, "tile_plan") else (0, 0, 0)
    if tiles[0] > 0 and mode == 1:
        # offset-code tiles read neither codes nor row_ptr (seam blocks included); the 64-row blocks outside the tiles (a few
        # per cent) are counted as if none of them did either — fewer bytes, i.e. the fraction errs on the low side
        nu = nb
        moved = int(nnz * (per_nnz - code_b) + 2 * n * s)
    cpair = mode == 2 and A.dtype.kind == "c"
    if cpair:
        # complex pair codes: one byte per entry + the row-value slot (one scalar per row: the diagonal; csrc/spmv_dict.hip cpair
        # stage); the lane-per-row walk reads row_ptr and every code byte
        moved = int(nnz * 1 + (n + 1) * 4 + n * s + 2 * n * s)
    out = dict(stream=STREAM_NAMES[mode], mode=mode, distinct_offsets=n_off, distinct_pairs=n_pair,
               bytes_per_nnz=per_nnz, format_bytes_per_launch=nnz * per_nnz + (n + 1) * 4 + 2 * n * s + (n * s if cpair else 0),
               row_blocks=nb, descriptor_only_blocks=nu, bytes_moved_per_launch=moved)
    if cpair:
        out.update(kernel_id=6)
    if tiles[0] > 0:        # same bytes: a tile reads x once per window instead of once per column, all of it on chip
        out.update(kernel_id=3 if mode == 2 else 4, lds_window_tiles=tiles[0], blocks_in_tiles=tiles[1], blocks_walked_singly=tiles[2])
    chain = A.chain_plan() if hasattr(A, "chain_plan") else (0, 0, 0, 0)
    if chain[0] > 0:
        out.update(kernel_id=5, chain_tiles=chain[0], chain_segments=chain[1], chains=chain[2], blocks_in_chains=16 * chain[0],
                   blocks_walked_singly=chain[3])
    return out


def roofline_of(sinfo, t_spmv, launches, n, nnz, s=8, dot_launches=0, fused=(0, 0), fused_passes=(3, 1)):
    """The roofline object of one measured SpMV kernel: fraction of the HBM peak on the bytes it READS AND WRITES.
    dot_launches: how many of the `launches` read a dot operand that is not their input vector (n*s bytes each).
    fused = (K2 launches, K4 launches) that formed their input on the fly (csrc/krylov.hip "fused SpMV input"): such a K4 also reads v
    (1 more vector; it does not store s — K5 forms it again), such a K2 also reads p and r and writes p' (3 more).
    MINRES ("M3 deferred"): fused = (launches that multiplied by the un-normalised v_new scaled in their gathers, 0), fused_passes =
    (0, 0): the same bytes as a plain launch."""
    extra_vec = fused_passes[0] * fused[0] + fused_passes[1] * fused[1]
    moved = sinfo["bytes_moved_per_launch"] + (n * s * (dot_launches + extra_vec) / launches if launches else 0.0)
    r = dict(bound="hbm", kernel=KERNEL_NAMES[sinfo.get("kernel_id", sinfo["mode"])], achieved=moved / t_spmv / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
             frac=moved / t_spmv / 1e9 / HBM_PEAK_GBS, traffic=None, stream=sinfo["stream"], bytes_per_nnz=sinfo["bytes_per_nnz"],
             algorithmic_bytes_per_launch=moved, bytes_moved_per_launch=moved,
             bytes_note=("mean over the %d timed launches of what a launch reads and writes: x, y, the stream (%d B/nnz), row_ptr%s of the "
                         "%d of %d blocks that read them, and the dot operand of the %d launches whose operand is not their input "
                         "vector" % (launches, sinfo["bytes_per_nnz"], " and code bytes" if sinfo["mode"] else "",
                                     sinfo["row_blocks"] - sinfo["descriptor_only_blocks"], sinfo["row_blocks"], dot_launches)),
             format_bytes_per_launch=sinfo["format_bytes_per_launch"],
             frac_format_bytes=sinfo["format_bytes_per_launch"] / t_spmv / 1e9 / HBM_PEAK_GBS,
             avg_launch_us=t_spmv * 1e6, launches=launches)
    if extra_vec or (fused[0] and fused_passes == (0, 0)):
        if fused_passes == (3, 1):
            r["fused_launches"] = dict(k2_with_k1=fused[0], k4_with_k3=fused[1],
                                       note="these launches also form the vector update that produces their input (K1 / K3 of the five-launch iteration: "
                                            "3 / 1 more vector passes each, counted in the bytes above); avg_launch_us is the mean over ALL timed SpMV launches")
        else:
            r["fused_launches"] = dict(m1_on_raw_v_new=fused[0],
                                       note="these launches ran MINRES' M3 prologue and multiplied by v_new / beta_new formed in their gathers (M3 deferred: "
                                            "two launches per iteration instead of three); same bytes as a plain launch")
    if sinfo["mode"] != 0:
        r["frac_format_bytes_note"] = "the format's size / time: counts code bytes and row_ptr the kernel does not read; NOT the roofline fraction"
        r["csr_equivalent_GBs"] = spmv_bytes(n, nnz, s) / t_spmv / 1e9
        r["csr_equivalent_note"] = ("the same launch time against SURVEY §8d's CSR bytes nnz*12 + (n+1)*4 + 2*n*8: what a plain-CSR "
                                    "kernel would have to sustain to be as fast; a speed-up figure, NOT a roofline fraction "
                                    "(the stream is lossless, y bit-identical; the plain-CSR kernel's own fraction is roofline_plain_csr)")
    else:
        r["survey_8d_bytes"] = spmv_bytes(n, nnz, s)
        r["frac_survey_8d"] = spmv_bytes(n, nnz, s) / t_spmv / 1e9 / HBM_PEAK_GBS
    return r


SPMV_SOURCES = ("device.hpp", "internal.hpp", "scalar.hpp", "spmv.hip", "spmv_dict.hip", "spmv_dict_dev.hpp", "spmv_tile.hip", "spmv_tile_off.hip",
                "spmv_chain.hip", "bicg_fuse.hpp")


def csrc_digest():
    """sha256 over the sources that define the SpMV kernels and their launches: a PMC summary taken with other kernels is
    stale (there is no .git on the GPU box to ask for a commit)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sprsolve_amd", "csrc")
    for name in SPMV_SOURCES:
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode()); h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(key):
    """HBM-side bytes per SpMV launch from the committed rocprofv3 PMC summary (separate FETCH_SIZE / WRITE_SIZE passes
    of this bench command, gfx950 x2 FETCH correction) — measured off-line, so it goes stale when the kernels change:
    the summary records the digest of csrc/ it was taken with, and the note says whether that is still the code running."""
    for name in ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                ent = json.load(f).get(key, {})
            pm = ent.get("spmv_in_solve")
            if pm:
                dig = ent.get("csrc_digest")
                stale = dig != csrc_digest()
                return pm["traffic_bytes"], ("profiles/%s[%s] (commit %s, csrc digest %s; %s): %s"
                                             % (name, key, ent.get("commit"), dig,
                                                "STALE — csrc/ has changed since" if stale else "current sources", pm["note"])), stale
    return None, "no PMC summary found", None


def stream_ceiling(torch, ctx, n, reps=20):
    """SURVEY §8d's secondary denominator: what this library's own streaming kernels reach on this box.
    axpy (y += a*x: 2 reads + 1 write of n doubles, `ew_kernel<AxpyF>`) and a device-to-device copy
    (1 read + 1 write), back to back on the library's stream, wall clock around `reps` launches."""
    import ctypes as C

    from sprsolve_amd import _lib
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.rand(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    L = _lib.lib()
    px, py = C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())
    out = {}
    for name, fn, nbytes in (("axpy", lambda: L.sprs_axpy_d(ctx.h, n, 1e-3, px, py), 3 * n * 8),
                             ("copy", lambda: L.sprs_memcpy_d2d(ctx.h, py, px, n * 8), 2 * n * 8)):
        for _ in range(3):
            fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        out[name + "_GBs"] = nbytes * reps / (time.perf_counter() - t0) / 1e9
    return out


def run_fixed_iterations(solver, precond, rhs, x, steps):
    """solve(max_iter=steps, tol=0): exactly `steps` iterations, ends in InsufficientIterNum."""
    import sprsolve_amd as sa
    try:
        if precond is not None:
            solver.precond_solve(precond, rhs, x, steps, 0.0)
        else:
            solver.solve(rhs, x, steps, 0.0)
    except sa.error.InsufficientIterNum as e:
        assert e.iters == steps
        return
    raise RuntimeError("the fixed-iteration solve returned early: timing would be invalid")


def allreduce_scalar(torch, dist, value, op):
    """All-reduce one float over the ranks (CUDA tensor under nccl, CPU tensor under gloo)."""
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=op)
    return float(t.item())


# The timed region of the headline measures its SpMV launches live, with HIP events on the solver's stream — on a SAMPLE of them:
# one pair of consecutive launches (a K2 and a K4) in PROFILE_STRIDE pairs.  A launch that carries events costs ~6 us more
# (profiles/r04_tuning.md §9): all of them = 11-13 us per cfg-5 iteration (1 %), a fifth of a cfg-3 / cfg-4 iteration.
PROFILE_STRIDE = 4


def time_solve(torch, dist, solver, precond, rhs, x, steps, warmup, world, profile=True):
    """W untimed warm-up iterations, then exactly K timed ones; returns seconds (max over ranks).
    profile: True / 1 = bracket every SpMV launch with HIP events on the solver's stream; k >= 2 = one pair of consecutive
    launches in k; False = none (the events cost ~6 us per launch: the small workloads time their value without them and
    measure the SpMV in a separate pass)."""
    x.zero_()
    if warmup > 0:
        run_fixed_iterations(solver, precond, rhs, x, warmup)
    x.zero_()
    solver.set_profile(int(profile))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run_fixed_iterations(solver, precond, rhs, x, steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = allreduce_scalar(torch, dist, dt, dist.ReduceOp.MAX)
    prof = solver.profile()
    solver.set_profile(False)
    return dt, prof


def time_marginal(torch, dist, solver, precond, rhs, x, steps, warmup, world, profile=True):
    """The contract's region (W warm-up, exactly K timed iterations) and, when K < 100, a second region of K + 100
    iterations: ms per step from the marginal rate, so that the solve's set-up (an SpMV, an axpy, two blocking norms, the
    unrolled first iteration) does not pass for iteration time.  Returns (ms_per_step, seconds of the K-step region,
    SpMV profile of the longest region, timing record for the JSON line)."""
    dt, prof = time_solve(torch, dist, solver, precond, rhs, x, steps, warmup, world, profile)
    rec = dict(timed_region=dict(steps=steps, seconds=dt, ms_per_step_with_setup=dt / steps * 1e3), ms_per_step_from="timed_region")
    ms = dt / steps * 1e3
    if steps < 100:
        k_hi = steps + 100
        dt_hi, prof_hi = time_solve(torch, dist, solver, precond, rhs, x, k_hi, 0, world, profile)
        rec["second_region"] = dict(steps=k_hi, seconds=dt_hi, ms_per_step_with_setup=dt_hi / k_hi * 1e3)
        if dt_hi > dt:
            ms = (dt_hi - dt) / (k_hi - steps) * 1e3
            rec["ms_per_step_from"] = "marginal: (T(%d) - T(%d)) / %d iterations" % (k_hi, steps, k_hi - steps)
        else:
            ms = dt_hi / k_hi * 1e3
            rec["ms_per_step_from"] = "second_region (the marginal difference was not positive)"
        prof = prof_hi
    return ms, dt, prof, rec


_CREATE_WARM = {}


def warm_create():
    """Once per process, before the first timed creation: create (and drop) a handle of a 300 k-row stencil with the tile
    plan forced, so that the one-time cost of loading the library's code objects and of the first launch of every creation
    kernel is not billed to the first matrix (it is 20-30 ms, as much as the whole 50 M-row creation; reported as
    `create_first_in_process_ms`)."""
    if _CREATE_WARM:
        return _CREATE_WARM["ms"]
    import sprsolve_amd as sa
    from sprsolve_amd import gen
    ctx = sa.default_ctx(0)
    ip, ix, d, _ = gen.poisson3d(160, 128, 16)
    n = ip.size - 1
    t0 = time.perf_counter()
    keep = ctx.get("spmv_tile")                            # (a --set spmv_tile=... of the caller stays in force afterwards)
    for vals in (d, d * (1.0 + 1e-3 * (ix % 7))):         # pair codes, then offset codes + values
        ctx.set("spmv_tile", 1)
        A = sa.HipCsr.new((n, n), ip, ix, vals, ctx=ctx)
        ctx.set("spmv_tile", keep)
        del A
    ctx.sync()
    _CREATE_WARM["ms"] = (time.perf_counter() - t0) * 1e3
    return _CREATE_WARM["ms"]


def timed_create(torch, make):
    """Handle creation (row blocks, dictionary collection + encoding, uniform / seam marking, tile plan, schedules): wall time
    of the blocking call with the device idle before and after, code objects already loaded (warm_create)."""
    warm_create()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    A = make()
    torch.cuda.synchronize()
    return A, (time.perf_counter() - t0) * 1e3


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    """Hardware threads this process may really use: the affinity mask, capped by the cgroup CPU quota (a container
    that sees 256 CPUs but is throttled to 16 runs SLOWER with 256 OpenMP threads than with 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = max(1, min(n, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = max(1, min(n, int(q / float(f.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(fn, ip, ix, dv, rhs, diag, budget_s, what, legs=None):
    """the reference's CPU path restated in C (oracle/: rayon-style row-parallel SpMV + SERIAL BLAS-1,
    usize indices), timed ON THE CONFIG ITSELF (all rows), at 4 threads (the reference's own bench setting,
    benches/bicgstab.rs:7-10) and at all host cores this process may use; x reset before every sample; median of >= 5
    samples per leg.  A sample is one solve with max_iter = k and tol = 0 (exactly k iterations + the solve's set-up);
    `value` is the MARGINAL iteration rate (k_hi - k_lo) / (median T(k_hi) - median T(k_lo)), so the set-up cancels.
    The budget cuts iterations per sample, never rows and never the number of samples."""
    import numpy as np

    from oracle import oracle as orc
    ip64 = np.ascontiguousarray(ip, dtype=np.int64); ix64 = np.ascontiguousarray(ix, dtype=np.int64)
    ncores = host_threads()
    legs = legs or [4, ncores]
    out = dict(unit="iterations/s", kind="port", cpu_model=cpu_model(), host_threads_available=ncores, samples=5)
    x0 = np.zeros_like(rhs)

    def solve(k):
        t0 = time.perf_counter()
        r = fn(ip64, ix64, dv, rhs, x0, k, 0.0, precond_diag=diag, parallel=True)   # x0 is copied inside: x reset per sample
        dt = time.perf_counter() - t0
        assert r.status == orc.INSUFFICIENT_ITER and r.its == k
        return dt
    for li, th in enumerate(legs):
        orc.set_threads(th)
        t1 = solve(1)                               # calibration sample (also warms the page cache of the arrays)
        per_leg = budget_s / len(legs)
        # 5 samples at k_lo and 5 at k_hi must fit the leg's budget: T(k) ~ t1 * (0.6 + 0.4 k) is only used to choose k
        k_lo = 1
        k_hi = int(max(2, min(50, (per_leg / 5.0 / max(t1, 1e-9) - 1.2) / 0.8)))
        lo = sorted(solve(k_lo) for _ in range(5))
        hi = sorted(solve(k_hi) for _ in range(5))
        marginal = (k_hi - k_lo) / max(hi[2] - lo[2], 1e-12)
        out["threads%d" % th if li == 0 else "all_cores"] = dict(
            threads=th, value=marginal, k_lo=k_lo, k_hi=k_hi, median_s_k_lo=lo[2], median_s_k_hi=hi[2],
            min_s_k_hi=hi[0], max_s_k_hi=hi[4], with_setup_it_s=k_hi / hi[2])
    best = max(out[k]["value"] for k in out if isinstance(out[k], dict))
    out["value"] = best
    out["cores"] = [out[k]["threads"] for k in out if isinstance(out[k], dict) and out[k]["value"] == best][0]
    out["sample"] = ("CPU restatement of the reference path (oracle/: row-parallel SpMV + serial BLAS-1, usize indices) on %s, ALL rows; "
                     "per thread leg 5 solves of %d and 5 of k_hi iterations (tol = 0, x reset), medians, marginal iterations/s; "
                     "CPU: %s, %d hardware threads usable; baseline, not target" % (what, 1, out["cpu_model"], ncores))
    return out


# ------------------------------------------------------------------------------------------ cfg 1
def bench100(args):
    """BASELINE cfg 1 = the reference's own bench (benches/bicgstab.rs:14-37): grid_laplacian(100,100) with Dirichlet
    rows, rhs = i+j on the border, x0 = 0, BiCGStab.  The reference times solve-to-1e-16 with x never reset
    (SURVEY §6 lists why that is not a measurement); here: fixed iteration count, x reset, 4 threads and all cores."""
    import numpy as np

    from oracle import oracle as orc
    from sprsolve_amd import gen
    R = 100
    ip, ix, dv = gen.grid_laplacian_dirichlet(R, R)
    rhs = gen.dirichlet_rhs(R, R)
    n, nnz = R * R, int(ip[-1])
    ip64 = ip.astype(np.int64); ix64 = ix.astype(np.int64)
    K = args.steps
    res = {}
    for name, th in (("threads4", 4), ("all_cores", host_threads())):
        orc.set_threads(th)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            r = orc.bicgstab(ip64, ix64, dv, rhs, np.zeros(n), K, 0.0, parallel=True)
            ts.append(time.perf_counter() - t0)
            assert r.status == orc.INSUFFICIENT_ITER
        ts.sort()
        res[name] = dict(threads=th, value=K / ts[3], median_s=ts[3], min_s=ts[0], max_s=ts[6], samples=7)
    # correctness of the restated path on the reference's own settings (tol 1e-16, max 1500): exact solution i+j
    r = orc.bicgstab(ip64, ix64, dv, rhs, np.zeros(n), 1500, 1e-16, parallel=True)
    g = np.arange(n)
    err = float(np.abs(r.x - (g // R + g % R)).max())
    out = dict(metric="BiCGStab iterations/s (f64, cfg 1: 100x100 Dirichlet grid Laplacian, benches/bicgstab.rs) — CPU restatement",
               value=res["threads4"]["value"], unit="iterations/s", n_gpus=0, steps=K, warmup=0,
               ms_per_step=1e3 / res["threads4"]["value"], higher_is_better=True, scaling="strong", vs_baseline=None,
               dtype="f64", data="synthetic",
               config=dict(workload="cfg1: grid_laplacian(100,100) + Dirichlet rhs, n=%d, nnz=%d, BiCGStab tol=0 fixed %d iterations, "
                                    "x reset per sample, median of 7" % (n, nnz, K), rows=n, nnz=nnz, index_type="usize"),
               cpu_baseline=dict(res, unit="iterations/s", kind="port", cpu_model=cpu_model(), value=res["threads4"]["value"], cores=4,
                                 sample="the whole of cfg 1; `value` is the 4-thread leg (the reference's rayon pool size, benches/bicgstab.rs:7-10)"),
               converge_check=dict(tol=1e-16, max_iter=1500, status=int(r.status), iters=int(r.its), rel_res=float(r.res),
                                   max_abs_err_vs_exact=err),
               roofline=None)
    # cfg 1 as BASELINE.json words it ("1e4-row random tridiagonal f64 CSR"; SURVEY §8d cfg 1 (ii)): off-diagonals U(-1,1),
    # diagonal 2 + |l| + |u|, rhs U(-1,1), x0 = 0, tol 1e-10 — CPU restatement, 4 threads, same sampling
    tip, tix, tdv, trhs = gen.random_tridiagonal(10000)
    t64, tx64 = tip.astype(np.int64), tix.astype(np.int64)
    orc.set_threads(4)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        rt = orc.bicgstab(t64, tx64, tdv, trhs, np.zeros(10000), K, 0.0, parallel=True)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    rc = orc.bicgstab(t64, tx64, tdv, trhs, np.zeros(10000), 1500, 1e-10, parallel=True)
    true_res = float(np.linalg.norm(orc.spmv(tip, tix, tdv, rc.x) - trhs) / np.linalg.norm(trhs))
    out["also"] = dict(cfg1_random_tridiagonal_1e4=dict(
        workload="random tridiagonal n=1e4, nnz=%d, BiCGStab tol=0 fixed %d iterations, 4 threads, median of 7" % (int(tip[-1]), K),
        value=K / ts[3], unit="iterations/s", median_s=ts[3], threads=4,
        converge_check=dict(tol=1e-10, status=int(rc.status), iters=int(rc.its), rel_res=float(rc.res), true_rel_res=true_res)))
    try:
        import torch
        if torch.cuda.is_available():
            import sprsolve_amd as sa
            ctx = sa.default_ctx(0)
            A = sa.HipCsr.new((n, n), ip, ix, dv, ctx=ctx)
            s = sa.BiCGStab.new(A, n)
            drhs = torch.from_numpy(rhs).cuda(); x = torch.zeros(n, dtype=torch.float64, device="cuda")
            dt, _ = time_solve(torch, None, s, None, drhs, x, K, min(args.warmup, K), 1, profile=False)
            out["gpu_same_config"] = dict(value=K / dt, unit="iterations/s", note="libsprsolve_hip on one MI355X, same matrix, same K; "
                                          "78 KB of matrix: pure launch latency (5 launches per iteration)")
    except Exception as e:     # the CPU line stands on its own
        out["gpu_same_config"] = dict(error=str(e))
    emit(json.dumps(out))


# ------------------------------------------------------------------------------------------ main
_JSON_OUT = None


def protect_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner on
    file descriptor 1 when its first communicator is created): from here on descriptor 1 points at stderr and the JSON
    line goes to a private duplicate of the original stdout."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _JSON_OUT if _JSON_OUT is not None else sys.stdout
    out.write(line + "\n")
    out.flush()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    protect_stdout()
    if args.workload == "bench100":
        return bench100(args)
    import torch   # first: libsprsolve_hip.so then shares torch's HIP runtime (same soname)
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        print("bench.py rank %d of %d started" % (rank, world), file=sys.stderr, flush=True)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    local_rank = int(os.environ.get("SPRS_BENCH_DEVICE", local_rank))   # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    elif args.force_dist:
        # one rank through the real RCCL: an in-process store — no port at all
        dist.init_process_group(backend="nccl", store=dist.HashStore(), rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank))

    import numpy as np

    import sprsolve_amd as sa
    from sprsolve_amd import gen_torch
    ctx = sa.default_ctx(local_rank)
    ctx.set("spmv_dict", STREAM_KNOB[args.stream])
    for kv in args.set:
        k, v = kv.split("=")
        ctx.set(k, int(v))
    dev = torch.device("cuda", local_rank)
    out = None
    also = {}

    def bench_poisson2d(steps, warmup):
        R = 1000
        ip, ix, dv, rhs, diag = gen_torch.grid_laplacian_dirichlet(R, R, device=dev)
        n, nnz = R * R, int(ip[-1].item())
        A, create_ms = timed_create(torch, lambda: sa.HipCsr.from_device((n, n), nnz, ip, ix, dv, adopt=True, ctx=ctx))
        P = sa.DiagPrecond.new(diag.cpu().numpy(), ctx=ctx)
        s = sa.BiCGStab.new(A, n)
        x = torch.zeros(n, dtype=torch.float64, device=dev)
        ms, dt, _, trec = time_marginal(torch, dist, s, P, rhs, x, steps, warmup, 1, profile=False)
        _, prof = time_solve(torch, dist, s, P, rhs, x, max(steps, 100), 0, 1, profile=True)    # separate pass: per-launch SpMV time
        # correctness of the same objects: converge to the known solution i+j
        x.zero_()
        its, res = s.precond_solve(P, rhs, x, 20000, 1e-8)
        g = torch.arange(n, device=dev)
        err = float((x - (g // R + g % R).to(torch.float64)).abs().max().item())
        # PCIe-inclusive rate of the literal reference signature (host slices in, host slice out): never `value`
        rhs_h = rhs.cpu().numpy(); x_h = np.zeros(n)
        t0 = time.perf_counter()
        try:
            s.precond_solve(P, rhs_h, x_h, steps, 0.0)
        except sa.error.InsufficientIterNum:
            pass
        pcie_dt = time.perf_counter() - t0
        bs = spmv_bytes(n, nnz, 8)
        sinfo = stream_info(A, n, nnz, 8)
        # stand-alone SpMV timing (back-to-back launches, HIP events on the library's stream)
        y = torch.empty_like(x)
        ms_alone = A.time_mul_vec(rhs, y, reps=200)
        t_spmv = prof["spmv_ms_total"] / max(prof["spmv_launches"], 1) * 1e-3
        return dict(workload="cfg2: 1000x1000 2-D 5-point Poisson (Dirichlet rows), n=1e6, nnz=4984016, BiCGStab + Jacobi",
                    value=1e3 / ms, ms_per_step=ms, timing=trec, create_ms=create_ms, n=n, nnz=nnz, spmv_stream=sinfo["stream"],
                    pcie_inclusive_it_s=steps / pcie_dt,
                    spmv_us_in_solve=t_spmv * 1e6, spmv_us_back_to_back=ms_alone * 1e3,
                    spmv_csr_equivalent_GBs_in_solve=bs / t_spmv / 1e9, spmv_csr_equivalent_GBs_back_to_back=bs / (ms_alone * 1e-3) / 1e9,
                    spmv_bytes=bs, iter_bytes_reference_oplist=2 * bs + 26 * n * 8 + 2 * n * 24,
                    effective_GBs=(2 * bs + 26 * n * 8 + 2 * n * 24) / (ms * 1e-3) / 1e9,
                    converge_check=dict(tol=1e-8, iters=its, rel_res=res, max_abs_err_vs_exact=err),
                    note="working set (~140 MB) fits the 256 MiB Infinity Cache: GB/s here is not an HBM figure"), t_spmv, bs, sinfo, prof

    if args.workload == "poisson3d":
        nx, ny, nz = (int(v) for v in args.grid.lower().split("x"))
        cpu_arrays = None
        if world == 1 and not args.force_dist:
            ip, ix, dv, rhs = gen_torch.poisson3d(nx, ny, nz, device=dev, values=args.values)
            n = nx * ny * nz
            nnz = int(ip[-1].item())
            if rank == 0 and not args.no_cpu_baseline:
                # host copy for the CPU leg, taken before the arrays are adopted (the baseline runs last)
                cpu_arrays = (ip.cpu().numpy(), ix.cpu().numpy(), dv.cpu().numpy(), rhs.cpu().numpy())
            A, create_ms = timed_create(torch, lambda: sa.HipCsr.from_device((n, n), nnz, ip, ix, dv, adopt=True, ctx=ctx))
            s = sa.BiCGStab.new(A, n)
            x = torch.zeros(n, dtype=torch.float64, device=dev)
            ms_step, dt, prof, trec = time_marginal(torch, dist, s, None, rhs, x, args.steps, args.warmup, 1, profile=max(1, args.profile_stride))
            t_spmv = prof["spmv_ms_total"] / max(prof["spmv_launches"], 1) * 1e-3
            dot_l = (prof["spmv_launches"] - 1) // 2          # K2 reads r0 beside its input; K4's operand is its input; the set-up SpMV has none
            bs = spmv_bytes(n, nnz, 8)
            # correctness on the very same objects: the exact solution is all ones
            x.zero_()
            its, res = s.solve(rhs, x, 5000, 1e-8)
            err = float((x - 1.0).abs().max().item())
            check = dict(tol=1e-8, iters=its, rel_res=res, max_abs_err_vs_exact=err)
            n_glob, nnz_glob = n, nnz
            sinfo = stream_info(A, n, nnz, 8)
            roof_csr = None
            if not args.no_also:
                # the same solve on the plain CSR stream (12 B/nnz): the kernel SURVEY §8d's formula describes;
                # identical iterates (the streams are bit-identical), so only the time differs
                k_csr = max(args.steps // 2, 10)
                if sinfo["mode"] != 0:
                    ctx.set("spmv_dict", 0)
                    ms_csr, t_csr, p_csr, trec_csr = time_marginal(torch, dist, s, None, rhs, x, k_csr, min(args.warmup, 5), 1, profile=max(1, args.profile_stride))
                    csr_info = stream_info(A, n, nnz, 8)         # under the knob: the plain stream's blocks
                    ctx.set("spmv_dict", STREAM_KNOB[args.stream])
                else:
                    ms_csr, t_csr, p_csr, trec_csr, k_csr, csr_info = ms_step, dt, prof, trec, args.steps, sinfo
                tl = p_csr["spmv_ms_total"] / max(p_csr["spmv_launches"], 1) * 1e-3
                roof_csr = roofline_of(csr_info, tl, p_csr["spmv_launches"], n, nnz, 8, p_csr.get("timed_dot_other", (p_csr["spmv_launches"] - 1) // 2))
                roof_csr["traffic"], roof_csr["traffic_note"], roof_csr["traffic_stale"] = pmc_traffic("cfg5_csr")
                if roof_csr["traffic"]:
                    roof_csr["traffic_over_bytes_moved"] = roof_csr["traffic"] / roof_csr["bytes_moved_per_launch"]
                also["cfg5_plain_csr_stream"] = dict(
                    value=1e3 / ms_csr, unit="iterations/s", ms_per_step=ms_csr, steps=k_csr, timing=trec_csr,
                    spmv_us_in_solve=tl * 1e6, spmv_GBs=roof_csr["achieved"], spmv_frac_of_hbm_peak=roof_csr["frac"],
                    spmv_frac_survey_8d=roof_csr["frac_survey_8d"],
                    note="spmv_kernel<double> on (col_idx, val); same matrix, same vectors, same iterates; the fraction is on the bytes "
                         "the launches read and write (roofline_plain_csr.bytes_note), SURVEY §8d's formula beside it")
            if not args.no_also and args.values == "poisson":
                # variable coefficients on the same pattern: no value dictionary => one-byte offset codes + values, 9 B/nnz
                del s, A
                x = None
                torch.cuda.empty_cache()
                ipr, ixr, dvr, rhsr = gen_torch.poisson3d(nx, ny, nz, device=dev, values="random")
                Ar, create_ms_r = timed_create(torch, lambda: sa.HipCsr.from_device((n, n), nnz, ipr, ixr, dvr, adopt=True, ctx=ctx))
                sr = sa.BiCGStab.new(Ar, n)
                xr = torch.zeros(n, dtype=torch.float64, device=dev)
                k_r = max(args.steps // 2, 10)
                ms_r, t_r, p_r, trec_r = time_marginal(torch, dist, sr, None, rhsr, xr, k_r, min(args.warmup, 5), 1, profile=max(1, args.profile_stride))
                tlr = p_r["spmv_ms_total"] / max(p_r["spmv_launches"], 1) * 1e-3
                sinfo_r = stream_info(Ar, n, nnz, 8)
                xr.zero_()
                its_r, res_r = sr.solve(rhsr, xr, 5000, 1e-8)
                err_r = float((xr - 1.0).abs().max().item())
                rr = roofline_of(sinfo_r, tlr, p_r["spmv_launches"], n, nnz, 8, p_r.get("timed_dot_other", (p_r["spmv_launches"] - 1) // 2))
                if (nx, ny, nz) == (500, 500, 200) and sinfo_r["mode"] == 1:
                    rr["traffic"], rr["traffic_note"], rr["traffic_stale"] = pmc_traffic("cfg5_random")
                also["cfg5_random_values"] = dict(
                    workload="the cfg-5 pattern (500x500x200 7-point) with off-diagonals U(-1,1) and diagonal 1 + sum|row off-diagonals| "
                             "(splitmix64 by nnz position), rhs = A*1, BiCGStab tol=0 fixed %d iterations" % k_r,
                    value=1e3 / ms_r, unit="iterations/s", ms_per_step=ms_r, steps=k_r, timing=trec_r, create_ms=create_ms_r,
                    spmv_stream=sinfo_r, roofline=rr,
                    converge_check=dict(tol=1e-8, iters=its_r, rel_res=res_r, max_abs_err_vs_exact=err_r))
                del sr, Ar, xr, ipr, ixr, dvr, rhsr
                torch.cuda.empty_cache()
        else:
            from sprsolve_amd import dist as sdist
            res_ = sdist.bench_poisson3d(torch, dist, ctx, rank, world, nx, ny, nz, args.steps, args.warmup, time_marginal,
                                         exchange=args.exchange)
            dt, prof, t_spmv, bs, check, n_glob, nnz_glob, sinfo = res_[:8]
            dist_info = res_[8] if len(res_) > 8 else {}
            ms_step, trec, create_ms = dist_info.pop("ms_per_step"), dist_info.pop("timing"), dist_info.pop("create_ms")
            dot_l = (prof["spmv_launches"] - 1) // 2
            roof_csr = None
        it_bytes = 2 * spmv_bytes(n_glob, nnz_glob, 8) + 26 * n_glob * 8
        n_rank = sinfo.get("rows", n_glob)
        nnz_rank = sinfo.get("nnz", nnz_glob)
        # what the TIMED launches were (all of them, or the sample — PROFILE_STRIDE): counted by the library
        roof = roofline_of(sinfo, t_spmv, prof["spmv_launches"], n_rank, nnz_rank, 8, prof.get("timed_dot_other", dot_l),
                           (prof.get("timed_fused_k2", 0), prof.get("timed_fused_k4", 0)))
        roof["launches_in_region"] = prof.get("steps", prof["spmv_launches"])
        if world == 1 and (nx, ny, nz) == (500, 500, 200) and not args.force_dist:
            key = {0: "cfg5_csr", 1: "cfg5_random", 2: "cfg5_pair"}[sinfo["mode"]]
            roof["traffic"], roof["traffic_note"], roof["traffic_stale"] = pmc_traffic(key)
        roof["note"] = ("per rank; `achieved` / `frac` = the bytes a launch reads and writes (bytes_note) / mean launch time (HIP events on the "
                        "solver's stream, inside the timed solve: `launches` of the region's `launches_in_region` SpMV launches carry them)"
                        + ("; this kernel is bound by the CUs' vector-memory path and gather latency, not by HBM (DESIGN.md §3): its fraction says "
                           "how little of the HBM bandwidth it needs, BASELINE's CSR figure is roofline_plain_csr" if sinfo["mode"] == 2 and "kernel_id" not in sinfo else "")
                        + ("; the tile kernel stages each near x window once per 4096 rows (DESIGN.md §3): what it moves crosses the fabric at the rate a "
                           "pure read stream reaches on this chip (5.5 TB/s); BASELINE's CSR figure is roofline_plain_csr" if sinfo.get("kernel_id") == 3 else "")
                        + ("; the chain kernel keeps three consecutive planes' x windows of a 2048-row column in LDS (DESIGN.md §3): x crosses the vector L1 "
                           "1.5 times and no far load exists; BASELINE's CSR figure is roofline_plain_csr" if sinfo.get("kernel_id") == 5 else "")
                        + ("; N>1: the bracketed time includes the wait for the halo exchange of the boundary rows" if world > 1 else ""))
        out = dict(metric="BiCGStab iterations/s (f64, 50M-row 7-point 3-D Poisson) + CSR SpMV GB/s vs HBM roofline",
                   value=1e3 / ms_step, unit="iterations/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=ms_step, higher_is_better=True, scaling="strong", vs_baseline=None,
                   dtype="f64", data="synthetic", timing=trec, create_ms=create_ms,
                   create_first_in_process_ms=_CREATE_WARM.get("ms"),
                   config=dict(workload="cfg5: %dx%dx%d 7-point 3-D Poisson%s, n=%d, nnz=%d, BiCGStab (no preconditioner), "
                                        "tol=0 fixed %d iterations" % (nx, ny, nz, " (random values)" if args.values == "random" else "",
                                                                       n_glob, nnz_glob, args.steps),
                               rows=n_glob, nnz=nnz_glob, index_type="i32", partition="z-slabs x%d" % world,
                               spmv_stream=sinfo,
                               bytes_per_iteration_reference_oplist=it_bytes),
                   effective_GBs_reference_oplist=it_bytes / (ms_step * 1e-3) / 1e9,
                   roofline=roof, converge_check=check)
        if roof_csr is not None:
            out["roofline_plain_csr"] = roof_csr
        if world > 1 or args.force_dist:
            out.update(dist_info)
        if world == 1 and rank == 0:
            x = None                      # release the solution vector before the microbench allocates its own
            sc = stream_ceiling(torch, ctx, n_glob if not args.force_dist else min(n_glob, 50_000_000))
            roof["measured_stream_ceiling"] = dict(sc, note="this library's axpy (2R+1W) and a d2d copy (1R+1W) on n doubles, "
                                                            "back to back — SURVEY §8d's secondary denominator")
            best = max(sc["axpy_GBs"], sc["copy_GBs"])
            roof["frac_of_measured_stream"] = roof["achieved"] / best
            if roof_csr is not None:
                roof_csr["frac_of_measured_stream"] = roof_csr["achieved"] / best
        if world == 1 and not args.no_also and not args.force_dist:
            r2, _, _, _, _ = bench_poisson2d(500, 50)
            also["cfg2_poisson2d_1M_bicgstab_jacobi"] = r2
        if cpu_arrays is not None:
            from oracle import oracle as orc
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(orc.bicgstab, *cpu_arrays, None, args.cpu_seconds,
                                               "the full cfg-5 system (%d rows, %d nnz)" % (n_glob, nnz_glob))
    elif args.workload == "poisson2d":
        if world != 1:
            raise SystemExit("poisson2d is a single-GPU workload (cfg 2)")
        r2, t_spmv, bs, sinfo, prof = bench_poisson2d(args.steps, args.warmup)
        roof = roofline_of(sinfo, t_spmv, prof["spmv_launches"], 10**6, r2["nnz"], 8, prof["spmv_launches"] - 1)   # Jacobi: both operands differ from the inputs
        roof["note"] = "cache-resident working set (fits the 256 MiB Infinity Cache): not an HBM figure; see detail.note"
        out = dict(metric="BiCGStab iterations/s (f64, 1M-row 2-D Poisson + Jacobi) + CSR SpMV GB/s",
                   value=r2["value"], unit="iterations/s", n_gpus=1, steps=args.steps, warmup=args.warmup,
                   ms_per_step=r2["ms_per_step"], higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f64",
                   data="synthetic", config=dict(workload=r2["workload"]), timing=r2["timing"], create_ms=r2["create_ms"], roofline=roof, detail=r2)
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            from sprsolve_amd import gen
            ip, ix, dv = gen.grid_laplacian_dirichlet(1000, 1000)
            out["cpu_baseline"] = cpu_baseline(orc.bicgstab, ip, ix, dv, gen.dirichlet_rhs(1000, 1000),
                                               np.where(np.diff(ip) == 1, 1.0, -4.0), args.cpu_seconds, "the full cfg-2 system (Jacobi)")
    else:
        if world != 1:
            raise SystemExit("%s is a single-GPU workload" % args.workload)
        from oracle import oracle as orc
        from sprsolve_amd import gen
        if args.workload == "banded":
            n = 10**6
            ip, ix, dv, rhs = gen.symmetric_banded(n)
            solver_cls, label, sbytes, ofn = sa.MinRes, "cfg3: symmetric banded (hbw 4), n=1e6, nnz=8999980, MINRES", 8, orc.minres
        else:
            ip, ix, dv, rhs, _ = gen.complex_symmetric_grid(500, 1000)
            n = 500000
            solver_cls, label, sbytes, ofn = sa.CSMinRes, "cfg4: complex-symmetric 500x1000 grid, n=5e5, nnz=2497000, CSMINRES", 16, orc.csminres
        A, create_ms = timed_create(torch, lambda: sa.HipCsr.new((n, n), ip, ix, dv, ctx=ctx))    # includes the H2D copy of the arrays
        s = solver_cls.new(A, n)
        tdt = torch.float64 if sbytes == 8 else torch.complex128
        drhs = torch.from_numpy(rhs).to(dev)
        x = torch.zeros(n, dtype=tdt, device=dev)
        # as cfg 2: the value from a region WITHOUT the per-launch events (they cost ~6 us per SpMV launch — a fifth of these 30-45 us
        # iterations; negligible only against cfg 5's), the SpMV's own time from a separate, event-bracketed pass of the same solve
        ms_step, dt, _, trec = time_marginal(torch, dist, s, None, drhs, x, args.steps, args.warmup, 1, profile=False)
        dt_prof, prof = time_solve(torch, dist, s, None, drhs, x, max(args.steps, 100), 0, 1, profile=True)
        trec["profiled_pass"] = dict(steps=max(args.steps, 100), ms_per_step_with_events=dt_prof / max(args.steps, 100) * 1e3,
                                     note="the pass `roofline` is measured in: HIP events around every SpMV launch")
        t_spmv = prof["spmv_ms_total"] / max(prof["spmv_launches"], 1) * 1e-3
        sinfo = stream_info(A, n, int(ip[-1]), sbytes)
        roof = roofline_of(sinfo, t_spmv, prof["spmv_launches"], n, int(ip[-1]), sbytes, 0,    # the Lanczos dot operand IS the input vector
                           (prof.get("timed_fused_k2", 0), 0), (0, 0))
        roof["kernel"] = roof["kernel"].replace("double", "double" if sbytes == 8 else "cplx")
        roof["note"] = "cache-resident working set (fits the 256 MiB Infinity Cache): the fraction is against the HBM peak all the same"
        out = dict(metric="%s iterations/s + CSR SpMV GB/s" % solver_cls.__name__, value=1e3 / ms_step, unit="iterations/s",
                   n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=ms_step, higher_is_better=True,
                   scaling="strong", vs_baseline=None, dtype="f64" if sbytes == 8 else "c64", data="synthetic", timing=trec,
                   create_ms=create_ms, config=dict(workload=label, spmv_stream=sinfo), roofline=roof)
        if not args.no_cpu_baseline:
            if args.workload == "banded":
                out["cpu_baseline"] = cpu_baseline(ofn, ip, ix, dv, rhs, None, args.cpu_seconds, "the full cfg-3 system")
            else:
                out["cpu_baseline"] = cpu_baseline(lambda a, b, c, d, e, k, tol, precond_diag=None, parallel=True: ofn(a, b, c, d, e, k, tol, parallel=parallel),
                                                   ip, ix, dv, rhs, None, args.cpu_seconds, "the full cfg-4 system")
    if also:
        out["also"] = also
    if rank == 0:
        emit(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
