#!/usr/bin/env python3
"""Benchmark of the sprsolve Krylov hot path on MI355X — BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--workload poisson3d|poisson2d|banded|complex]

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
    whole working set sits in the 256 MiB Infinity Cache — it is genuinely HBM-bound, so the
    roofline fraction is an honest HBM number.  N > 1: the same system row-partitioned in
    z-slabs over the ranks (strong scaling), halo exchange + dot all-reduce over RCCL.
  * N = 1 additionally measures BASELINE cfg 2 (1 M-row 2-D Poisson, BiCGStab + Jacobi) and
    reports it in the "also" object; `--workload poisson2d` makes it the headline instead.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="poisson3d", choices=["poisson3d", "poisson2d", "banded", "complex"])
    ap.add_argument("--grid", default="500x500x200", help="poisson3d grid nx x ny x nz")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the bootstrap (gloo + SPRS_BENCH_DEVICE + SPRS_RCCL_LIB rehearse "
                         "the N>1 leg with several ranks on one GPU)")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"],
                    help="N>1 SpMV input exchange: sparse halo (default) or north_star's literal full all-gather of x")
    ap.add_argument("--stream", default="auto", choices=["auto", "csr", "offsets", "dict"],
                    help="SpMV stream: plain CSR (12 B/nnz), one-byte column-offset codes, offset + value codes; "
                         "auto = the most compact one the matrix qualifies for (csrc/spmv_dict.hip)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="library tuning knob (sprs_ctx_set), e.g. --set spmv_grid=2048; experiments only")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed (RCCL) code path even with one rank — rehearsal of the N>1 leg on a 1-GPU box")
    return ap.parse_args()


def spmv_bytes(n, nnz, s):
    """nnz*(s+4) + (n+1)*4 + n*s (x) + n*s (y)."""
    return nnz * (s + 4) + (n + 1) * 4 + 2 * n * s


STREAM_NAMES = {0: "csr", 1: "offset-codes", 2: "pair-codes"}


def stream_info(A, n, nnz, s):
    """What the SpMV of handle A actually streams (csrc/spmv_dict.hip) and the bytes that format needs per launch
    (x and y counted once, like spmv_bytes): plain CSR nnz*(s+4); offset codes nnz*(s+1); pair codes nnz*1."""
    mode, n_off, n_pair = A.stream_format()
    per_nnz = {0: s + 4, 1: s + 1, 2: 1}[mode]
    return dict(stream=STREAM_NAMES[mode], mode=mode, distinct_offsets=n_off, distinct_pairs=n_pair,
                bytes_per_nnz=per_nnz, format_bytes_per_launch=nnz * per_nnz + (n + 1) * 4 + 2 * n * s)


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


def time_solve(torch, dist, solver, precond, rhs, x, steps, warmup, world, profile=True):
    """W untimed warm-up iterations, then exactly K timed ones; returns seconds (max over ranks).
    profile: bracket every SpMV launch with HIP events on the solver's stream (2 event records per
    launch — negligible against the 1.2 ms SpMV of cfg 5, not against the 15 us SpMV of cfg 2)."""
    x.zero_()
    if warmup > 0:
        run_fixed_iterations(solver, precond, rhs, x, warmup)
    x.zero_()
    solver.set_profile(bool(profile))
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


def cpu_baseline_poisson3d(nx, ny, nz_full, target_s):
    """The reference's CPU path (row-parallel SpMV + serial BLAS-1) restated in C (oracle/),
    timed on this box's host cores on a z-slab sample of the same system."""
    import numpy as np

    from oracle import oracle as orc
    from sprsolve_amd import gen
    cores = os.cpu_count() or 1
    orc.set_threads(cores)
    nz_s = 16
    indptr, indices, data, rhs = gen.poisson3d(nx, ny, nz_s)
    n = rhs.size
    ip64 = indptr.astype(np.int64); ix64 = indices.astype(np.int64)
    # calibrate with 3 iterations, then run enough for ~target_s
    t0 = time.perf_counter()
    orc.bicgstab(ip64, ix64, data, rhs, np.zeros(n), 3, 0.0, parallel=True)
    per_it = (time.perf_counter() - t0) / 3
    its = int(max(5, min(200, target_s / max(per_it, 1e-6))))
    t0 = time.perf_counter()
    r = orc.bicgstab(ip64, ix64, data, rhs, np.zeros(n), its, 0.0, parallel=True)
    dt = time.perf_counter() - t0
    assert r.status == orc.INSUFFICIENT_ITER
    it_s_sample = its / dt
    scale = (nx * ny * nz_full) / float(n)
    return dict(value=it_s_sample / scale, unit="iterations/s", cores=cores, kind="port",
                sample="%d BiCGStab iterations of the CPU restatement (oracle/, rayon-style row-parallel SpMV on %d "
                       "threads + serial BLAS-1, usize indices) on a %dx%dx%d slab (%d rows) of the same 7-point system; "
                       "%.2f it/s on the slab, scaled by rows to the %dx%dx%d system" %
                       (its, cores, nx, ny, nz_s, n, it_s_sample, nx, ny, nz_full))


def cpu_baseline_rows(build, full_rows, target_s, what):
    import numpy as np

    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    orc.set_threads(cores)
    indptr, indices, data, rhs, diag, fn, n = build()
    ip64 = indptr.astype(np.int64); ix64 = indices.astype(np.int64)
    t0 = time.perf_counter()
    fn(ip64, ix64, data, rhs, np.zeros_like(rhs), 3, 0.0, precond_diag=diag, parallel=True)
    per_it = (time.perf_counter() - t0) / 3
    its = int(max(5, min(500, target_s / max(per_it, 1e-6))))
    t0 = time.perf_counter()
    fn(ip64, ix64, data, rhs, np.zeros_like(rhs), its, 0.0, precond_diag=diag, parallel=True)
    dt = time.perf_counter() - t0
    return dict(value=(its / dt) / (full_rows / float(n)), unit="iterations/s", cores=cores, kind="port",
                sample="%d iterations of the CPU restatement (oracle/, row-parallel SpMV on %d threads + serial BLAS-1) "
                       "on %s (%d rows), scaled by rows" % (its, cores, what, n))


def main():
    args = parse()
    import torch   # first: libsprsolve_hip.so then shares torch's HIP runtime (same soname)
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    local_rank = int(os.environ.get("SPRS_BENCH_DEVICE", local_rank))   # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    elif args.force_dist:
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank))

    import numpy as np

    import sprsolve_amd as sa
    from sprsolve_amd import gen_torch
    ctx = sa.default_ctx(local_rank)
    ctx.set("spmv_dict", {"auto": -1, "csr": 0, "offsets": 1, "dict": 2}[args.stream])
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
        A = sa.HipCsr.from_device((n, n), nnz, ip, ix, dv, adopt=True, ctx=ctx)
        P = sa.DiagPrecond.new(diag.cpu().numpy(), ctx=ctx)
        s = sa.BiCGStab.new(A, n)
        x = torch.zeros(n, dtype=torch.float64, device=dev)
        dt, _ = time_solve(torch, dist, s, P, rhs, x, steps, warmup, 1, profile=False)
        _, prof = time_solve(torch, dist, s, P, rhs, x, steps, 0, 1, profile=True)    # separate pass: per-launch SpMV time
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
        # stand-alone SpMV timing (back-to-back launches, HIP events on the library's stream)
        y = torch.empty_like(x)
        ms_alone = A.time_mul_vec(rhs, y, reps=200)
        t_spmv = prof["spmv_ms_total"] / max(prof["spmv_launches"], 1) * 1e-3
        return dict(workload="cfg2: 1000x1000 2-D 5-point Poisson (Dirichlet rows), n=1e6, nnz=4984016, BiCGStab + Jacobi",
                    value=steps / dt, ms_per_step=dt / steps * 1e3, n=n, nnz=nnz,
                    pcie_inclusive_it_s=steps / pcie_dt,
                    spmv_us_in_solve=t_spmv * 1e6, spmv_us_back_to_back=ms_alone * 1e3,
                    spmv_GBs_in_solve=bs / t_spmv / 1e9, spmv_GBs_back_to_back=bs / (ms_alone * 1e-3) / 1e9,
                    spmv_bytes=bs, iter_bytes_reference_oplist=2 * bs + 26 * n * 8 + 2 * n * 24,
                    effective_GBs=(2 * bs + 26 * n * 8 + 2 * n * 24) * steps / dt / 1e9,
                    converge_check=dict(tol=1e-8, iters=its, rel_res=res, max_abs_err_vs_exact=err),
                    note="working set (~140 MB) fits the 256 MiB Infinity Cache: GB/s here is not an HBM figure"), t_spmv, bs

    if args.workload == "poisson3d":
        nx, ny, nz = (int(v) for v in args.grid.lower().split("x"))
        if world == 1 and not args.force_dist:
            ip, ix, dv, rhs = gen_torch.poisson3d(nx, ny, nz, device=dev)
            n = nx * ny * nz
            nnz = int(ip[-1].item())
            A = sa.HipCsr.from_device((n, n), nnz, ip, ix, dv, adopt=True, ctx=ctx)
            s = sa.BiCGStab.new(A, n)
            x = torch.zeros(n, dtype=torch.float64, device=dev)
            dt, prof = time_solve(torch, dist, s, None, rhs, x, args.steps, args.warmup, 1)
            t_spmv = prof["spmv_ms_total"] / max(prof["spmv_launches"], 1) * 1e-3
            bs = spmv_bytes(n, nnz, 8)
            # correctness on the very same objects: the exact solution is all ones
            x.zero_()
            its, res = s.solve(rhs, x, 5000, 1e-8)
            err = float((x - 1.0).abs().max().item())
            check = dict(tol=1e-8, iters=its, rel_res=res, max_abs_err_vs_exact=err)
            n_glob, nnz_glob = n, nnz
            sinfo = stream_info(A, n, nnz, 8)
            if sinfo["mode"] != 0 and not args.no_also:
                # the same solve on the plain CSR stream (12 B/nnz): the apples-to-apples figure against SURVEY's
                # CSR roofline; identical iterates (the streams are bit-identical), so only the time differs
                ctx.set("spmv_dict", 0)
                t_csr, p_csr = time_solve(torch, dist, s, None, rhs, x, max(args.steps // 2, 10), min(args.warmup, 5), 1)
                ctx.set("spmv_dict", {"auto": -1, "csr": 0, "offsets": 1, "dict": 2}[args.stream])
                k_csr = max(args.steps // 2, 10)
                tl = p_csr["spmv_ms_total"] / max(p_csr["spmv_launches"], 1) * 1e-3
                also["cfg5_plain_csr_stream"] = dict(
                    value=k_csr / t_csr, unit="iterations/s", ms_per_step=t_csr / k_csr * 1e3, steps=k_csr,
                    spmv_us_in_solve=tl * 1e6, spmv_GBs=bs / tl / 1e9, spmv_frac_of_hbm_peak=bs / tl / 1e9 / HBM_PEAK_GBS,
                    note="spmv_kernel<double> on (col_idx, val); same matrix, same vectors, same iterates")
        else:
            from sprsolve_amd import dist as sdist
            res_ = sdist.bench_poisson3d(torch, dist, ctx, rank, world, nx, ny, nz, args.steps, args.warmup, time_solve,
                                         exchange=args.exchange)
            dt, prof, t_spmv, bs, check, n_glob, nnz_glob, sinfo = res_
        it_bytes = 2 * spmv_bytes(n_glob, nnz_glob, 8) + 26 * n_glob * 8
        traffic, traffic_note = None, "no PMC summary found"
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        pmc_key = {0: "bench_csr", 1: "bench_offsets", 2: "bench_pair"}[sinfo["mode"]]
        if world == 1 and (nx, ny, nz) == (500, 500, 200) and os.path.exists(pmc_path):
            # HBM-side bytes per launch from rocprofv3 PMC (separate FETCH_SIZE / WRITE_SIZE passes of this same
            # command, gfx950 x2 FETCH correction calibrated in the SpMV's access pattern) — profiles/
            with open(pmc_path) as f:
                pm = json.load(f).get(pmc_key, {}).get("spmv_in_solve")
            if pm:
                traffic, traffic_note = pm["traffic_bytes"], "profiles/r01_pmc_summary.json[%s]: %s" % (pmc_key, pm["note"])
        kernel = {0: "spmv_kernel<double> (plain CSR stream, LDS product path, fused dot epilogue)",
                  1: "spmv_dict_kernel<double, PAIR=false> (one-byte column-offset codes + values)",
                  2: "spmv_pair2_kernel<DOT> (one-byte (offset, value) pair codes, two rows per lane, uniform blocks from a "
                     "scalar pattern; csrc/spmv_dict.hip)"}[sinfo["mode"]]
        fb = sinfo["format_bytes_per_launch"]          # per rank, like bs
        roof = dict(bound="hbm", kernel=kernel,
                    achieved=bs / t_spmv / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=bs / t_spmv / 1e9 / HBM_PEAK_GBS,
                    traffic=traffic, traffic_note=traffic_note, algorithmic_bytes_per_launch=bs,
                    avg_launch_us=t_spmv * 1e6, launches=prof["spmv_launches"],
                    stream=sinfo["stream"], format_bytes_per_launch=fb,
                    format_GBs=fb / t_spmv / 1e9, format_frac_of_hbm_peak=fb / t_spmv / 1e9 / HBM_PEAK_GBS,
                    note="per rank; `achieved` / `frac` use SURVEY §8d's CSR bytes nnz*12 + (n+1)*4 + 2*n*8 (x counted once), as the "
                         "contract says"
                         + ("" if sinfo["mode"] == 0 else "; the %s stream moves %d B/nnz instead of 12 (lossless, y bit-identical), so `frac` "
                            "exceeds what a 12 B/nnz CSR kernel could reach at the HBM peak — `format_*` is the same time against the bytes "
                            "this stream really needs, and also.cfg5_plain_csr_stream is the plain CSR kernel on the same matrix; the "
                            "compressed kernel is bound by the CUs' vector-memory issue and latency, not by HBM (DESIGN.md §5)"
                            % (sinfo["stream"], sinfo["bytes_per_nnz"]))
                         + ("; N>1: the bracketed time includes the wait for the halo exchange of the boundary rows" if world > 1 else ""))
        out = dict(metric="BiCGStab iterations/s (f64, 50M-row 7-point 3-D Poisson) + CSR SpMV GB/s vs HBM roofline",
                   value=args.steps / dt, unit="iterations/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="strong", vs_baseline=None,
                   dtype="f64", data="synthetic",
                   config=dict(workload="cfg5: %dx%dx%d 7-point 3-D Poisson, n=%d, nnz=%d, BiCGStab (no preconditioner), "
                                        "tol=0 fixed %d iterations" % (nx, ny, nz, n_glob, nnz_glob, args.steps),
                               rows=n_glob, nnz=nnz_glob, index_type="i32", partition="z-slabs x%d" % world,
                               spmv_stream=sinfo,
                               bytes_per_iteration_reference_oplist=it_bytes),
                   effective_GBs_reference_oplist=it_bytes * args.steps / dt / 1e9,
                   roofline=roof, converge_check=check)
        if world == 1 and rank == 0:
            x = None                      # release the solution vector before the microbench allocates its own
            sc = stream_ceiling(torch, ctx, n_glob if not args.force_dist else min(n_glob, 50_000_000))
            roof["measured_stream_ceiling"] = dict(sc, note="this library's axpy (2R+1W) and a d2d copy (1R+1W) on n doubles, "
                                                            "back to back — SURVEY §8d's secondary denominator")
            best = max(sc.values())
            roof["frac_of_measured_stream"] = roof["achieved"] / best
            roof["format_frac_of_measured_stream"] = roof["format_GBs"] / best
            if "cfg5_plain_csr_stream" in also:
                also["cfg5_plain_csr_stream"]["spmv_frac_of_measured_stream"] = also["cfg5_plain_csr_stream"]["spmv_GBs"] / best
        if world == 1 and not args.no_also and not args.force_dist:
            r2, _, _ = bench_poisson2d(500, 50)
            also["cfg2_poisson2d_1M_bicgstab_jacobi"] = r2
        if world == 1 and rank == 0 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_poisson3d(nx, ny, nz, args.cpu_seconds)
    elif args.workload == "poisson2d":
        if world != 1:
            raise SystemExit("poisson2d is a single-GPU workload (cfg 2)")
        r2, t_spmv, bs = bench_poisson2d(args.steps, args.warmup)
        out = dict(metric="BiCGStab iterations/s (f64, 1M-row 2-D Poisson + Jacobi) + CSR SpMV GB/s",
                   value=r2["value"], unit="iterations/s", n_gpus=1, steps=args.steps, warmup=args.warmup,
                   ms_per_step=r2["ms_per_step"], higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f64",
                   data="synthetic", config=dict(workload=r2["workload"]),
                   roofline=dict(bound="hbm", achieved=bs / t_spmv / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=bs / t_spmv / 1e9 / HBM_PEAK_GBS, traffic=None,
                                 note="cache-resident working set; see detail.note"),
                   detail=r2)
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            from sprsolve_amd import gen

            def build():
                ip, ix, dv = gen.grid_laplacian_dirichlet(1000, 1000)
                return ip, ix, dv, gen.dirichlet_rhs(1000, 1000), np.where(np.diff(ip) == 1, 1.0, -4.0), orc.bicgstab, 10**6
            out["cpu_baseline"] = cpu_baseline_rows(build, 10**6, args.cpu_seconds, "the full cfg-2 system")
    else:
        if world != 1:
            raise SystemExit("%s is a single-GPU workload" % args.workload)
        from sprsolve_amd import gen
        if args.workload == "banded":
            n = 10**6
            ip, ix, dv, rhs = gen.symmetric_banded(n)
            solver_cls, label, sbytes = sa.MinRes, "cfg3: symmetric banded (hbw 4), n=1e6, nnz=8999980, MINRES", 8
        else:
            ip, ix, dv, rhs, _ = gen.complex_symmetric_grid(500, 1000)
            n = 500000
            solver_cls, label, sbytes = sa.CSMinRes, "cfg4: complex-symmetric 500x1000 grid, n=5e5, nnz=2497000, CSMINRES", 16
        A = sa.HipCsr.new((n, n), ip, ix, dv, ctx=ctx)
        s = solver_cls.new(A, n)
        tdt = torch.float64 if sbytes == 8 else torch.complex128
        drhs = torch.from_numpy(rhs).to(dev)
        x = torch.zeros(n, dtype=tdt, device=dev)
        dt, prof = time_solve(torch, dist, s, None, drhs, x, args.steps, args.warmup, 1)
        t_spmv = prof["spmv_ms_total"] / max(prof["spmv_launches"], 1) * 1e-3
        bs = spmv_bytes(n, int(ip[-1]), sbytes)
        out = dict(metric="%s iterations/s + CSR SpMV GB/s" % solver_cls.__name__, value=args.steps / dt, unit="iterations/s",
                   n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
                   scaling="strong", vs_baseline=None, dtype="f64" if sbytes == 8 else "c64", data="synthetic",
                   config=dict(workload=label),
                   roofline=dict(bound="hbm", achieved=bs / t_spmv / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=bs / t_spmv / 1e9 / HBM_PEAK_GBS, traffic=None, avg_launch_us=t_spmv * 1e6))
    if also:
        out["also"] = also
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
