#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the synchronous stepper on BASELINE config 2.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (mean field + occupancy at every particle -> rates -> Philox draw -> commit)
over all N particles, in the formulation --method selects: "tiles" (= "auto", the default: the reference's histogram ->
smoothing -> gather kept incrementally on the L sites, the whole step in ONE kernel over a site-centric state; the steps of
a call run inside one launch, the resident loop, where the grid of tiles fits the device), "lattice" (the same field, three
kernels, particle-indexed state) or "pairs" (all-pairs tile kernel).  Workload: N=100000 particles on
L=200000 sites, K=1, reflecting walls, sigma=0.005 (sigma_g=1000, 4001 taps), beta=0.7, rate_active=5, rate_diffusion=0.02,
dt=0.0125, uniform-in-box synthetic initial condition, float64.  With --gpus N ONE system is sharded by SITE RANGE over
the ranks (strong scaling, BASELINE config 3): every k-th step each rank sends its boundary state to its two neighbour
ranks -- by peer stores into IPC-mapped buffers ("exchange": "ipc-peer"), else by ncclSend / ncclRecv ("rccl");
--workload config4 deals independent ensembles to the ranks instead (weak scaling, no data-path communication).
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"

WORK = dict(N=100_000, L=200_000, K=1, xlim=1.0, sigma=0.005, beta=0.7, rate_active=5.0, rate_diffusion=0.02,
            dt=0.0125, seed=0)
# other BASELINE configurations, for the record only (never the default bench line): --workload config4 / config5
EXTRA = {
    "config4": dict(WORK, N=50_000, L=100_000, betas=[3.0 * i / 15 for i in range(16)]),     # 16 beta ensembles, one GPU
    "config5": dict(WORK, N=1_000_000, L=2_000_000, fp32=True),                               # BASELINE config 5 says float32: the 32-bit field (--f64 for the exact one)
    # not a BASELINE configuration: the same model at a size where the working set (1.5 GB) no longer fits the caches,
    # short-ranged kernel (sigma_g = 10 sites) -- shows the kernels against the HBM roofline they are priced on
    "hbm": dict(WORK, N=16_000_000, L=32_000_000, sigma=10.0 / 32_000_000),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
ALGO_BYTES_PER_PARTICLE_STEP = 16.0   # SURVEY 8d (all-pairs): 4 B state read + 4 B write + 4 B proposal + 4 B occupancy/commit
# Lattice formulation, algorithmic bytes per launch, N particles, L sites, C changed particles, D deposits:
#   propose_lattice  N * (4 state + 4 index + 16 {W,S} + 12 occupancy + 1 proposal) + 4 L (site counters cleared)
#   apply            N * (1 proposal + 4 state + 4 index) + C * (8 state/source word + 8 occupancy) + 4 D + 16 N / 64
#   field_update     L * 32 ({W,S} read + write) + 4 D
#   tile_step        L * (32 {W,S} read + write [16 with the 32-bit field]  +  8 K cell words read + write) + 8 D (deposit written, read) + 8 T (counters)
def step_algo_bytes(kernel, N, L, K, deposits, fp32=False, changed=None):
    if changed is None:               # particles whose state a step changes; only `apply` uses it.  bench.py counts them (one step, states compared)
        changed = 0.6 * deposits
    return {"propose_lattice": 37.0 * N + 4.0 * L,
            "apply": 9.25 * N + 16.0 * changed + 4.0 * deposits,
            "field_update": 32.0 * L + 4.0 * deposits,
            "tile_step": ((16.0 if fp32 else 32.0) + 8.0 * K) * L + 8.0 * deposits + 8.0 * L / 316.0}.get(kernel, 0.0)


VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD32 x 2.4 GHz
LDS_CYCLES_PER_64_PAIRS = 5.0  # measured (PMC 4.93): 4.5 per table gather (2.0 + 2.5 bank conflicts) + 0.5 source broadcast
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")   # rocprofv3 FETCH_SIZE / WRITE_SIZE passes, per workload


def measured_traffic_bytes(workload, kernel, steps_per_launch=1):
    """HBM-side bytes per launch of `kernel` in `workload` from the committed rocprofv3 --pmc passes of this round's
    binary (bench.py cannot collect PMC itself; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide
    streaming reads is NOT applied: these kernels read 4-16 bytes per lane).  None when no pass was recorded for this
    workload and kernel -- never another workload's number."""
    try:
        with open(TRAFFIC_FILE) as fh:
            d = json.load(fh)["workloads"][workload][kernel]
        return (d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1024.0 / d.get("steps_per_dispatch", 1) * steps_per_launch
    except Exception:
        return None


def initial_state(w):
    rng = np.random.default_rng(w["seed"])
    pos = rng.choice(w["L"], size=w["N"], replace=False).astype(np.int32)
    spin = rng.choice(np.array([1, -1], np.int8), size=w["N"])
    return pos, spin


def make_handle(capi, w, device=0, rank=0, world=1, method="auto"):
    return capi.Handle(L=w["L"], K=w["K"], periodic=False, sigma_grid=w["sigma"] / (w["xlim"] / w["L"]),
                       rate_diffusion=w["rate_diffusion"], rate_active=w["rate_active"], beta=w.get("betas", [w["beta"]]),
                       dt=w["dt"], seed=w["seed"], n_particles=w["N"], device=device, rank=rank, world=world, method=method,
                       fp32=bool(w.get("fp32", False)))


def cpu_baseline(w, budget_s=12.0):
    """The oracle's C port of the same step, timed on one host core on a bounded sample."""
    from oracle.gillespie_numpy import LatticeGasParams
    from oracle import sync_oracle as so
    par = LatticeGasParams.from_kwargs(L=w["L"], xlim=w["xlim"], rate_diffusion=w["rate_diffusion"],
                                       rate_active=w["rate_active"], beta=w["beta"], scale_rates=False,
                                       local_kernel_sigma=w["sigma"], site_capacity=w["K"])
    orc = so.SyncOracle(par, dt=w["dt"], seed=w["seed"])
    orc.set_state(*initial_state(w))
    orc.run(1)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s and n < 50:
        orc.run(1)
        n += 1
    el = time.perf_counter() - t0
    return {"value": w["N"] * n / el, "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} synchronous steps of the same N={w['N']} workload by oracle/sync_oracle.c "
                      f"(lattice formulation, gcc -O2, 1 thread), {el:.1f} s"}


def cpu_reference_loop(w, budget_s=12.0):
    """The reference's own algorithm (one Gillespie event per iteration, NumPy) restated in
    oracle/gillespie_numpy.py, on the same N and L: events/s and the equivalent particle-steps/s
    (one event advances ONE particle; an event is worth 1/(mean rate * dt) particle-steps)."""
    from oracle.gillespie_numpy import GillespieOracle
    orc = GillespieOracle(L=w["L"], xlim=w["xlim"], rate_diffusion=w["rate_diffusion"], rate_active=w["rate_active"],
                          beta=w["beta"], N=w["N"], scale_rates=False, local_kernel_sigma=w["sigma"],
                          site_capacity=w["K"], rng=np.random.default_rng(w["seed"]))
    pos, sigma = orc.init_particles()
    bound = np.zeros(len(pos), bool)
    cp = np.bincount(pos[sigma == 1], minlength=w["L"])
    cm = np.bincount(pos[sigma == -1], minlength=w["L"])
    t0 = time.perf_counter()
    n, sim = 0, 0.0
    while time.perf_counter() - t0 < budget_s and n < 200:
        field = orc.mean_field(cp, cm)
        pos, sigma, bound, tau = orc.fire_event(pos, sigma, bound, field, cp, cm, sim, ([], []))
        sim += tau
        n += 1
    el = time.perf_counter() - t0
    return {"events_per_s": n / el, "equiv_particle_steps_per_s": (sim * w["N"] / w["dt"]) / el, "cores": 1,
            "sample": f"{n} Gillespie events (compute_local_m_field + step_gillespie restated in NumPy), {el:.1f} s"}


class stdout_to_stderr:
    """RCCL prints a banner to fd 1 while a communicator is created; the contract is ONE JSON line on stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def bench_pde(args):
    """--workload pde (for the record, never the default line): the reference's sweep shape
    (IMEX_PDE_solver_run_sweep.py:26-48: L=1000, dt=5e-4, periodic, bidirectional, flat kernel, 1000 tracers) for
    256 beta values in one launch; a step = one pass of IMEXPDE.solve's loop body for one system."""
    pde = importlib.import_module(PKG + ".pde")
    kw = dict(L=1000, xlim=1.0, dt=5e-4, gamma=0.2, lam=0.6, bc="periodic", active_model="bidirectional",
              gaussian_kernel=True, kernel_sigma=1e5 - 10, snapshot_interval=50, seed=0)
    nsteps, nsys = args.steps, 256
    s = pde.IMEXPDE(T=nsteps * kw["dt"], beta=2.0, record_fft=False, **kw)
    s.initialize(mode="homogeneous", rho0=1.0, noise=0.3, n_tracers=1000)
    betas = np.linspace(0.0, 3.0, nsys)
    s.solve_batch(betas[:2])                                   # warm-up (module load, first launch)
    r = s.solve_batch(betas)
    ms = r["kernel_ms"]
    out = {"metric": "PDE system-steps/sec (IMEXPDE.solve loop body), 256 systems x L=1000", "value": nsys * (nsteps + 1) / (ms * 1e-3),
           "unit": "system-steps/s", "n_gpus": 1, "steps": nsteps, "warmup": 0, "ms_per_step": ms / (nsteps + 1),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "IMEX_PDE_solver_run_sweep.py shape: L=1000, dt=5e-4, periodic, bidirectional, ring-wide "
                                  "Gaussian kernel (1001 taps), 1000 tracers, 256 beta values in one launch",
                      "per_system_steps_per_s": (nsteps + 1) / (ms * 1e-3)}}
    if not args.no_cpu_baseline:
        from oracle.pde_numpy import PdeOracle
        n_cpu = 300
        orc = PdeOracle(T=n_cpu * kw["dt"], beta=2.0, **kw)
        orc.initialize(mode="homogeneous", rho0=1.0, noise=0.3, n_tracers=1000)
        t0 = time.perf_counter()
        orc.solve()
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": (n_cpu + 1) / el, "unit": "system-steps/s", "cores": 1, "kind": "port",
                               "sample": f"{n_cpu + 1} steps of ONE system by oracle/pde_numpy.py (scipy spsolve + numpy rfft, "
                                         f"bit-identical to the reference's IMEXPDE), {el:.1f} s"}
    print(json.dumps(out))


def bench_gillespie(args):
    """--workload gillespie (for the record): the reference's sweep shape (..._sweep_beta.py:829-857: L=1000, N=500, K=1,
    sigma=0.005, T=20, obs_dt=0.1) as 1024 independent systems (beta = linspace(0,3,32) x 32 runs) in ONE launch of the
    device-resident exact event loop; unit = Gillespie events."""
    gil = importlib.import_module(PKG + ".gillespie")
    L, N, nsys = 1000, 500, 1024
    rng = np.random.default_rng(0)
    states = []
    for _ in range(nsys):
        states.append((rng.choice(L, size=N, replace=False).astype(np.int32), rng.choice(np.array([1, -1], np.int8), size=N)))
    betas = np.repeat(np.linspace(0.0, 3.0, 32), 32)
    times = np.arange(0.0, 20.0, 0.1)
    kw = dict(L=L, K=1, periodic=False, sigma_grid=0.005 * L, rate_diffusion=0.02, rate_active=5.0, times_obs=times, T=20.0, seed=1)
    gil.run_raw(betas=betas[:4], states=states[:4], want_states=False, **kw)          # warm-up
    r = gil.run_raw(betas=betas, states=states, want_states=False, **kw)
    ev, ms = int(r["n_events"].sum()), r["kernel_ms"]
    out = {"metric": "exact Gillespie events/sec, 1024 systems x N=500 (reference sweep shape)", "value": ev / (ms * 1e-3), "unit": "events/s",
           "n_gpus": 1, "steps": ev, "warmup": 0, "ms_per_step": ms / max(1, int(r["n_events"].max())), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "L=1000, N=500, K=1, sigma=0.005, rate_active=5, rate_diffusion=0.02, T=20, obs_dt=0.1; 32 beta x 32 runs "
                                  "as 1024 persistent workgroups", "events_per_system": float(r["n_events"].mean()),
                      "per_system_events_per_s": float(r["n_events"].max()) / (ms * 1e-3), "kernel_ms": ms}}
    if not args.no_cpu_baseline:
        from oracle.gillespie_numpy import GillespieOracle
        orc = GillespieOracle(L=L, xlim=1.0, rate_diffusion=0.02, rate_active=5.0, beta=0.7, N=N, scale_rates=False,
                              local_kernel_sigma=0.005, site_capacity=1, rng=np.random.default_rng(0))
        t0 = time.perf_counter()
        orc.run(T=20.0, obs_dt=0.1, max_events=20000)
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": orc.n_events / el, "unit": "events/s", "cores": 1, "kind": "port",
                               "sample": f"{orc.n_events} events of ONE system by oracle/gillespie_numpy.py (the reference's loop, "
                                         f"bit-identical for seeded generators), {el:.1f} s"}
    print(json.dumps(out))


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, popen=None, port=None):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU) with the
    torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), wait for all of them and relay
    rank 0's single JSON line.  The parent never touches HIP or torch (a process that has initialised the GPU must not
    be re-exec'd; here nothing is exec'd at all: the ranks are children).  Returns the exit code."""
    import subprocess
    popen = popen or subprocess.Popen
    port = port or free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), APS_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stdout carries the JSON line; the other ranks have nothing to say on stdout
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                           stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    # rank 0's pipe is drained by a thread so that a rank that dies early never leaves the others (blocked in a
    # collective) running: as soon as one rank fails, the rest are ended -- by their own PIDs, nothing else
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read() if procs[0].stdout else ""), daemon=True)
    reader.start()
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except Exception:                        # noqa: BLE001
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = chunks[0] if chunks else ""
    if any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        if out0:
            print(out0, file=sys.stderr)
        return next(c for c in codes if c) or 1
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        print(f"bench.py: expected ONE JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="config2", choices=["config2", "pde", "gillespie"] + sorted(EXTRA))
    ap.add_argument("--method", default="auto", choices=["auto", "lattice", "pairs", "tiles"])
    ap.add_argument("--repeats", type=int, default=5, help="timed repeats of the K steps; the line reports their median")
    ap.add_argument("--fp32", action="store_true", help="32-bit integer field (aps_params.fp32); default only for --workload config5")
    ap.add_argument("--f64", action="store_true", help="force the exact binary64 field")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:                                   # no launcher: this process only starts the ranks
            raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={os.environ['WORLD_SIZE']} of the launcher")
    # a rank that hangs (a collective that never completes on some fabric) must not sit there until the caller's own limit:
    # after APS_BENCH_WATCHDOG_S seconds (default 900) the process says so and leaves with code 124; a launcher or
    # spawn_ranks then ends the other ranks
    import threading
    limit = float(os.environ.get("APS_BENCH_WATCHDOG_S", "900"))

    def _give_up():
        print(f"bench.py: rank {os.environ.get('RANK', '0')}: no result after {limit:.0f} s -- giving up (exit 124)", file=sys.stderr, flush=True)
        os._exit(124)
    dog = threading.Timer(limit, _give_up)
    dog.daemon = True
    dog.start()
    if args.workload == "pde":
        return bench_pde(args)
    if args.workload == "gillespie":
        return bench_gillespie(args)
    w = dict(WORK) if args.workload == "config2" else dict(EXTRA[args.workload])
    if args.fp32:
        w["fp32"] = True
    if args.f64:
        w["fp32"] = False
    n_ens = len(w.get("betas", [0]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    capi = importlib.import_module(PKG + ".capi")
    if capi.device_count() < 1:
        raise SystemExit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
    pos, spin = initial_state(w)
    roof, comm_path, extra = None, "", {}
    sharded_path = world > 1 or os.environ.get("APS_BENCH_FORCE_SHARDED") == "1"   # the switch lets one rank rehearse it
    scaling, sharding_note, exchange, ranks_seen = "strong", "", "", None
    if sharded_path and n_ens > 1:
        # Independent ensembles (BASELINE config 4) across GPUs: every rank steps its own n_ens ensembles (its own beta
        # values and Philox streams), no data-path communication at all -- the weak / throughput curve.
        import torch
        import torch.distributed as dist
        device = int(os.environ.get("APS_BENCH_DEVICE", local_rank))
        torch.cuda.set_device(device)
        with stdout_to_stderr():
            dist.init_process_group("gloo")                  # barriers and the MAX over ranks only
        lo, hi = min(w["betas"]), max(w["betas"])
        all_betas = [lo + (hi - lo) * i / (n_ens * world - 1) for i in range(n_ens * world)]
        wr = dict(w, betas=all_betas[rank * n_ens:(rank + 1) * n_ens])
        h = capi.Handle(L=wr["L"], K=wr["K"], periodic=False, sigma_grid=wr["sigma"] / (wr["xlim"] / wr["L"]),
                        rate_diffusion=wr["rate_diffusion"], rate_active=wr["rate_active"], beta=wr["betas"], dt=wr["dt"],
                        seed=wr["seed"], n_particles=wr["N"], device=device, ensemble_base=rank * n_ens, method=args.method,
                        fp32=bool(wr.get("fp32", False)))
        for e in range(n_ens):
            h.set_state(pos, spin, ensemble=e)
        h.step(args.warmup)
        times = []
        for _ in range(max(1, args.repeats)):
            dist.barrier()
            t0 = time.perf_counter()
            h.step(args.steps)
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.barrier()
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            times.append(float(el.item()))
        elapsed = float(np.median(times))
        extra = {"repeats": len(times), "repeats_ms_per_step": [t / args.steps * 1e3 for t in times], "graph_replay": h.step_info()[0] > 0}
        scaling, exchange, ranks_seen = "weak", "none (independent ensembles)", dist.get_world_size()
        sharding_note = f"{n_ens} ensembles per GPU x {world} GPU(s), no data-path communication"
        n_ens_total = n_ens * world
    elif sharded_path:
        import torch
        import torch.distributed as dist
        device = int(os.environ.get("APS_BENCH_DEVICE", local_rank))   # rehearsals put several ranks on one GPU
        torch.cuda.set_device(device)
        with stdout_to_stderr():                             # gloo announces its connections on stdout
            dist.init_process_group("gloo")                  # rendezvous / barriers only; the data path is RCCL below

        def all_agree(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        # ONE system over several GPUs (BASELINE config 3).  Preferred: site-range shards of the tiles formulation
        # (every rank steps its own tiles; halo by ncclSend / ncclRecv); if the table's reach does not fit the ranks'
        # ranges: particle-index shards (every rank applies all proposals; all-gather of 1 byte per particle).
        h, site_shards = None, False
        if args.method in ("auto", "tiles"):
            try:
                h = make_handle(capi, w, device=device, rank=rank, world=world, method="tiles")
                site_shards = True
            except capi.ApsError as exc:
                print(f"[rank {rank}] site-range shards unavailable ({exc}); particle-index shards instead", file=sys.stderr)
        site_shards = all_agree(site_shards)
        if not site_shards:
            if h is not None:
                h.close()
            h = make_handle(capi, w, device=device, rank=rank, world=world, method="auto" if args.method == "tiles" else args.method)
        h.set_state(pos, spin)
        forced = os.environ.get("APS_BENCH_EXCHANGE", "")
        run, path = None, ""
        if site_shards and forced in ("", "ipc-peer"):
            # preferred: peer stores into the neighbour's IPC-mapped landing buffer (include/aps.h: aps_ipc_export / aps_ipc_connect);
            # the 256-byte blobs go round over gloo once
            ok, blobs = True, [None] * world
            try:
                mine_blob = h.ipc_export()
            except Exception as exc:                         # noqa: BLE001
                ok, mine_blob = False, None
                print(f"[rank {rank}] peer-store transport unavailable ({exc})", file=sys.stderr)
            dist.all_gather_object(blobs, mine_blob)
            if all_agree(ok and all(b is not None for b in blobs)):
                try:
                    h.ipc_connect(blobs[rank - 1] if rank > 0 else None, blobs[rank + 1] if rank + 1 < world else None)
                except Exception as exc:                     # noqa: BLE001
                    ok = False
                    print(f"[rank {rank}] peer-store transport unavailable ({exc})", file=sys.stderr)
                if all_agree(ok):
                    try:
                        h.step(max(1, h.halo_info()[0]))     # up to the first exchange: the mapping works in both directions
                    except Exception as exc:                 # noqa: BLE001
                        ok = False
                        print(f"[rank {rank}] peer-store exchange failed ({exc})", file=sys.stderr)
                    if all_agree(ok):
                        run, exchange, ranks_seen = h.step, "ipc-peer", len([b for b in blobs if b is not None])
                        path = "peer stores into the neighbour ranks' IPC-mapped landing buffers + one arrival word per block, inside aps_step"
                if run is None:                              # a half-connected handle is of no use: start over for the next transport
                    dist.barrier()
                    h.close()
                    h = make_handle(capi, w, device=device, rank=rank, world=world, method="tiles")
                    h.set_state(pos, spin)
        if run is None and forced in ("", "rccl"):
            ids = [None]
            if rank == 0:                                    # a failure here must not unbalance the collectives below
                try:
                    with stdout_to_stderr():
                        ids[0] = capi.comm_unique_id()
                except Exception as exc:                     # noqa: BLE001
                    print(f"[rank 0] in-library RCCL unavailable ({exc})", file=sys.stderr)
            dist.broadcast_object_list(ids, src=0)
            ok = ids[0] is not None
            if ok:
                try:
                    with stdout_to_stderr():
                        h.comm_init(ids[0])
                        h.comm_selftest()                    # ncclSend / ncclRecv rank -> itself: the transport calls work at all
                        h.step(max(1, h.halo_info()[0]))     # up to the first exchange (lazy channel setup) also under the redirect
                except Exception as exc:                     # noqa: BLE001
                    ok = False
                    print(f"[rank {rank}] in-library RCCL unavailable ({exc})", file=sys.stderr)
            if all_agree(ok):
                run, exchange = h.step, "rccl"
                path = "in-library RCCL: " + ("ncclSend/ncclRecv of the halo to the two neighbour ranks" if site_shards else "all-gather of the proposal bytes")
                ranks_seen = h.comm_ranks()
            elif ok:                                         # this rank has a communicator the others lack: start over without it
                h.close()
                h = make_handle(capi, w, device=device, rank=rank, world=world, method="tiles" if site_shards else args.method)
                h.set_state(pos, spin)
        if run is None and not site_shards and forced in ("", "rccl", "torch-nccl"):
            sharded = importlib.import_module(PKG + ".sharded")
            ok, stepper = True, None
            try:
                with stdout_to_stderr():
                    group = dist.new_group(backend="nccl")
                    stepper = sharded.ShardedStepper(sharded.HipEngine(h, torch.device("cuda", device)), group=group)
                    stepper.step(1)
                    torch.cuda.synchronize()
            except Exception as exc:                         # noqa: BLE001
                ok = False
                print(f"[rank {rank}] torch.distributed nccl unavailable ({exc})", file=sys.stderr)
            if all_agree(ok):
                path, exchange = "torch.distributed nccl all_gather_into_tensor", "torch-nccl"
                ranks_seen = dist.get_world_size(group)

                def run(n, stepper=stepper):
                    stepper.step(n)
                    torch.cuda.synchronize()
        if run is None:
            # only a rehearsal with several ranks on one GPU may fall back to host copies: with a device per rank the
            # missing RCCL path is an error, not something to time and report as the multi-GPU result
            if capi.device_count() >= world and forced != "gloo-host":
                raise SystemExit(f"bench.py: rank {rank}: no RCCL path works between {world} ranks on {capi.device_count()} devices; "
                                 "refusing to time the host-copy fallback")
            path, exchange = "host copies + torch.distributed gloo (rehearsal: several ranks share one GPU)", "gloo-host"
            ranks_seen = dist.get_world_size()
            if site_shards:
                left, right = rank - 1, rank + 1
                recv_sizes = h.halo_sizes()[1]

                def run(n):
                    for _ in range(int(n)):
                        h.propose()
                        if not h.halo_info()[2]:             # the ghost zone still covers this step: no exchange
                            h.commit()
                            continue
                        reqs, from_left, from_right = [], None, None
                        if left >= 0:
                            first = torch.from_numpy(h.halo_pack(0).copy())
                            from_left = torch.zeros(recv_sizes[1], dtype=torch.uint8)
                            reqs += [dist.isend(first, left), dist.irecv(from_left, left)]
                        if right < world:
                            last = torch.from_numpy(h.halo_pack(1).copy())
                            from_right = torch.zeros(recv_sizes[0], dtype=torch.uint8)
                            reqs += [dist.isend(last, right), dist.irecv(from_right, right)]
                        for q in reqs:
                            q.wait()
                        if from_right is not None:
                            h.halo_unpack(0, from_right.numpy())
                        if from_left is not None:
                            h.halo_unpack(1, from_left.numpy())
                        h.commit()
            else:
                _, total, off, mine = h.exchange_buffer()
                buf = torch.zeros(total, dtype=torch.uint8, device=torch.device("cuda", device))
                h.set_stream(torch.cuda.current_stream(torch.device("cuda", device)).cuda_stream)
                h.bind_exchange_buffer(buf.data_ptr(), total)
                gathered = [torch.zeros(mine, dtype=torch.uint8) for _ in range(world)]

                def run(n):
                    for _ in range(int(n)):
                        h.propose()
                        dist.all_gather(gathered, buf[off:off + mine].cpu())
                        buf.copy_(torch.cat(gathered))
                        h.commit()
                    torch.cuda.synchronize()
        run(args.warmup)
        times = []
        for _ in range(max(1, args.repeats)):                # each repeat: barrier, K steps, drain, barrier; MAX over ranks
            dist.barrier()
            t0 = time.perf_counter()
            run(args.steps)                                  # returns after the stream has drained
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.barrier()
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            times.append(float(el.item()))
        elapsed = float(np.median(times))
        extra = {"repeats": len(times), "repeats_ms_per_step": [t / args.steps * 1e3 for t in times], "graph_replay": False}
        # consistency across ranks: site shards -> every particle is owned by exactly one rank; index shards -> same state everywhere
        p, s, b, a = h.get_state()
        dist.barrier()                                       # (peer stores: no rank frees its landing buffers before all are done)
        if site_shards:
            owned = torch.tensor([int((a != 2).sum()), int(p[a == 1].astype(np.int64).sum())], dtype=torch.int64)
            dist.all_reduce(owned, op=dist.ReduceOp.SUM)
            assert int(owned[0]) == w["N"], f"ranks own {int(owned[0])} particles of {w['N']}"
        else:
            chk = torch.tensor([int(p.astype(np.int64).sum()), int((s > 0).sum())], dtype=torch.int64)
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert bool((lo == hi).all()), "ranks diverged"
        comm_path = path
        if site_shards:
            k_halo = h.halo_info()[0]
            extra.update({"halo_interval": k_halo, "halo_bytes_per_message": max(h.halo_sizes()[0])})
        sharding_note = (("site ranges of one system over %d GPU(s); every %d step(s) one message to each neighbour rank (the ghost tiles this "
                          "rank steps redundantly in between: cells, {W,S}, deposit lists)" % (world, k_halo))
                         if site_shards else "particle index over %d GPU(s), 1 all-gather of 1 B/particle per step" % world) + f" ({path})"
        n_ens_total = n_ens
    else:
        h = make_handle(capi, w, method=args.method)
        for e in range(n_ens):
            h.set_state(pos, spin, ensemble=e)
        h.step(args.warmup)                       # aps_step synchronises its stream before returning
        times = []
        for _ in range(max(1, args.repeats)):     # median of >= 5 repeats of the K timed steps
            t0 = time.perf_counter()
            h.step(args.steps)
            times.append(time.perf_counter() - t0)
        graph_steps, single_steps = h.step_info()
        loop_steps, loop_state, loop_why = h.loop_info()
        elapsed = float(np.median(times))
        extra = {"repeats": len(times), "repeats_ms_per_step": [t / args.steps * 1e3 for t in times],
                 "graph_replay": single_steps == 0 and loop_steps == 0, "steps_from_graphs": graph_steps, "steps_launched_singly": single_steps,
                 # resident loop (csrc/tile_loop.hpp): steps of a timed call taken inside ONE launch, every tile resident
                 "steps_in_resident_loop": loop_steps, "resident_loop": {1: "used", 0: "not eligible: " + loop_why, -1: "gave up: " + loop_why}.get(loop_state, "not tried")}
        hbm_copy = h.copy_bandwidth(1 << 30, 5)   # this box's streaming ceiling (read + written bytes of a 1 GiB copy)
        # per-kernel durations, timed live with HIP events on the stream the kernels are launched on
        reps = max(10, min(args.steps, 50))
        if h.method == "pairs":
            ms, launches, pairs = h.step_timed(reps)
            avg_s = ms / launches * 1e-3
            achieved = ALGO_BYTES_PER_PARTICLE_STEP * w["N"] * n_ens / avg_s / 1e9
            pairs_per_s = pairs / (ms * 1e-3)
            lds_peak_pairs = 256 * 2.4e9 / LDS_CYCLES_PER_64_PAIRS * 64      # one LDS pipe per CU
            roof = {"bound": "hbm", "bound_measured": "lds (table gather; PMC in profiles/r01_pmc_traffic.json)",
                    "kernel": "pair_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic_bytes(args.workload, "pair_accumulate"),
                    "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PARTICLE_STEP * w["N"] * n_ens,
                    "note": "the contract's HBM figure; this kernel is bound by the LDS table gather (the 0.4 MB state "
                            "lives in L2), see on_chip",
                    "on_chip": {"pairs_per_launch": pairs / launches, "pairs_per_s": pairs_per_s,
                                "lds_bound_pairs_per_s": lds_peak_pairs, "lds_frac": pairs_per_s / lds_peak_pairs,
                                "lds_cycles_per_64_pairs": LDS_CYCLES_PER_64_PAIRS,
                                "valu_issue_slots_per_pair": 4, "valu_frac": pairs_per_s * 4 / VALU_LANE_OPS_PER_S}}
        else:
            ms_fu, n_fu, deposits = h.step_timed(reps)            # deposits = field changes of the sampled steps (device counters)
            prof = h.step_profile(reps)
            # seconds per launch: start/stop events attached to each kernel's own dispatch (hipExtLaunchKernel), i.e.
            # the begin -> end interval rocprofv3 --kernel-trace reports.  APS_PROF_BRACKET=1 switches to events recorded
            # around the launch instead, which read ~2 us more per kernel (the event packets' own cost).
            kern = {k: v[0] / v[1] * 1e-3 for k, v in prof.items() if v[1]}
            ntt = h.ntt_info()
            if ntt["on"] and "tile_step" in kern:      # convolution handles step by tile_dense (csrc/tile_dense.hpp): no sweep, no lists
                kern["tile_dense"] = kern.pop("tile_step")
            if ntt["on"] and ntt["prof_launches"]:     # the field update as an exact convolution (csrc/ntt_conv.hpp): three (five) launches per step
                kern["ntt_conv"] = ntt["prof_ms"] / ntt["prof_launches"] * 1e-3
            bracketed = os.environ.get("APS_PROF_BRACKET") is not None
            dep_per_step = deposits / max(n_fu, 1)
            N_all, L_all = w["N"] * n_ens, w["L"] * n_ens
            changed = None
            if "apply" in kern:                          # accepted events of one step, counted: the states before and after compared
                before = [h.get_state(ensemble=e) for e in range(n_ens)]
                h.step(1)
                changed = float(sum(int(((x[0] != y[0]) | (x[1] != y[1]) | (x[2] != y[2]) | (x[3] != y[3])).sum())
                                    for x, y in zip(before, [h.get_state(ensemble=e) for e in range(n_ens)])))
                del before
            algo = {k: step_algo_bytes(k, N_all, L_all, w["K"], dep_per_step, bool(w.get("fp32")), changed) for k in kern}
            if "tile_dense" in algo:                   # cell words read and written, {W, S} (int2 / double2) at the particles' sites; the coefficient atomics left out
                algo["tile_dense"] = 8.0 * w["K"] * L_all + (8.0 if w.get("fp32") else 16.0) * N_all
            launches_per_step = {k: 1 for k in kern}
            if "ntt_conv" in kern:
                # per launch and prime: both signals of 2^m residues read and written (4 B each) -- the first sweep reads the deposit
                # signals instead (once for all primes), the last one reads residues and reads + writes {W, S} of the sites (int2 / double2);
                # per step on top: the table's spectrum, read by the middle launch
                M_ntt = float(1 << ntt["log2_m"]) * n_ens
                n_pr = 1 if w.get("fp32") else 2                       # binary64 field: two primes
                lps = ntt["prof_launches"] / reps
                launches_per_step["ntt_conv"] = lps
                algo["ntt_conv"] = (lps * 16.0 * M_ntt * n_pr - (n_pr - 1) * 8.0 * M_ntt - 8.0 * M_ntt * n_pr      # (the last sweep writes no residues)
                                    + 4.0 * M_ntt * n_pr + (16.0 if w.get("fp32") else 32.0) * L_all) / lps
            step_bytes = sum(algo[k] * launches_per_step[k] for k in kern)
            if loop_steps > 0:
                # the timed steps ran inside tile_loop: ONE launch = loop_steps steps; its duration from events attached to
                # that dispatch; algorithmic bytes per launch = the per-step figure x the steps the launch takes (the state
                # a streaming step would read and write stays on chip between the steps: that is the point of the kernel)
                lt = [h.step_loop_timed(args.steps) for _ in range(5)]
                assert all(n == loop_steps for _, n in lt)
                kern["tile_loop"] = float(np.median([ms for ms, _ in lt])) * 1e-3
                algo["tile_loop"] = algo["tile_step"] * loop_steps
            dom = "tile_loop" if loop_steps > 0 else max(kern, key=lambda k: kern[k] * launches_per_step[k])
            achieved = algo[dom] / kern[dom] / 1e9
            us_step = elapsed / args.steps * 1e6
            # what limits the kernel: HBM only if the algorithmic bytes move at a sizeable fraction of the measured copy rate
            if loop_steps > 0:
                limiter = ("latency of the per-step chain inside the resident loop (wait for the neighbours' records -> deposit sweep -> "
                           "proposals -> exclusion -> publish); the state never leaves the chip between steps")
            elif dom == "ntt_conv":
                limiter = ("f64 issue of the modular butterflies (%d launches of an exact number-theoretic transform, 2 x 2^%d residues each) and the "
                           "memory passes between them" % (ntt["launches"], ntt["log2_m"]))
            elif achieved >= 0.5 * hbm_copy:
                limiter = "hbm"
            elif L_all * (32 + 8 * w["K"]) < 200e6:
                limiter = "latency (cache-resident working set: dependent load -> compute -> store chain of a ~10 us launch)"
            else:
                limiter = "valu/lds issue of the deposit sweep (f64 fma + LDS table gather per deposit x 64-site row)"
            roof = {"bound": "hbm", "bound_measured": limiter, "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic_bytes(args.workload, dom, max(loop_steps, 1)),
                    "hbm_copy_GBps": hbm_copy, "frac_of_copy": achieved / hbm_copy,
                    "avg_launch_us": kern[dom] * 1e6, "algorithmic_bytes_per_launch": algo[dom],
                    "timing": "HIP events around each launch" if bracketed else "HIP start/stop events attached to each dispatch",
                    "per_kernel": {k: {"avg_launch_us": kern[k] * 1e6, "algorithmic_bytes_per_launch": algo[k],
                                       "achieved_GBps": algo[k] / kern[k] / 1e9,
                                       "traffic": measured_traffic_bytes(args.workload, k, loop_steps if k == "tile_loop" else 1)} for k in kern},
                    "whole_step": {"algorithmic_bytes": step_bytes, "us_per_step": us_step,
                                   "achieved_GBps": step_bytes / (us_step * 1e-6) / 1e9,
                                   "frac": step_bytes / (us_step * 1e-6) / 1e9 / HBM_PEAK_GBS},
                    "note": ("algorithmic bytes = what a step must read and write when the state streams through HBM ((32 + 8K) B per site + the deposits); "
                             "inside the resident loop the state stays in LDS between steps and only the records travel (counter traffic about a tenth, "
                             "profiles/r02_loop_config2_pmc.json): the fraction says how fast the steps go, not how busy HBM is") if loop_steps > 0 else
                            "algorithmic bytes = (32 + 8K) B per site + the deposits, per step",
                    "deposits_per_step": dep_per_step, "changed_particles_per_step": changed, "kernels_per_step": sum(launches_per_step.values()) if loop_steps == 0 else 1.0 / loop_steps,
                    "launches_per_step": launches_per_step,
                    "steps_per_launch": loop_steps if loop_steps > 0 else 1}
        roof = dict(roof or {}, **{"hbm_copy_GBps": hbm_copy})
    if not sharded_path:
        n_ens_total, sharding_note = n_ens, "one GPU"
    p, s, b, a = h.get_state()
    assert (a != 0).all() and np.bincount(p[a == 1], minlength=w["L"]).max() <= w["K"]
    h_method = h.method
    h.close()
    if rank != 0:
        return
    out = {
        "metric": "particle-steps/sec at N=1e5", "value": w["N"] * n_ens_total * args.steps / elapsed, "unit": "particle-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "i32 (fixed point 2^-q, the fp32 mode of aps_params)" if w.get("fp32") else "f64", "data": "synthetic",
        "config": {"workload": ("BASELINE config 2: N=100000 particles, L=200000 sites, K=1, reflecting walls, "
                                "sigma=0.005 (4001-tap table), beta=0.7, dt=0.0125, exclusion on") if args.workload == "config2"
                               else f"{'BASELINE ' if args.workload.startswith('config') else ''}{args.workload}: N={w['N']} x {n_ens} ensemble(s), L={w['L']}, K=1, sigma_g={w['sigma'] * w['L']:.0f} sites, dt=0.0125",
                   "method": h_method,
                   "sharding": sharding_note},
    }
    if sharded_path:
        out["exchange"], out["ranks_seen"] = exchange, ranks_seen
    out.update(extra)
    if roof:
        out["roofline"] = roof
    if world == 1 and not args.no_cpu_baseline and args.workload == "config2":
        out["cpu_baseline"] = cpu_baseline(w)
        out["cpu_reference_loop"] = cpu_reference_loop(w)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
