#!/usr/bin/env python
"""Headline benchmark: frames/s of RDF + window MSD on a ~10k-atom ZIF-4 trajectory.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input:
``Rdf.from_trajectory(traj)`` (dr = 0.01, rmax = half cell) followed by
``WindowMsd.from_trajectory(traj, delta_time=100, timestep=1)`` on a trajectory
that is already resident in HBM (BASELINE.json configs[2], the headline
config: 3x3x4 ZIF-4 supercell = 9792 atoms, 5000 frames, Gaussian random walk
sigma = 0.05 A/frame/axis wrapped into a constant orthorhombic cell).

N > 1 (weak scaling): every rank owns its own block of 5000 frames (frames
shard embarrassingly); the only data-path collective is the RCCL all-reduce of
the integer RDF histograms at the end of each step.  MSD couples frames, so
each rank's block is its own time series (no collective).

Rank 0 prints ONE JSON line (see the keys below).  ``roofline`` is computed for
the dominant kernel (the RDF tile kernel) from HIP events recorded inside the
library on the launching stream; ``cpu_baseline`` times the CPU oracle on a
bounded sample on this box's host cores (N = 1 only).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK = 78.6e12    # flop/s, vector FP64 (half the 157.3 TF FP32 vector rate)


def make_trajectory(device, reps, n_frames, sigma, seed):
    """Synthetic trajectory generated directly in HBM (torch), float64."""
    import torch
    from amof_amd.frames import PackedTrajectory
    from amof_amd.io import read_extxyz
    base = read_extxyz(os.path.join(ROOT, "tests", "golden", "ZIF-4.xyz"), 0)
    pos, num = [], []
    for a in range(reps[0]):
        for b in range(reps[1]):
            for c in range(reps[2]):
                pos.append(base.positions + a * base.cell[0] + b * base.cell[1] + c * base.cell[2])
                num.append(base.numbers)
    pos0 = np.concatenate(pos)
    numbers = np.concatenate(num)
    lengths = np.diag(base.cell) * np.array(reps)          # constant orthorhombic cell
    cell = np.diag(lengths)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = len(numbers)
    L = torch.tensor(lengths, dtype=torch.float64, device=device)
    traj = torch.empty((n_frames, n, 3), dtype=torch.float64, device=device)
    cur = torch.tensor(pos0, dtype=torch.float64, device=device)   # unwrapped position of the last frame
    chunk = 250
    for f0 in range(0, n_frames, chunk):
        f1 = min(f0 + chunk, n_frames)
        steps = torch.randn((f1 - f0, n, 3), dtype=torch.float64, device=device, generator=g) * sigma
        if f0 == 0:
            steps[0] = 0.0                                  # frame 0 is the base structure
        walk = cur + torch.cumsum(steps, dim=0)
        cur = walk[-1].clone()
        traj[f0:f1] = walk - torch.floor(walk / L) * L     # wrapped into the cell
        del steps, walk
    return PackedTrajectory(traj, cell, numbers)


def cpu_baseline(packed, rmax, nbins, window, rdf_frames):
    """CPU oracle ("port") on a bounded sample, single thread (the reference is
    single-threaded on this path: OMP_NUM_THREADS=1, serial frame loop,
    amof/rdf.py:6,88-93; parallel=False default in amof/msd.py:157)."""
    from oracle import clib, numpy_oracle as no
    from tests import helpers as H
    import torch
    F, N = packed.n_frames, packed.n_atoms
    kinds, sp = H.species_of(packed.numbers)
    pick = np.linspace(0, F - 1, rdf_frames).astype(int)
    pos_s = packed.pos[torch.as_tensor(pick, device=packed.pos.device)].cpu().numpy()
    t0 = time.perf_counter()
    clib.rdf_hist(pos_s, packed.cell, sp, len(kinds), rmax, nbins, cell_list=True)
    t_rdf = (time.perf_counter() - t0) / rdf_frames
    # MSD: the reference's O(W*F) numpy loop structure on the full trajectory for a
    # subset of the atoms (the loops are per element and linear in atoms)
    sub = np.zeros(N, dtype=bool)
    for z in set(packed.numbers.tolist()):
        sub[np.nonzero(packed.numbers == z)[0][::8]] = True     # every 8th atom of each element
    pos_h = packed.pos.cpu().numpy()
    t0 = time.perf_counter()
    no.window_msd(pos_h, packed.cell, packed.numbers, packed.masses, window, atom_subset=sub)
    t_msd_sub = time.perf_counter() - t0
    del pos_h
    t_msd = t_msd_sub * (N / float(sub.sum()))
    fps = F / (t_rdf * F + t_msd)
    # the same RDF sample frame-parallel on every host core this process may use (the reference's
    # parallel=True mode is frame-parallel too, amof/bad.py:151); ctypes releases the GIL
    from concurrent.futures import ThreadPoolExecutor
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                     # container CPU quota (cgroup v2), e.g. "1600000 100000"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    cores = min(cores, 16)                   # one GPU's share of the host
    per = 2
    pick2 = np.linspace(0, F - 1, per * cores).astype(int)
    pos_p = packed.pos[torch.as_tensor(pick2, device=packed.pos.device)].cpu().numpy()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda k: clib.rdf_hist(pos_p[per * k:per * (k + 1)], packed.cell, sp, len(kinds), rmax, nbins,
                                            cell_list=True), range(cores)))
    t_rdf_par = (time.perf_counter() - t0) / (per * cores)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
        "all_cores": {"value": F / (t_rdf_par * F + t_msd), "unit": "frames/s", "cores": cores,
                      "sample": "RDF: %d frames on %d threads, %.4f s/frame aggregate; MSD as above (single thread)"
                                % (per * cores, cores, t_rdf_par)},
        "cpu_model": cpu_model,
        "sample": "RDF: C oracle (cell list) on %d of %d frames, %.3f s/frame; MSD: numpy restatement of the "
                  "reference loops on all %d frames for 1/8 of the atoms (%.1f s), scaled x%.1f to all atoms"
                  % (rdf_frames, F, t_rdf, F, t_msd_sub, N / float(sub.sum())),
        "rdf_s_per_frame": t_rdf, "msd_s_full_est": t_msd,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=5000)
    ap.add_argument("--reps", type=str, default="3,3,4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rdf-frames", type=int, default=24)
    ap.add_argument("--with-bad", action="store_true", help="BASELINE configs[3]: add Bad({'Zn-N': 2.5}, dtheta=0.05) to the step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for rehearsals")
    ap.add_argument("--all-on-device", type=int, default=None, help="rehearsal: put every rank on this GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from amof_amd import _hip
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if args.all_on_device is not None:
        local_rank = args.all_on_device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    reps = tuple(int(x) for x in args.reps.split(","))
    F = args.frames
    packed = make_trajectory(device, reps, F, 0.05, 20261003 + rank)
    N = packed.n_atoms
    ctx = _hip.get_context(local_rank)
    mode = 'local' if world > 1 else False

    def step():
        rdf = Rdf.from_trajectory(packed, device=local_rank, distributed=mode)
        t_rdf = ctx.last_kernel_seconds(dominant=True)
        t_rdf_all = ctx.last_kernel_seconds(dominant=False)
        msd = WindowMsd.from_trajectory(packed, delta_time=100, timestep=1, device=local_rank, distributed=mode)
        t_msd_dom = ctx.last_kernel_seconds(dominant=True)
        t_msd_all = ctx.last_kernel_seconds(dominant=False)
        if args.with_bad:
            from amof_amd.bad import Bad
            bad = Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05, device=local_rank, distributed=mode)
            k_bad.append(ctx.last_kernel_seconds(dominant=False))
            assert "N-Zn-N" in bad.data.columns
        return rdf, msd, t_rdf, t_msd_dom, t_msd_all, t_rdf_all

    k_bad = []
    for _ in range(args.warmup):
        step()
    k_bad.clear()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    k_rdf, k_msd_dom, k_msd_all, k_rdf_all = [], [], [], []
    for _ in range(args.steps):
        rdf, msd, a, b, c, d = step()
        k_rdf.append(a); k_msd_dom.append(b); k_msd_all.append(c); k_rdf_all.append(d)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = world * F * args.steps
        fps = total_frames / elapsed
        rmax, nbins = rdf.rmax, len(rdf.data)
        t_rdf = float(np.mean(k_rdf))
        alg_bytes = F * (24 * N + 72)                      # 24N+72 bytes per frame per pass
        pairs = F * N * (N - 1) / 2.0                      # unordered pair evaluations per launch
        in_range = float(rdf.hist.sum()) / 2.0 / world if world > 1 else float(rdf.hist.sum()) / 2.0
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, gfx950 corrections applied),
        # recorded under profiles/ for this exact workload; null for any other size
        traffic = {}
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile) and (N, F) == (9792, 5000):
            with open(tfile) as fh:
                traffic = json.load(fh).get("cfg3", {})
        out = {
            "metric": "frames/s (RDF+MSD, 10k-atom ZIF-4)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2] headline: %d-atom ZIF-4 %dx%dx%d supercell, %d frames per GPU, "
                                   "Rdf(dr=0.01, rmax=half_cell -> %.4f A, %d bins) + WindowMsd(delta_time=100, W=%d)"
                                   % (N, reps[0], reps[1], reps[2], F, rmax, nbins, len(msd.data)),
                       "n_atoms": N, "frames_per_gpu": F, "rdf_bins": nbins, "msd_windows": len(msd.data),
                       "parallelism": "frames x%d, RCCL all-reduce of u64 histograms" % world},
            "roofline": {"kernel": "rdf_tile_kernel_fast", "bound": "hbm", "achieved": alg_bytes / t_rdf / 1e9,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_bytes / t_rdf / 1e9 / HBM_PEAK_GBPS,
                         "traffic": traffic.get("rdf_tile_kernel_fast"), "launch_seconds": t_rdf,
                         "algorithmic_bytes": alg_bytes,
                         "note": "all-pairs RDF at half-cell rmax does N(N-1)/2 = 4.8e7 pair evaluations per 235 kB "
                                 "frame: VALU-issue bound, not HBM bound (SURVEY 8d, DESIGN 4.1); the honest HBM "
                                 "fraction is tiny by construction -- see pair_evals_per_s"},
            "pair_evals_per_s": pairs / t_rdf, "pairs_in_range_per_s": in_range / t_rdf,
            # supplementary, the bound that actually applies: SIMD cycles spent per wave-level
            # pair (64 pairs) at the nominal 2.4 GHz, all N(N-1)/2 pairs counted although ~1/3 are culled;
            # the instruction mix of the inner loop sums to ~70 issue cycles per visited wave-pair
            "valu_issue": {"simd_cycles_per_wave_pair": t_rdf * 1024 * 2.4e9 / (pairs / 64.0),
                           "model_cycles_per_visited_wave_pair": 70.0, "visited_fraction": 0.67},
            "roofline_msd": {"kernel": "msd pipeline (com + delta_transpose + msd_comb + reduce)", "bound": "hbm",
                             "achieved": alg_bytes / float(np.mean(k_msd_all)) / 1e9, "peak": HBM_PEAK_GBPS,
                             "unit": "GB/s", "frac": alg_bytes / float(np.mean(k_msd_all)) / 1e9 / HBM_PEAK_GBPS,
                             "traffic": traffic.get("msd_pipeline"), "pipeline_seconds": float(np.mean(k_msd_all)),
                             "msd_window_kernel_seconds": float(np.mean(k_msd_dom))},
            "kernel_seconds_per_step": {"rdf_tile": t_rdf, "rdf_all_incl_quantize": float(np.mean(k_rdf_all)),
                                        "msd_all": float(np.mean(k_msd_all))},
        }
        if args.with_bad:
            out["metric"] = "frames/s (RDF+BAD+MSD, 10k-atom ZIF-4)"
            out["config"]["workload"] += " + Bad({'Zn-N': 2.5}, dtheta=0.05) [configs[3]]"
            out["kernel_seconds_per_step"]["bad_all"] = float(np.mean(k_bad))
            args.no_cpu_baseline = True         # the CPU leg times RDF+MSD only
        if world == 1 and not args.no_cpu_baseline:
            window = np.arange(0, (F // 2), 100)
            out["cpu_baseline"] = cpu_baseline(packed, rmax, nbins, window, args.cpu_rdf_frames)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
            out["speedup_vs_cpu_1core"] = fps / out["cpu_baseline"]["value"]
            # supplementary (never `value`): the same step from a host-resident packed trajectory, i.e. including
            # the 24*N*F-byte PCIe staging of both passes
            from amof_amd.frames import PackedTrajectory
            host = PackedTrajectory(packed.pos.cpu().numpy(), packed.cell, packed.numbers)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Rdf.from_trajectory(host, device=local_rank, distributed=False)
            WindowMsd.from_trajectory(host, delta_time=100, timestep=1, device=local_rank, distributed=False)
            torch.cuda.synchronize()
            out["host_resident_frames_per_s"] = F / (time.perf_counter() - t0)
            # supplementary: packing a list of ase.Atoms-like frames into the arrays above (pure Python + memcpy,
            # identical for a CPU and a GPU path; SURVEY 8d asks for it separately)
            from amof_amd.frames import Frame, pack_trajectory
            nfr = min(F, 200)
            frames = [Frame(host.numbers, host.pos[k], host.cell_of(k)) for k in range(nfr)]
            t0 = time.perf_counter()
            pack_trajectory(frames)
            out["pack_atoms_list_frames_per_s"] = nfr / (time.perf_counter() - t0)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
