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

N > 1 -- STRONG scaling of the same job: every rank
generates the same 5000-frame trajectory (same seed) in its own HBM; RDF (and
BAD / CN) shard the FRAMES, one RCCL all-reduce of the integer histograms that
never leave HBM in between; MSD shards the ATOMS, one all-reduce of the S x W
float64 sums.  ``value`` = 5000 frames / step time, whatever N is.
``--scaling weak`` instead gives every rank its own 5000-frame block.

After the RDF+MSD steps the same K steps are repeated with
``Bad({'Zn-N': 2.5}, dtheta=0.05)`` added (BASELINE configs[3]) and reported
under ``configs3`` -- never in ``value``, so that the metric means the same
work at every N.

Rank 0 prints ONE JSON line.  ``roofline`` is computed for the dominant kernel
(the RDF tile kernel) from HIP events recorded inside the library on the
launching stream; ``cpu_baseline`` times the CPU oracle on a bounded sample on
this box's host cores (N = 1 only); ``verified`` says whether the timed results
themselves were tied to the oracle after the timed region (frames sampled by
leave-one-out for RDF, an atom slice for MSD).
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
CLOCK_HZ = 2.4e9            # nominal engine clock
N_SIMD = 1024               # 256 CUs x 4 SIMDs


def sources_stamp():
    """sha256 over the kernel sources (the same files profiles/tools/pmc_to_json.py stamps profiles/traffic.json and
    profiles/valu_model.json with): counter-derived annotations are only quoted for the kernels they were collected on"""
    import hashlib
    h = hashlib.sha256()
    for fn in ("rdf.hip", "msd.hip", "quant.hip", "amof_internal.h", "guard_math.h"):
        with open(os.path.join(ROOT, "amof_amd", "csrc", fn), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def make_trajectory(device, reps, n_frames, sigma, seed):
    """Synthetic trajectory generated directly in HBM (torch), float64."""
    from tests import helpers as H
    return H.device_walk(device, reps, n_frames, sigma, seed)


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
    cores_usable = cores                     # what the process may use (affinity, cgroup quota)
    if os.environ.get("AMOF_BENCH_CPU_THREADS"):
        cores = max(1, min(cores, int(os.environ["AMOF_BENCH_CPU_THREADS"])))
    per = 2 if cores <= 32 else 1            # (bounded sample: about the same CPU seconds whatever the core count)
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
        "all_cores": {"value": F / (t_rdf_par * F + t_msd), "unit": "frames/s", "cores": cores, "cores_usable": cores_usable,
                      "sample": "RDF: %d frames on %d threads, %.4f s/frame aggregate; MSD as above (single thread); "
                                "%d threads = what this process may use (affinity and cgroup CPU quota) of the host's %d CPUs"
                                % (per * cores, cores, t_rdf_par, cores_usable, os.cpu_count() or 0)},
        "cpu_model": cpu_model,
        "sample": "RDF: C oracle (cell list) on %d of %d frames, %.3f s/frame; MSD: numpy restatement of the "
                  "reference loops on all %d frames for 1/8 of the atoms (%.1f s), scaled x%.1f to all atoms"
                  % (rdf_frames, F, t_rdf, F, t_msd_sub, N / float(sub.sum())),
        "rdf_s_per_frame": t_rdf, "msd_s_full_est": t_msd,
    }


def verify(packed, ctx, rdf, msd, frames):
    """Tie the TIMED results to the oracle (run after the timed region, on rank 0, which holds the whole
    trajectory in strong-scaling mode).  RDF: for sampled frames k,  H[0,k) + oracle(k) + H[k+1,F) must equal the
    timed histogram bit for bit (the two big partial launches run at the bench's own launch geometry).  MSD: the
    GPU sums of the first 272-atom replica equal the numpy restatement (rtol 1e-9) and the atom slices of the whole
    system add up to the timed sums."""
    from oracle import clib, numpy_oracle as no
    from tests import helpers as H
    import torch
    F, N = packed.n_frames, packed.n_atoms
    kinds, sp = H.species_of(packed.numbers)
    nb = len(rdf.data)
    out = {"rdf_frames": [int(k) for k in frames], "rdf": True, "msd": True}
    pos_s = packed.pos[torch.as_tensor(np.asarray(frames), device=packed.pos.device)].cpu().numpy()
    timed = np.asarray(rdf.hist).view(np.uint64)
    for q, k in enumerate(frames):
        h_cpu, _ = clib.rdf_hist(pos_s[q:q + 1], packed.cell, sp, len(kinds), rdf.rmax, nb, cell_list=True)
        acc = h_cpu.copy()
        if k > 0:
            acc += ctx.rdf_accumulate(packed, rdf.rmax, nb, frame_range=(0, k))[0]
        if k + 1 < F:
            acc += ctx.rdf_accumulate(packed, rdf.rmax, nb, frame_range=(k + 1, F))[0]
        out["rdf"] = out["rdf"] and bool(np.array_equal(acc, timed))
    window = np.asarray(msd.data["Time"].values, dtype=np.int64)       # timestep = 1
    a1 = min(272, N)
    mask = np.zeros(N, dtype=bool)
    mask[:a1] = True
    pos_h = packed.pos.cpu().numpy()
    elements, ref = no.window_msd_fast(pos_h, packed.cell, packed.numbers, packed.masses, window, atom_subset=mask)
    del pos_h
    sumsq, k2 = ctx.msd_window(packed, window, atom_range=(0, a1))
    worst = 0.0
    for e, r in zip(elements, ref):
        n_e = int((packed.numbers[:a1] == e).sum())
        got = sumsq[k2.index(int(e))] / n_e / (F - window)
        worst = max(worst, float(np.max(np.abs(got - r) / np.maximum(np.abs(r), 1e-300))))
    parts = sumsq.copy()
    step = max(1, (N - a1 + 2) // 3)
    for b in range(a1, N, step):
        parts += ctx.msd_window(packed, window, atom_range=(b, min(b + step, N)))[0]
    dev = float(np.max(np.abs(parts - msd.sumsq) / np.maximum(np.abs(msd.sumsq), 1e-300)))
    out["msd_slice_max_rel_dev"] = worst
    out["msd_slices_vs_timed_max_rel_dev"] = dev
    out["msd"] = bool(worst < 1e-9 and dev < 1e-11)
    out["ok"] = bool(out["rdf"] and out["msd"])
    return out


def verify_rdf_timed(ctx, packed, rmax, nb, timed_hist, frames):
    """Tie a TIMED RDF histogram to the oracle: for every sampled frame k,  H[0,k) + oracle(k) + H[k+1,F) must equal
    the timed histogram bit for bit (the partial launches run at the leg's own launch geometry)."""
    from oracle import clib
    from tests import helpers as H
    import torch
    F = packed.n_frames
    kinds, sp = H.species_of(packed.numbers)
    pos_s = packed.pos[torch.as_tensor(np.asarray(frames), device=packed.pos.device)].cpu().numpy()
    timed = np.asarray(timed_hist).view(np.uint64)
    ok = True
    for q, k in enumerate(frames):
        cell_k = packed.cell if packed.cell.shape[0] == 1 else packed.cell[k:k + 1]
        acc, _ = clib.rdf_hist(pos_s[q:q + 1], cell_k, sp, len(kinds), rmax, nb, cell_list=True)
        if k > 0:
            acc = acc + ctx.rdf_accumulate(packed, rmax, nb, frame_range=(0, k))[0]
        if k + 1 < F:
            acc = acc + ctx.rdf_accumulate(packed, rmax, nb, frame_range=(k + 1, F))[0]
        ok = ok and bool(np.array_equal(acc, timed))
    return ok


def supplementary(device, local_rank, ctx, do_verify=True):
    """The other single-GPU configs of BASELINE.json, timed by the same process (never `value`); each leg's timed
    result is tied to the oracle afterwards (`verified`), like the headline's."""
    import torch
    from amof_amd.rdf import Rdf
    from amof_amd.cn import CoordinationNumber
    from amof_amd.frames import PackedTrajectory
    from tests import helpers as H
    out = {}
    # configs[1]: 2x2x2 = 2176 atoms, 1000 frames, partial Zn-N RDF with rmax = 10 A + CoordinationNumber
    p1 = make_trajectory(device, (2, 2, 2), 1000, 0.05, 11)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        r1 = Rdf.from_trajectory(p1, dr=0.01, rmax=10.0, device=local_rank, distributed=False).result()
        t_r = time.perf_counter() - t0
        k_r, path_r = r1._stats["kernel_s_all"], r1._stats["path"]
        t0 = time.perf_counter()
        c1 = CoordinationNumber.from_trajectory(p1, {'Zn-N': 2.5}, device=local_rank, distributed=False).result()
        t_c = time.perf_counter() - t0
        k_c, path_c = c1._stats["kernel_s_all"], c1._stats["path"]
    alg1 = p1.n_frames * (24 * p1.n_atoms + 72)
    pairs1 = float(np.asarray(r1.hist).sum()) / 2.0
    out["configs1"] = {
        "workload": "configs[1]: %d-atom ZIF-4 2x2x2, %d frames, Rdf(dr=0.01, rmax=10 -> %d bins; column 'Zn-N') + "
                    "CoordinationNumber({'Zn-N': 2.5})" % (p1.n_atoms, p1.n_frames, len(r1.data)),
        "frames_per_s": p1.n_frames / (t_r + t_c), "rdf_wall_s": t_r, "cn_wall_s": t_c,
        "rdf_kernel_s": k_r, "cn_kernel_s": k_c, "rdf_path": path_r, "cn_path": path_c,
        "pairs_in_range_per_s": pairs1 / k_r,
        "roofline_rdf": {"bound": "hbm", "achieved": alg1 / k_r / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": alg1 / k_r / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": alg1},
        "roofline_cn": {"bound": "hbm", "achieved": alg1 / k_c / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": alg1 / k_c / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": alg1},
        "cn_first_frame": float(c1.data['Zn-N'].values[0]),
    }
    if do_verify:
        try:
            from oracle import clib
            from amof_amd import atom as amatom
            ok_rdf = verify_rdf_timed(ctx, p1, r1.rmax, len(r1.data), r1.hist, [0, 499, 999])
            kinds1, sp1 = H.species_of(p1.numbers)
            rcm = amatom.cutoff_matrix(amatom.format_cutoff({'Zn-N': 2.5}), kinds1)
            zn, n = kinds1.index(30), kinds1.index(7)
            pick = [0, 333, 999]
            pos_c = p1.pos[torch.as_tensor(pick, device=p1.pos.device)].cpu().numpy()
            sums = clib.cn_counts(pos_c, p1.cell, sp1, len(kinds1), rcm, [(zn, n)])
            n_zn = int((p1.numbers == 30).sum())
            ok_cn = bool(np.array_equal(sums[:, 0] / n_zn, c1.data['Zn-N'].values[pick]))
            out["configs1"]["verification"] = {"rdf_frames": [0, 499, 999], "rdf": ok_rdf, "cn_frames": pick, "cn": ok_cn}
            out["configs1"]["verified"] = bool(ok_rdf and ok_cn)
        except Exception as exc:
            out["configs1"]["verification"] = {"error": repr(exc)}
            out["configs1"]["verified"] = False
    del p1, r1, c1
    # configs[4]: 7x7x8 = 106 624 atoms in a sheared cell, 2000 frames (5.1 GB), cell-list RDF at rmax = 10 A
    base = H.replicate(H.zif4_frame(), (7, 7, 8))
    shear = np.eye(3) + np.array([[0, 0.15, 0.10], [0, 0, 0.20], [0, 0, 0]])
    cell = base.cell @ shear
    F4 = 2000
    g = torch.Generator(device=device)
    g.manual_seed(44)
    frac0 = torch.tensor(np.linalg.solve(base.cell.T, base.positions.T).T, dtype=torch.float64, device=device)
    C = torch.tensor(cell, dtype=torch.float64, device=device)
    Cinv = torch.linalg.inv(C)
    pos4 = torch.empty((F4, len(base.numbers), 3), dtype=torch.float64, device=device)
    cur = frac0 @ C
    for f0 in range(0, F4, 50):
        f1 = min(f0 + 50, F4)
        steps = torch.randn((f1 - f0, len(base.numbers), 3), dtype=torch.float64, device=device, generator=g) * 0.05
        if f0 == 0:
            steps[0] = 0.0
        walk = cur + torch.cumsum(steps, dim=0)
        cur = walk[-1].clone()
        s = walk @ Cinv
        pos4[f0:f1] = (s - torch.floor(s)) @ C
        del steps, walk, s
    p4 = PackedTrajectory(pos4, cell, base.numbers)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        r4 = Rdf.from_trajectory(p4, dr=0.01, rmax=10.0, device=local_rank, distributed=False).result()
        t4 = time.perf_counter() - t0
        k4_dom, k4_all, path4 = r4._stats["kernel_s_dominant"], r4._stats["kernel_s_all"], r4._stats["path"]
    alg4 = F4 * (24 * p4.n_atoms + 72)
    out["configs4"] = {
        "workload": "configs[4]: %d-atom sheared (triclinic) ZIF-4 7x7x8, %d frames, Rdf(dr=0.01, rmax=10 -> %d bins)"
                    % (p4.n_atoms, F4, len(r4.data)),
        "frames_per_s": F4 / t4, "wall_s": t4, "kernel_s_all": k4_all, "kernel_s_dominant": k4_dom, "path": path4,
        "pairs_in_range_per_s": float(np.asarray(r4.hist).sum()) / 2.0 / k4_dom,
        "roofline": {"kernel": path4, "bound": "hbm", "achieved": alg4 / k4_dom / 1e9, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": alg4 / k4_dom / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": alg4,
                     "incl_cell_sort_frac": alg4 / k4_all / 1e9 / HBM_PEAK_GBPS},
        "kernel_launches": int(r4._stats["kernel_launches"]),
    }
    if do_verify:
        try:
            # (the partial launches [0, k) and [k+1, F) cut the frame batches -- 1 GiB of sorted scratch each -- elsewhere
            #  than the timed launch did: a batch-boundary error would break the identity; tests/test_gpu_large.py
            #  pins the frames either side of a boundary explicitly)
            frames4 = [0, F4 // 3 + 1, F4 - 1]
            ok4 = verify_rdf_timed(ctx, p4, r4.rmax, len(r4.data), r4.hist, frames4)
            out["configs4"]["verification"] = {"rdf_frames": frames4, "rdf": ok4}
            out["configs4"]["verified"] = bool(ok4)
        except Exception as exc:
            out["configs4"]["verification"] = {"error": repr(exc)}
            out["configs4"]["verified"] = False
    del p4, r4, pos4
    torch.cuda.empty_cache()
    # from_text: the headline system from an XYZ text file (53 bytes per atom line) -- read at once, then analysed (the
    # reference's order: amof/trajectory.py:37-60 -> amof/rdf.py:88-93) versus streamed in frame batches whose parse
    # overlaps the previous batch's analysis (amof_amd.stream.XyzStream)
    try:
        import tempfile
        import pandas as pd
        from amof_amd import trajectory as T
        from amof_amd import data as eldata
        from amof_amd.stream import XyzStream
        Ft = 400
        pt = make_trajectory(device, (3, 3, 4), Ft, 0.05, 20261003)
        host = pt.pos.cpu().numpy()
        syms = np.array([eldata.chemical_symbols[int(z)] for z in pt.numbers])
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
            path = os.path.join(tmp, "headline_%d.xyz" % Ft)
            with open(path, "w") as fh:
                for k in range(Ft):
                    fh.write("%d\nframe %d\n" % (pt.n_atoms, k))
                    pd.DataFrame({"s": syms, "x": host[k, :, 0], "y": host[k, :, 1], "z": host[k, :, 2]}).to_csv(
                        fh, sep=" ", header=False, index=False, float_format="%.10f")
            size = os.path.getsize(path)
            cell = pt.cell[0]
            res = {}
            for rep in range(2):
                t0 = time.perf_counter()
                whole = T.read_lammps_traj(path, ":", cell=cell)
                t_parse = time.perf_counter() - t0
                r_a = Rdf.from_trajectory(whole, device=local_rank, distributed=False).result()
                t_serial = time.perf_counter() - t0
                t0 = time.perf_counter()
                r_b = Rdf.from_trajectory(XyzStream(path, cell=cell, batch_frames=50), device=local_rank, distributed=False).result()
                t_stream = time.perf_counter() - t0
            same = bool(np.array_equal(np.asarray(r_a.hist), np.asarray(r_b.hist)) and r_a.data.equals(r_b.data))
            ok_t = None
            if do_verify:
                pd_dev = PackedTrajectory(torch.as_tensor(whole.pos).to(device), whole.cell, whole.numbers)
                ok_t = verify_rdf_timed(ctx, pd_dev, r_b.rmax, len(r_b.data), r_b.hist, [0, Ft - 1])
                del pd_dev
            out["from_text"] = {
                "workload": "headline system from a %d-frame XYZ text file (%.0f MB): Rdf(dr=0.01, half cell)" % (Ft, size / 1e6),
                "read_then_analyse_frames_per_s": Ft / t_serial, "streamed_frames_per_s": Ft / t_stream,
                "parse_only_frames_per_s": Ft / t_parse, "parse_GB_per_s": size / t_parse / 1e9, "file_bytes": size,
                "parse_note": "rate on THIS %.0f MB file (page cache warm, /dev/shm); DESIGN 4.4 quotes 6.6 GB/s for a 2.6 GB file "
                              "-- thread start-up and the frame index weigh more on a small one" % (size / 1e6),
                "host_threads": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
                "streamed_equals_whole": same, "verified": (bool(ok_t and same) if do_verify else None)}
        del pt, host, whole
    except Exception as exc:
        out["from_text"] = {"error": repr(exc)}
    torch.cuda.empty_cache()
    return out


def realistic_cells(device, local_rank, ctx, head, do_verify=True, frames=400):
    """Cells the reference's users actually have (amof/rdf.py:74-79 takes half the shortest cell LENGTH over all frames:
    written for changing, non-rectangular cells), timed by the same process and tied to the oracle like every other leg:

    * fixture: the TRUE lattice of examples/files/ZIF-4.xyz (off-diagonals -1.9e-4 .. -5.8e-4 A) replicated 3x3x4 -- the
      headline system without the bench's exactly-diagonal cell;
    * npt: a near-cubic 4x4x3 supercell (13 056 atoms) sheared by 2 %, a different (breathing) cell every frame;
    * hexagonal: the 3x3x4 system mapped affinely into a hexagonal cell (a = b, gamma = 120 degrees) of the same volume.

    Every leg runs Rdf.from_trajectory with the reference's defaults (dr = 0.01, rmax = half the shortest length).  `head`
    = (kernel seconds, frames, atoms, visited fraction) of the headline launch: `vs_headline_per_visited_pair` is this leg's
    kernel time per VISITED pair evaluation over the headline's (both kernels cull by slabs along the longest axis: the
    visited share is the geometric one, min(1, 2 rmax / L_slab + 3/256)); `vs_diagonal_same_frames_per_visited_pair`
    compares with the diagonal headline system run over the legs' own number of frames (`diagonal_reference`)."""
    import torch
    from amof_amd.rdf import Rdf
    from amof_amd.frames import Frame
    from tests import helpers as H
    t_head, f_head, n_head, vis_head = head
    head_cost = t_head / (f_head * n_head * (n_head - 1) / 2.0 * vis_head)
    out = {}
    zif = H.zif4_frame()
    legs = []
    base = H.replicate(zif, (3, 3, 4))
    # the yardstick at the legs' own launch size: the headline system (exactly diagonal cell) over `frames` frames -- a
    # 400-frame launch runs a few per cent below the 5000-frame one (ramp-up, tails), which is not the cells' doing
    same_cost = None
    try:
        ref = H.device_walk(device, (3, 3, 4), frames, 0.05, 20261003)
        torch.cuda.synchronize()
        best = None
        for rep in range(3):
            r = Rdf.from_trajectory(ref, device=local_rank, distributed=False).result()
            k = r._stats["kernel_s_dominant"]
            best = k if best is None or k < best else best
        lz = float(np.max(ref.cell_lengths()))
        vis = min(1.0, 2.0 * r.rmax / lz + 3.0 / 256.0) if 2.0 * r.rmax * 1.05 < lz else 1.0
        same_cost = best / (frames * ref.n_atoms * (ref.n_atoms - 1) / 2.0 * vis)
        out["diagonal_reference"] = {"workload": "the headline system (diagonal cell) over the legs' %d frames" % frames,
                                     "path": r._stats["path"], "kernel_ms_per_frame": 1e3 * best / frames,
                                     "visited_fraction_geometric": vis}
        del ref, r
    except Exception as exc:
        out["diagonal_reference"] = {"error": repr(exc)}
    legs.append(("fixture", "true ZIF-4.xyz lattice (off-diagonals kept) x 3x3x4, constant cell", base, base.cell, 20261004))
    cub = H.replicate(zif, (4, 4, 3))
    cub = Frame(cub.numbers, cub.positions, np.diag(np.diag(cub.cell)))
    shear = np.eye(3) + np.array([[0, 0.02, 0.01], [0, 0, 0.02], [0, 0, 0]])
    rng = np.random.default_rng(9)
    npt_cells = np.array([(cub.cell @ shear) * (1.0 + 0.005 * rng.normal()) for _ in range(frames)])
    legs.append(("npt", "near-cubic 4x4x3 (13 056 atoms), 2 % shear, a breathing cell per frame (sigma 0.5 %)",
                 Frame(cub.numbers, cub.positions @ shear, cub.cell @ shear), npt_cells, 20261005))
    d3 = np.diag(np.diag(base.cell))
    a_hex = float(np.sqrt(d3[0, 0] * d3[1, 1] / (np.sqrt(3.0) / 2.0)))          # same volume, same c
    hexc = np.array([[a_hex, 0.0, 0.0], [-0.5 * a_hex, np.sqrt(3.0) / 2.0 * a_hex, 0.0], [0.0, 0.0, d3[2, 2]]])
    frac = np.linalg.solve(np.asarray(base.cell).T, base.positions.T).T
    legs.append(("hexagonal", "3x3x4 system mapped into a hexagonal cell (a = b = %.3f A, gamma = 120), constant cell" % a_hex,
                 Frame(base.numbers, frac @ hexc, hexc), hexc, 20261006))
    for key, what, frame0, cells, seed in legs:
        try:
            tr = H.device_walk_cell(device, frame0, cells, frames, 0.05, seed)
            torch.cuda.synchronize()
            # one COLD call (first use of this cell's kernel variant in the process: code-object load, scratch growth), then
            # three warm ones; kernel and wall time are the minima over the WARM calls, each taken on its own
            reps = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = Rdf.from_trajectory(tr, device=local_rank, distributed=False).result()
                reps.append((time.perf_counter() - t0, r._stats["kernel_s_dominant"], r._stats["kernel_s_all"], r._stats["path"]))
            first_call = reps[0][0]
            warm = reps[1:]
            wall_best = min(w for w, _, _, _ in warm)
            best = min(k for _, k, _, _ in warm)
            span = min(a for _, _, a, _ in warm)
            path = warm[-1][3]
            N = tr.n_atoms
            lengths = tr.cell_lengths()
            rmax = r.rmax
            lz = float(np.max(lengths))
            visited = min(1.0, 2.0 * rmax / lz + 3.0 / 256.0) if 2.0 * rmax * 1.05 < lz else 1.0
            pairs = frames * N * (N - 1) / 2.0
            leg = {"workload": "%s; %d atoms x %d frames, Rdf(dr=0.01, rmax=half shortest length = %.4f A -> %d bins)"
                               % (what, N, frames, rmax, len(r.data)),
                   "path": path, "kernel_ms_per_frame": 1e3 * best / frames, "wall_ms_per_frame": 1e3 * wall_best / frames,
                   "frames_per_s": frames / wall_best, "first_call_s": first_call, "warm_call_s": wall_best,
                   "stream_span_s": span,        # first to last kernel of the call on its stream (quantize + tile kernel + gaps)
                   "pair_evals_per_s": pairs / best, "visited_fraction_geometric": visited,
                   "vs_headline_per_pair": (best / pairs) / (head_cost * vis_head),
                   "vs_headline_per_visited_pair": (best / (pairs * visited)) / head_cost,
                   "vs_diagonal_same_frames_per_visited_pair": ((best / (pairs * visited)) / same_cost) if same_cost else None}
            if do_verify:
                fr = [0, frames - 1]
                leg["verified"] = bool(verify_rdf_timed(ctx, tr, rmax, len(r.data), r.hist, fr))
                leg["verification"] = {"rdf_frames": fr}
            out[key] = leg
            del tr, r
        except Exception as exc:
            out[key] = {"error": repr(exc)}
        torch.cuda.empty_cache()
    return out


def host_legs(packed, rdf, msd, local_rank, F, N):
    """Supplementary (never `value`): the same analyses from HOST input, i.e. what a caller of the reference's API pays
    before the kernels see a byte (amof/trajectory.py:27-35: a trajectory IS a list of Atoms).

    * host_resident: Rdf + WindowMsd on a packed host trajectory (numpy).  Round 5: ONE upload, started by the first analysis,
      which walks the frames that have arrived (amof_amd.frames.ResidentCopy); round 4 staged 1.2 GB per class.
    * pack_atoms_list: a list of 5000 frames -> packed array (native copy on all cores, amof_pack_frames), no GPU involved.
    * dropin_from_atoms_list: the list of 5000 frames handed to Rdf, WindowMsd, Bad and cn.CoordinationNumber one after the
      other, as in the reference's example script (examples/Compute structural properties.py:58-118): packed once into
      page-locked memory, uploaded once while it is packed, remembered for the three later constructors (a checksum of every
      frame says the list is unchanged); every result compared with the device-resident step's."""
    import torch
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd
    from amof_amd.bad import Bad
    from amof_amd.cn import CoordinationNumber
    from amof_amd import frames as fr
    out = {}
    pos_host = packed.pos.cpu().numpy()
    best = None
    for rep in range(3):
        host = fr.PackedTrajectory(pos_host, packed.cell, packed.numbers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a_ = Rdf.from_trajectory(host, device=local_rank, distributed=False)
        b_ = WindowMsd.from_trajectory(host, delta_time=100, timestep=1, device=local_rank, distributed=False)
        a_.result(), b_.result()
        w = time.perf_counter() - t0
        best = w if best is None or w < best else best
        same = bool(np.array_equal(np.asarray(a_.hist), np.asarray(rdf.hist)))
        host.release_device()
        del host, a_, b_
    out["host_resident_frames_per_s"] = F / best
    out["host_resident"] = {"wall_s": best, "verified": same,
                            "what": "Rdf + WindowMsd on a host PackedTrajectory (pageable numpy): one upload through page-locked "
                                    "staging, the RDF walks the frames as they arrive"}
    frames = [fr.Frame(packed.numbers, pos_host[k], packed.cell_of(k)) for k in range(F)]
    del pos_host
    best = None
    for rep in range(3):
        fr.forget_packed_lists()
        t0 = time.perf_counter()
        fr.pack_trajectory(frames)
        w = time.perf_counter() - t0
        best = w if best is None or w < best else best
    out["pack_atoms_list_frames_per_s"] = F / best
    out["pack_atoms_list"] = {"wall_s": best, "frames": F, "threads": fr._usable_cpus(),
                              "what": "list of %d frames -> packed host array (amof_pack_frames), no GPU" % F}
    best, first = None, None
    for rep in range(3):
        fr.forget_packed_lists()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r_ = Rdf.from_trajectory(frames, device=local_rank, distributed=False)
        ta = time.perf_counter()
        m_ = WindowMsd.from_trajectory(frames, delta_time=100, timestep=1, device=local_rank, distributed=False)
        tb = time.perf_counter()
        b_ = Bad.from_trajectory(frames, {'Zn-N': 2.5}, dtheta=0.05, device=local_rank, distributed=False)
        tc = time.perf_counter()
        c_ = CoordinationNumber.from_trajectory(frames, {'Zn-N': 2.5}, device=local_rank, distributed=False)
        t1 = time.perf_counter()
        n_ = len(r_.data) + len(m_.data) + len(b_.data) + len(c_.data)
        w = time.perf_counter() - t0
        first = w if first is None else first
        if best is None or w < best:
            best, t_ctor, t_each = w, t1 - t0, [ta - t0, tb - ta, tc - tb, t1 - tc]
    ok_r = bool(np.array_equal(np.asarray(r_.hist), np.asarray(rdf.hist)) and r_.data.equals(rdf.data))
    ok_m = bool(np.allclose(m_.sumsq, msd.sumsq, rtol=1e-12, atol=0.0))
    b_ref = Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05, device=local_rank, distributed=False)
    c_ref = CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5}, device=local_rank, distributed=False)
    ok_b = bool(np.array_equal(np.asarray(b_.hist), np.asarray(b_ref.hist)) and b_.data.equals(b_ref.data))
    ok_c = bool(c_.data.equals(c_ref.data))
    out["dropin_from_atoms_list"] = {
        "workload": "list of %d Frame objects (%d atoms) -> Rdf + WindowMsd + Bad({'Zn-N': 2.5}) + CoordinationNumber({'Zn-N': 2.5}), "
                    "every .data read" % (F, N),
        "wall_s": best, "first_call_s": first, "constructors_returned_after_s": t_ctor,
        "constructor_s": {"Rdf (packs, starts the upload)": t_each[0], "WindowMsd (recognises the list)": t_each[1], "Bad": t_each[2],
                          "CoordinationNumber": t_each[3]}, "frames_per_s": F / best,
        "verified": bool(ok_r and ok_m and ok_b and ok_c),
        "verification": {"rdf_equals_device_resident": ok_r, "msd": ok_m, "bad": ok_b, "cn": ok_c, "rows": n_}}
    fr.forget_packed_lists()
    del frames
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=5000)
    ap.add_argument("--reps", type=str, default="3,3,4")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = one shared trajectory sharded over the ranks (default); weak = a block per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rdf-frames", type=int, default=24)
    ap.add_argument("--no-bad", action="store_true", help="skip the configs[3] leg (RDF+BAD+MSD steps after the timed ones)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the realistic-cell and the supplementary configs[1] / configs[4] legs (N = 1)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle spot checks of the timed results")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for rehearsals")
    ap.add_argument("--all-on-device", type=int, default=None, help="rehearsal: put every rank on this GPU")
    ap.add_argument("--dump-hist", type=str, default=None, help="rank 0 saves the merged RDF histogram (npy) here")
    ap.add_argument("--rank-probe", type=int, default=None, metavar="CODE",
                    help="launcher check (no GPU needed): every rank prints its RANK / WORLD_SIZE and exits with CODE")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`): this process becomes the launcher.  It has not touched the
        # GPU (no torch import yet) and starts the ranks as a CHILD process -- never exec -- relaying their output
        # (rank 0's JSON line) and the exit code.
        raise SystemExit(self_launch(args.gpus))
    if args.rank_probe is not None:
        # (one write per rank: the ranks share the launcher's pipe, and a line split over two writes can interleave)
        os.write(1, (json.dumps({"probe": True, "rank": int(os.environ.get("RANK", "0")),
                                 "world": int(os.environ.get("WORLD_SIZE", "1")), "gpus": args.gpus,
                                 "master_addr": os.environ.get("MASTER_ADDR")}) + "\n").encode())
        raise SystemExit(args.rank_probe)

    import torch
    import torch.distributed as dist
    from amof_amd import _hip
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd
    from amof_amd.bad import Bad

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it plainly, or with --nproc-per-node %d)"
                         % (args.gpus, world, args.gpus))
    if args.all_on_device is not None:
        local_rank = args.all_on_device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # AMOF_DIST_FORCE_MERGE=1 (rehearsal on one GPU): a one-rank group, every collective of the N > 1 path really runs
    forced = world == 1 and os.environ.get("AMOF_DIST_FORCE_MERGE") == "1"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or forced:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    reps = tuple(int(x) for x in args.reps.split(","))
    F = args.frames
    strong = world == 1 or args.scaling == "strong"
    # strong scaling: the SAME trajectory on every rank (same seed), sharded by the classes; weak: own block per rank
    packed = make_trajectory(device, reps, F, 0.05, 20261003 + (0 if strong else rank))
    torch.cuda.synchronize()                        # generation finished before anything is timed
    N = packed.n_atoms
    same_traj = None
    if strong and (world > 1 or forced):
        # strong scaling shards ONE trajectory: every rank must have generated the same bits (same seed, same
        # generator on identical GPUs) -- checked with a wrapping integer checksum, min == max over the ranks
        chk = packed.pos.view(torch.int64).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        if args.backend != "nccl":
            lo, hi = lo.cpu(), hi.cpu()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same_traj = bool(int(lo.item()) == int(hi.item()))
        if not same_traj:
            if args.backend != "nccl":
                raise SystemExit("bench.py: the ranks generated different trajectories; strong scaling would merge unrelated frames")
            dist.broadcast(packed.pos, src=0)       # rank 0's trajectory for everyone (1.2 GB over xGMI, untimed)
            torch.cuda.synchronize()
    ctx = _hip.get_context(local_rank)
    mode = False if (world == 1 and not forced) else (None if strong else 'local')

    def step(with_bad, rec):
        # the constructors enqueue (amof_amd/_lazy.py: RDF on the device's first lane, the memory-bound analyses on the
        # second, a high-priority stream) and return; looking at `.data` waits -- EVERY result is looked at inside the
        # timed region, so nothing is left running when the clock stops
        t0 = time.perf_counter()
        rdf = Rdf.from_trajectory(packed, device=local_rank, distributed=mode)
        msd = WindowMsd.from_trajectory(packed, delta_time=100, timestep=1, device=local_rank, distributed=mode)
        bad = Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05, device=local_rank, distributed=mode) if with_bad else None
        t1 = time.perf_counter()
        n_rdf = len(rdf.data)
        t2 = time.perf_counter()
        n_msd = len(msd.data)
        t3 = time.perf_counter()
        if with_bad:
            n_bad = len(bad.data)
            rec["bad_wall"].append(time.perf_counter() - t3)      # what BAD adds after RDF and MSD are there
            rec["bad_all"].append(bad._stats["kernel_s_all"])
        assert n_rdf > 0 and n_msd > 0
        rec["submit_wall"].append(t1 - t0)
        rec["rdf_wall"].append(t2 - t0)                            # constructor call -> RDF DataFrame in hand
        rec["msd_wall"].append(t3 - t2)                            # what MSD adds after that (0 when it ran beside the RDF launch)
        rec["rdf_dom"].append(rdf._stats["kernel_s_dominant"])
        rec["rdf_all"].append(rdf._stats["kernel_s_all"])
        rec["msd_dom"].append(msd._stats["kernel_s_dominant"])
        rec["msd_all"].append(msd._stats["kernel_s_all"])
        return rdf, msd, bad

    def fence():
        torch.cuda.synchronize()
        if world > 1 or forced:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(with_bad):
        keys = ("rdf_dom", "rdf_all", "msd_dom", "msd_all", "rdf_wall", "msd_wall", "bad_wall", "bad_all", "submit_wall")
        rec = {k: [] for k in keys}
        for _ in range(args.warmup):
            step(with_bad, rec)
        rec = {k: [] for k in keys}
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step(with_bad, rec)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1 or forced:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        mean = {k: (float(np.mean(v)) if v else None) for k, v in rec.items()}
        return res, elapsed, mean

    (rdf, msd, _), elapsed, mean = timed(False)
    cfg3 = None
    if not args.no_bad:
        (rdf3, msd3, bad3), elapsed3, mean3 = timed(True)
        cfg3 = (bad3, elapsed3, mean3)
        assert "N-Zn-N" in bad3.data.columns

    # the MSD pipeline ALONE (nothing else on the GPU): what roofline_msd prices.  Inside the steps above its kernels
    # share the chip with the RDF launch, and their in-step durations say how well the two overlap, not how fast they are.
    seq = {"all": [], "dom": [], "wall": []}
    fence()
    for rep in range(args.warmup + args.steps):
        t0 = time.perf_counter()
        m = WindowMsd.from_trajectory(packed, delta_time=100, timestep=1, device=local_rank, distributed=mode)
        m.result()
        if rep >= args.warmup:
            seq["wall"].append(time.perf_counter() - t0)
            seq["all"].append(m._stats["kernel_s_all"])
            seq["dom"].append(m._stats["kernel_s_dominant"])
    msd_path = m._stats["path"]
    seq = {k: float(np.mean(v)) for k, v in seq.items()}
    del m
    fence()

    per_rank = {"rank": rank, "device": local_rank, "rdf_frames": list(_rdf_range(F, rank, world, strong)),
                "kernel_s": {k: mean[k] for k in ("rdf_dom", "rdf_all", "msd_all")},
                "wall_s": {k: mean[k] for k in ("rdf_wall", "msd_wall")}}
    if cfg3:
        per_rank["kernel_s"]["bad_all"] = cfg3[2]["bad_all"]
        per_rank["wall_s"]["bad_wall"] = cfg3[2]["bad_wall"]
    ranks = [per_rank]
    if world > 1 or forced:
        ranks = [None] * world
        dist.all_gather_object(ranks, per_rank)

    if rank == 0:
        frames_per_step = F if strong else world * F
        fps = frames_per_step * args.steps / elapsed
        rmax, nbins = rdf.rmax, len(rdf.data)
        t_rdf = mean["rdf_dom"]
        f_lo, f_hi = _rdf_range(F, rank, world, strong)
        f_loc = f_hi - f_lo
        alg_bytes = f_loc * (24 * N + 72)                  # 24N+72 bytes per frame per pass (this rank's launch)
        pairs = f_loc * N * (N - 1) / 2.0                  # unordered pair evaluations per launch
        in_range = float(np.asarray(rdf.hist).sum()) / 2.0 * (f_loc / float(frames_per_step))
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, gfx950 corrections applied by
        # profiles/tools/pmc_to_json.py), recorded under profiles/ for this exact workload; null for any other size
        traffic = {}
        stamp = sources_stamp()
        counters_current = {"traffic": False, "valu_model": False, "sources_sha256": stamp}
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile) and (N, F, world) == (9792, 5000, 1):
            with open(tfile) as fh:
                tj = json.load(fh)
            counters_current["traffic"] = tj.get("_sources_sha256") == stamp
            if counters_current["traffic"]:       # (counters of another build of the kernels are not quoted)
                traffic = tj.get("cfg3", {})
        # The bound that applies to the all-pairs kernel is VALU issue.  roofline_valu prices the kernel's
        # EXECUTED vector instructions (SQ counters of this very launch shape, profiles/tools/collect_pmc.sh ->
        # pmc_to_json.py -> profiles/valu_model.json) with the issue cost of each class measured on the box in-kernel
        # (profiles/tools/ubench2.hip -> profiles/r03/ubench2_valu_issue.txt: 2.19 cycles per wave-instruction and SIMD
        # for add/mul/fma/int, 4.1 for conversions / fract / compares / min / shift-add, 8.1 for sqrt), against the SIMD
        # cycles of the LIVE launch (live kernel seconds x the effective clock of the counter run).  Instructions the
        # counters do not classify are priced all-full-rate (frac) and all-half-rate (frac_high).
        lz = float(np.max(packed.cell_lengths()))
        visited = min(1.0, (2.0 * rmax) / lz + 3.0 / 256.0) if 2.0 * rmax * 1.05 < lz else 1.0
        valu = {"simd_cycles_per_wave_pair_all_pairs": t_rdf * N_SIMD * CLOCK_HZ / (pairs / 64.0),
                "visited_fraction_geometric": visited,
                "simd_cycles_per_visited_wave_pair": t_rdf * N_SIMD * CLOCK_HZ / (pairs / 64.0) / visited}
        roofline_valu = None
        mfile = os.path.join(ROOT, "profiles", "valu_model.json")
        if os.path.exists(mfile) and (N, F, world) == (9792, 5000, 1):
            with open(mfile) as fh:
                mj = json.load(fh)
            model = mj.get("rdf_tile_kernel_fast")
            counters_current["valu_model"] = mj.get("_sources_sha256") == stamp
            if model and model.get("effective_clock_ghz") and counters_current["valu_model"]:
                lo, hi = model["valu_issue_cycles_per_simd"]
                clock = model["effective_clock_ghz"] * 1e9
                live_cycles = t_rdf * clock
                valu["simd_cycles_per_visited_wave_pair"] = live_cycles * N_SIMD / (pairs / 64.0) / visited
                valu["simd_cycles_per_wave_pair_all_pairs"] = live_cycles * N_SIMD / (pairs / 64.0)
                valu["pmc"] = {"valu_instructions_per_launch": model["valu_instructions_per_launch"],
                               "valu_instructions_per_visited_wave_pair":
                                   model["valu_instructions_per_launch"] / (pairs / 64.0 * visited),
                               "by_class": model["by_class"], "other_valu": model["other_valu"],
                               "lane_utilisation": model["lane_utilisation"],
                               "cycles_per_instruction_used": model["cycles_per_instruction_used"],
                               "issue_slot_utilisation_in_counter_run": model["issue_slot_utilisation"]}
                roofline_valu = {
                    "kernel": model["kernel"], "bound": "valu_issue",
                    "achieved": lo / t_rdf / 1e9, "peak": clock / 1e9, "unit": "G issue-cycles/s per SIMD",
                    "frac": lo / live_cycles, "frac_high": min(1.0, hi / live_cycles),
                    "peak_fp32_vector_tflops": 157.3, "fp32_equivalent_tflops": 157.3 * lo / live_cycles,
                    "effective_clock_ghz": clock / 1e9, "cost_table": model.get("cost_table"),
                    "note": "share of the SIMDs' issue cycles the kernel's executed VALU instructions need at their "
                            "measured per-class issue costs; the rest is LDS / scalar / wait time"}
        # (atom-sharded: a rank reads ITS atoms, twice; the algorithmic bytes of its share)
        msd_bytes = alg_bytes if world == 1 else F * (24 * ((N + world - 1) // world) + 72)
        out = {
            "metric": "frames/s (RDF+MSD, 10k-atom ZIF-4)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2] headline: %d-atom ZIF-4 %dx%dx%d supercell, %d frames%s, "
                                   "Rdf(dr=0.01, rmax=half_cell -> %.4f A, %d bins) + WindowMsd(delta_time=100, W=%d)"
                                   % (N, reps[0], reps[1], reps[2], F, "" if strong else " per GPU", rmax, nbins,
                                      len(msd.data)),
                       "n_atoms": N, "frames_total": frames_per_step, "trajectory_identical_on_all_ranks": same_traj, "rdf_bins": nbins, "msd_windows": len(msd.data),
                       "parallelism": ("RDF frames sharded x%d + RCCL all-reduce of the u64 histograms in HBM; MSD atoms "
                                       "sharded x%d + all-reduce of the f64 sums" % (world, world)) if strong else
                                      ("own %d-frame block per rank x%d, RCCL all-reduce of the u64 histograms" % (F, world))},
            # ("bound" names the roof this block prices against -- the contract's "hbm" | "mfma"; the resource that
            #  actually limits this kernel is VALU issue: "limited_by", priced in roofline_valu)
            "roofline": {"kernel": "rdf_tile_kernel_fast", "bound": "hbm", "limited_by": "valu_issue (roofline_valu)",
                         "achieved": alg_bytes / t_rdf / 1e9,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_bytes / t_rdf / 1e9 / HBM_PEAK_GBPS,
                         "traffic": traffic.get("rdf_tile_kernel_fast"), "launch_seconds": t_rdf,
                         "algorithmic_bytes": alg_bytes, "frames_in_launch": f_loc,
                         "note": "all-pairs RDF at half-cell rmax does N(N-1)/2 = 4.8e7 pair evaluations per 235 kB "
                                 "frame: VALU-issue bound, not HBM bound (SURVEY 8d, DESIGN 4.1); the honest HBM "
                                 "fraction is tiny by construction and north_star's >= 50 % HBM target does not apply "
                                 "to this kernel -- see valu_issue and pair_evals_per_s"},
            "pair_evals_per_s": pairs / t_rdf, "pairs_in_range_per_s": in_range / t_rdf,
            "roofline_valu": roofline_valu, "valu_issue": valu, "counter_files_match_built_sources": counters_current,
            "roofline_msd": {"kernel": "msd pipeline, fused form (msd_seg: segment sums + tile sums of m p | com_tiles / com_steps / seg_scan | "
                                       "msd_fused: second pass over pos, products from registers and LDS | colreduce)"
                                       if world == 1 else "msd pipeline, atom-sharded fused form (shard_begin | all-reduce of the [F][3] sums | shard_finish)",
                             "bound": "hbm",
                             "achieved": msd_bytes / seq["all"] / 1e9, "peak": HBM_PEAK_GBPS,
                             "unit": "GB/s", "frac": msd_bytes / seq["all"] / 1e9 / HBM_PEAK_GBPS,
                             "traffic": traffic.get("msd_pipeline"), "pipeline_seconds": seq["all"],
                             "msd_window_kernel_seconds": seq["dom"], "call_wall_seconds": seq["wall"], "path": msd_path,
                             "algorithmic_bytes": msd_bytes,
                             "measured": "the MSD class alone, %d calls back to back after the timed steps (in the steps its "
                                         "kernels run BESIDE the RDF launch on the second lane: in_step_seconds)" % args.steps,
                             "in_step_seconds": mean["msd_all"]},
            "kernel_seconds_per_step": {"rdf_tile": t_rdf, "rdf_all_incl_quantize": mean["rdf_all"],
                                        "msd_all": mean["msd_all"]},
            "wall_seconds_per_step": {"constructors_return_after": mean["submit_wall"], "rdf_data_after": mean["rdf_wall"],
                                      "msd_data_adds": mean["msd_wall"],
                                      "note": "asynchronous constructors (amof_amd/_lazy.py): every .data is read inside the "
                                              "timed region; msd_data_adds = wait for the MSD result once the RDF result is there"},
            "host_seconds_per_step": {"rdf_wall_minus_kernels": mean["rdf_wall"] - mean["rdf_all"]},
            "async": os.environ.get("AMOF_ASYNC", "1") != "0",
            "per_rank": ranks,
        }
        if cfg3:
            bad3, elapsed3, mean3 = cfg3
            out["configs3"] = {
                "workload": "configs[3]: the same trajectory, Rdf + WindowMsd + Bad({'Zn-N': 2.5}, dtheta=0.05) per step",
                "value": frames_per_step * args.steps / elapsed3, "unit": "frames/s",
                "ms_per_step": 1e3 * elapsed3 / args.steps,
                "kernel_seconds_per_step": {"rdf_all": mean3["rdf_all"], "msd_all": mean3["msd_all"],
                                            "bad_all": mean3["bad_all"]},
                "bad_wall_s": mean3["bad_wall"],
                "roofline_bad": {"kernel": "bad pipeline (lists_frame_kernel + bad_rows_kernel: whole frame in LDS)", "bound": "hbm",
                                 "achieved": alg_bytes / mean3["bad_all"] / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": alg_bytes / mean3["bad_all"] / 1e9 / HBM_PEAK_GBPS,
                                 "algorithmic_bytes": alg_bytes},
                "n_angles": [int(x) for x in np.asarray(bad3.n_angles)],
            }
        if args.dump_hist:
            np.save(args.dump_hist, np.asarray(rdf.hist).view(np.uint64))
        if not args.no_verify and strong:
            try:
                out["verification"] = verify(packed, ctx, rdf, msd, [0, F // 3 + 1, F - 1][:3 if F > 2 else 1])
                out["verified"] = out["verification"]["ok"]
            except Exception as exc:        # the checker failing to run is not a pass
                out["verification"] = {"error": repr(exc)}
                out["verified"] = False
        else:
            out["verified"] = None
        if world == 1 and not args.no_cpu_baseline:
            window = np.arange(0, (F // 2), 100)
            out["cpu_baseline"] = cpu_baseline(packed, rmax, nbins, window, args.cpu_rdf_frames)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
            out["speedup_vs_cpu_1core"] = fps / out["cpu_baseline"]["value"]
            try:
                out.update(host_legs(packed, rdf, msd, local_rank, F, N))
            except Exception as exc:
                out["host_legs_error"] = repr(exc)
        if world == 1 and not args.no_extra:
            # coordination numbers on the headline trajectory (never `value`): the one kernel of the path that is HBM bound as
            # north_star pictures it -- the Zn / N rows of every frame are fetched once and searched inside LDS
            try:
                from amof_amd.cn import CoordinationNumber
                best_k, best_w = None, None
                for rep in range(args.warmup + 3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    cnh = CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5}, device=local_rank, distributed=False).result()
                    w = time.perf_counter() - t0
                    k = cnh._stats["kernel_s_all"]
                    if best_k is None or k < best_k:
                        best_k, best_w = k, w
                out["cn_headline"] = {
                    "workload": "CoordinationNumber({'Zn-N': 2.5}) on the headline trajectory (%d atoms x %d frames)" % (N, F),
                    "path": cnh._stats["path"], "frames_per_s": F / best_w, "wall_s": best_w, "kernel_s": best_k,
                    "roofline_cn": {"kernel": "cn_frame_kernel (whole frame of the species pair in LDS)", "bound": "hbm",
                                    "achieved": F * (24 * N + 72) / best_k / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": F * (24 * N + 72) / best_k / 1e9 / HBM_PEAK_GBPS,
                                    "algorithmic_bytes": F * (24 * N + 72), "traffic": traffic.get("cn_pipeline")},
                    "mean_cn": float(cnh.data['Zn-N'].mean())}
                if not args.no_verify:
                    from oracle import clib
                    from amof_amd import atom as amatom
                    from tests import helpers as H
                    kinds_h, sp_h = H.species_of(packed.numbers)
                    rcm = amatom.cutoff_matrix(amatom.format_cutoff({'Zn-N': 2.5}), kinds_h)
                    pick = [0, F // 2, F - 1]
                    pos_c = packed.pos[torch.as_tensor(pick, device=packed.pos.device)].cpu().numpy()
                    sums = clib.cn_counts(pos_c, packed.cell, sp_h, len(kinds_h), rcm, [(kinds_h.index(30), kinds_h.index(7))])
                    n_zn = int((packed.numbers == 30).sum())
                    out["cn_headline"]["verified"] = bool(np.array_equal(sums[:, 0] / n_zn, cnh.data['Zn-N'].values[pick]))
                del cnh
            except Exception as exc:
                out["cn_headline"] = {"error": repr(exc)}
            del packed
            torch.cuda.empty_cache()
            try:
                out["realistic_cells"] = realistic_cells(device, local_rank, ctx, (t_rdf, f_loc, N, visited),
                                                         do_verify=not args.no_verify)
            except Exception as exc:
                out["realistic_cells"] = {"error": repr(exc)}
            try:
                out["supplementary"] = supplementary(device, local_rank, ctx, do_verify=not args.no_verify)
            except Exception as exc:
                out["supplementary"] = {"error": repr(exc)}
        print(json.dumps(out))
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a child process on 127.0.0.1 and a free port; returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def _rdf_range(F, rank, world, strong):
    from amof_amd import dist as adist
    return adist.shard_range(F, rank, world) if (strong and world > 1) else (0, F)


if __name__ == "__main__":
    main()
