/*
 * amof_hip.h -- C ABI of libamofhip.so: MI355X (gfx950) kernels for aMOF's
 * per-frame pair-distance hot path (RDF / CN / BAD / window MSD).
 *
 * The reference (coudertlab/amof v1.1.0) is pure Python and has no FFI of its
 * own; the seam this library fills is the set of third-party / numpy call
 * sites underneath its analysis classes.  Each entry point names the reference
 * interface it replaces (paths relative to the reference root).  Host Python
 * (the amof_amd Python package) keeps everything that is O(bins): bin-count arithmetic,
 * normalisation, column naming, DataFrames.
 *
 * Conventions
 *   - plain C types only; no C++ exceptions cross the boundary; never aborts.
 *   - return value: 0 = AMOF_OK, negative = error; amof_last_error(ctx) gives
 *     a human-readable message for the last failing call on that context.
 *   - buffers are caller-owned; the library keeps no caller pointer after a call
 *     returns.  Every entry point is synchronous: it returns after its work on the
 *     context's stream has completed.  "_dev" entry points take caller-owned DEVICE
 *     output buffers (e.g. a torch tensor that an RCCL all-reduce consumes next) and
 *     run on the stream set with amof_ctx_set_stream.
 *   - a context is bound to one device and is not thread-safe; distinct
 *     contexts may be used concurrently.  ctypes releases the GIL during calls.
 *   - all integer results are exact and independent of launch geometry.
 */
#ifndef AMOF_HIP_H
#define AMOF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMOF_OK 0
#define AMOF_EINVAL (-1)     /* bad argument */
#define AMOF_ESINGULAR (-2)  /* singular cell */
#define AMOF_EANGLE (-3)     /* undefined angle (zero-length bond vector): ASE raises ZeroDivisionError */
#define AMOF_ENOMEM (-4)     /* host or device allocation failed */
#define AMOF_EHIP (-5)       /* HIP runtime error (message in amof_last_error) */
#define AMOF_ECAPACITY (-6)  /* a documented kernel capacity was exceeded */
#define AMOF_ENODEVICE (-7)  /* no usable GPU */
#define AMOF_EUNSUPPORTED (-8) /* a specialised entry point does not take these arguments: use the general one */

#define AMOF_ABI_VERSION 4

/* capacities */
#define AMOF_MAX_LDS_BINS 36864     /* histogram bins held in LDS per workgroup (u32); more bins: global-memory kernels */
#define AMOF_MAX_NEIGHBOURS 32      /* neighbours per centre atom the exact BAD kernel keeps in LDS (the fast kernels keep
                                       16 and hand a call with a fuller centre to the exact kernel); a centre with more
                                       sends the call through a further pass with lists in global memory (no error) */
#define AMOF_MAX_IMAGES 4096        /* extra periodic images per frame (AMOF_ECAPACITY when exceeded) */

typedef struct amof_ctx amof_ctx;

/*
 * A trajectory as the reference sees it: a list of F frames of the same N atoms
 * (amof/trajectory.py:27-35; species read from frame 0 only, amof/rdf.py:71).
 */
typedef struct amof_traj {
    const double *pos;      /* [F][N][3] xyz, float64, C-contiguous                  */
    int32_t pos_on_device;  /* 0: pos is a host pointer; 1: device pointer (resident) */
    int32_t n_species;      /* S                                                     */
    const double *cell;     /* HOST [n_cells][3][3], rows = cell vectors (ASE)       */
    int64_t n_cells;        /* 1 (constant cell) or F                                */
    int64_t n_frames;       /* F                                                     */
    int64_t n_atoms;        /* N                                                     */
    const int32_t *species; /* HOST [N], species index 0..S-1                        */
    const double *masses;   /* HOST [N]; MSD only (centre of mass), else may be NULL */
    uint8_t pbc[3];         /* periodic flags per cell vector (ase.Atoms.pbc)        */
    uint8_t _pad[5];
} amof_traj;

int amof_abi_version(void);

/* number of visible GPUs (hipGetDeviceCount); 0 when none */
int amof_device_count(void);

int amof_ctx_create(int device, amof_ctx **out);
/* The same with flags.  AMOF_CTX_HIGH_PRIORITY: the context's stream is created at the device's highest stream
 * priority -- the second context of a device, on which the memory-bound analyses (MSD, BAD, CN) run beside the
 * pair-evaluation-bound RDF launch of the first (the reference runs its analyses as independent calls,
 * examples/Compute structural properties.py:58-118, and parallelises inside them with joblib, amof/msd.py:252-256);
 * their workgroups are dispatched ahead of the RDF kernel's whenever a CU has room. */
#define AMOF_CTX_HIGH_PRIORITY 1
/* AMOF_CTX_LOW_PRIORITY: the lowest stream priority -- its workgroups are dispatched where the other context's kernel
 * leaves compute units free: before that kernel starts and in its tail, instead of delaying it. */
#define AMOF_CTX_LOW_PRIORITY 2
int amof_ctx_create2(int device, int flags, amof_ctx **out);
void amof_ctx_destroy(amof_ctx *ctx);
const char *amof_last_error(const amof_ctx *ctx);

/* Run subsequent work on the caller's hipStream_t (e.g. torch's current
 * stream).  NULL restores the context's own stream. */
int amof_ctx_set_stream(amof_ctx *ctx, void *hip_stream);
int amof_ctx_synchronize(amof_ctx *ctx);
/* Order the context's stream after everything queued so far on another stream of the same device
 * (hipEventRecord + hipStreamWaitEvent; NULL = the legacy default stream).  A caller that hands over
 * a device-resident `pos` produced by work still pending on its own stream (a torch tensor, say)
 * calls this first: the library's kernels otherwise run on the context's non-blocking stream with
 * no ordering against the producer. */
int amof_ctx_wait_stream(amof_ctx *ctx, void *hip_stream);
/* Two contexts of one device, driven by two host threads (amof_amd/_lazy.py: the RDF on one, the memory-bound analyses on
 * the other).  Their kernels do not run well side by side -- the RDF tile kernel fills the LDS and the register file of every
 * CU, a second stream's workgroups trickle in between and both lose (profiles/r05/stops.txt A) -- so the second context
 * FOLLOWS the first: amof_ctx_follow waits on the HOST (at most timeout_s) until `other` has begun its call number
 * min_calls (amof_ctx_calls counts them; 0: no host wait) and queued that call's dominant kernel, then orders ctx's stream
 * after everything `other` has queued so far (amof_ctx_wait_stream).  The follower's host work (tables, uploads) then runs
 * beside the leader's kernel, its kernels right behind it, beside the leader's read-back and result assembly.
 * Returns 1 when ordered, 0 on timeout (nothing ordered: the caller decides), < 0 on error.  Safe to call while `other` is
 * inside a call on another thread.  The classes use it only on request (AMOF_LANE_ORDER=1): on the headline workload the
 * ordered lanes measure the same as the unordered ones at N = 1 and 0.1 - 0.2 ms slower for one rank of eight
 * (profiles/r05/lane_order.txt).  Replaces nothing in the reference (its analyses run one after the other:
 * examples/Compute structural properties.py:58-118). */
int amof_ctx_follow(amof_ctx *ctx, amof_ctx *other, int64_t min_calls, double timeout_s);
/* calls of this context that have started device work so far (any thread may ask) */
int64_t amof_ctx_calls(const amof_ctx *ctx);
/* Diagnostics for the test-suite: fill every scratch buffer the context currently owns with `byte`
 * (after a synchronisation).  No call may depend on what a previous call left in scratch; a test
 * that poisons between calls turns such a dependence into a wrong answer instead of a lucky one. */
int amof_ctx_debug_poison(amof_ctx *ctx, int byte);

/* Seconds spent inside kernels of the last call, measured with HIP events on
 * the context's stream around the dominant kernel's launches:
 * which = 0 total, 1 dominant kernel only. Returns < 0 if unavailable. */
double amof_last_kernel_seconds(const amof_ctx *ctx, int which);
/* number of launches of the dominant kernel in the last call */
int64_t amof_last_kernel_launches(const amof_ctx *ctx);
/* kernel family that produced the result of the last call (diagnostics and tests; "" before the first call):
 *   RDF  "rdf_tile_zf" (diagonal cells, f32 slab coordinates), "rdf_tile_tri" (general cells in the orthogonalised
 *        lattice frame, f32 slab coordinates), "rdf_tile" (plain general tile kernel), "rdf_tile_img" (cutoffs beyond
 *        half a cell height where "rdf_tile_tri" does not apply), "rdf_cell" (3-D cell list), "rdf_range" (2-level list),
 *        "rdf_exact" (canonical float64 arithmetic per pair)
 *   CN   "cn_frame" (whole frame in LDS), "cn_frame_slabs" (z-slabs of a frame in LDS: pairs of more than 8192 atoms),
 *        "cn_cell", "cn_fast", "cn_exact"
 *   BAD  "bad_frame" (whole frame in LDS), "bad_frame_slabs" (z-slabs of a frame in LDS), "bad_cell", "bad_fast",
 *        "bad_exact", "bad_exact_biglist"
 *   MSD  "msd_fused" (no transposed copy: diagonal cells, evenly spaced windows), "msd_stream" (register-ring comb, window
 *        spacing 64..256), "msd_comb" (block comb kernels incl. the
 *        double-buffered and > 32-window passes), "msd_group" (arbitrary window lists), "msd_comb_global" /
 *        "msd_global" (series too long for LDS), "msd_direct", "msd_com" (amof_msd_com_dev alone) */
const char *amof_last_path(const amof_ctx *ctx);

/*
 * RDF histogram accumulation.
 * Replaces asap3.analysis.rdf.RadialDistributionFunction(atoms, rMax, nBins)
 * + .update() per frame, as driven by amof/rdf.py:88-93 (and :181-185).
 *   hist[(a*S + b)*nbins + k] += number of ORDERED pairs (i of species a,
 *       j of species b, any periodic image, zero-shift self pair excluded)
 *       with bin k = (int)(r / (rmax/nbins)), r < rmax
 *   *volume_sum += sum over frames of the cell volume (asap3 keeps it for
 *       the normalisation done by get_rdf, amof/rdf.py:96,109)
 * The total histogram is the sum over (a,b).  hist is accumulated into
 * (+=), so the caller zeroes it first.
 */
int amof_rdf_accumulate(amof_ctx *ctx, const amof_traj *traj, double rmax, int32_t nbins,
                        uint64_t *hist /* host [S*S][nbins] */, double *volume_sum);
int amof_rdf_accumulate_dev(amof_ctx *ctx, const amof_traj *traj, double rmax, int32_t nbins,
                            uint64_t *hist_dev /* device [S*S][nbins] */, double *volume_sum);

/*
 * Coordination-number counts.
 * Replaces amof.atom.get_neighborlist (amof/atom.py:72-87, i.e.
 * ase.neighborlist.neighbor_list('ij', atoms, {(Z1,Z2): rc})) + the counting
 * loop of amof/cn.py:67-73.
 *   cutoff[a*S + b]: per-species-pair cutoff (symmetric); 0 = never neighbours.
 *       Neighbour test is strict: sqrt(d2) < rc, every periodic image counts.
 *   sets[s] = (A, B): sums[f*n_sets + s] = sum over atoms i of species A of
 *       the number of neighbours of species B (the mean is sums / N_A).
 *   per_atom (optional, may be NULL) [F][n_sets][N]: that count per atom,
 *       -1 where the atom is not of species A.
 */
int amof_cn_count(amof_ctx *ctx, const amof_traj *traj, const double *cutoff /* [S][S] */,
                  const int32_t *sets /* [n_sets][2] */, int32_t n_sets,
                  int64_t *sums /* host [F][n_sets] */, int32_t *per_atom /* host or NULL */);

/*
 * Bond-angle histograms.
 * Replaces amof.atom.get_neighborlist + ase.Atoms.get_angles(idx, mic=True) +
 * numpy.histogram(bins=edges) as driven by amof/bad.py:70-114,154-160.
 *   triples[t] = (A, B): centre species A, neighbour species B; -1 = "X" (any).
 *   edges[nb+1]: histogram edges in degrees (host builds arange(bins+2)*dtheta,
 *       amof/bad.py:143); bin k holds edges[k] <= x < edges[k+1], last bin
 *       right-closed, values outside are dropped (numpy.histogram).
 *   hist[t*nb + k] += counts;  n_angles[t] += number of angles found
 * Returns AMOF_EANGLE for a zero-length bond vector (ASE: ZeroDivisionError).
 */
int amof_bad_hist(amof_ctx *ctx, const amof_traj *traj, const double *cutoff /* [S][S] */,
                  const int32_t *triples /* [T][2] */, int32_t n_triples,
                  const double *edges /* [nb+1] */, int32_t nb,
                  uint64_t *hist /* host [T][nb] */, uint64_t *n_angles /* host [T] */);
int amof_bad_hist_dev(amof_ctx *ctx, const amof_traj *traj, const double *cutoff,
                      const int32_t *triples, int32_t n_triples, const double *edges, int32_t nb,
                      uint64_t *hist_dev /* device [T][nb] */, uint64_t *n_angles_dev /* device [T] */);

/*
 * Bond-angle histograms split by the centre atom's number of B-neighbours.
 * Replaces BadByCn.bad_BAB (amof/bad.py:190-224): slot c (0..cn_max) of triple t holds the
 * angles of centres with exactly c B-neighbours (c = cn_max also collects any larger count;
 * 1 <= cn_max <= 65535).  hist[(t*(cn_max+1) + c)*nb + k], n_angles[t*(cn_max+1) + c].
 */
int amof_bad_hist_by_cn(amof_ctx *ctx, const amof_traj *traj, const double *cutoff,
                        const int32_t *triples, int32_t n_triples, const double *edges, int32_t nb,
                        int32_t cn_max, uint64_t *hist, uint64_t *n_angles);

/*
 * Window-averaged MSD partial sums.
 * Replaces amof.trajectory.get_delta_pos (amof/trajectory.py:285-303, i.e.
 * ase.geometry.wrap_positions(d, cell[k], center=0)) + the per-window loop
 * WindowMsd.compute_msd_of_m (amof/msd.py:185-205) + the centre-of-mass
 * removal and optional unwrap of amof/msd.py:222-237.
 *   windows[w] = m (frames); 0 <= m < F
 *   sumsq[s*W + w] = sum over atoms i of species s in [atom_begin, atom_end)
 *                    of sum_{k=1}^{F-m-1} |u_i(k+m) - u_i(k)|^2
 *       where u_i is the running sum of wrapped frame-to-frame displacements.
 *       (The reference never evaluates time origin k=0 and divides by F-m:
 *        MSD_s(m) = sumsq / N_s / (F-m); the host applies it.)
 *   unwrap != 0: rebuild unwrapped positions first (amof/msd.py:222-230).
 *   remove_com != 0: subtract the mass-weighted centre of mass of ALL atoms
 *       per frame (amof/msd.py:235-237; always on in the reference).
 * [atom_begin, atom_end) lets several devices split the atoms; partial sums
 * add up exactly like the full call up to float64 summation order.
 */
int amof_msd_window(amof_ctx *ctx, const amof_traj *traj, const int32_t *windows, int32_t n_windows,
                    int32_t unwrap, int32_t remove_com, int64_t atom_begin, int64_t atom_end,
                    double *sumsq /* host [S][W] */);
/*
 * The same with the sums ADDED into a device buffer (it stays in HBM for the ranks' all-reduce) and, optionally, the
 * per-frame centre of mass handed in (com_dev: device [F][3], NULL = computed here from all atoms; not with unwrap).
 * amof_msd_com_dev writes rows [frame_begin, frame_end) of that table -- masses @ positions / masses.sum() per frame
 * (ase get_center_of_mass, amof/msd.py:235-237) -- and leaves the others alone: the ranks of an atom-sharded run
 * each compute the centre of mass of their FRAME share into a zeroed table and sum the tables (x + 0 = x: exact),
 * instead of every rank reading every frame.  The element-parallel split this stands in for: amof/msd.py:252-256.
 */
int amof_msd_window_dev(amof_ctx *ctx, const amof_traj *traj, const int32_t *windows, int32_t n_windows,
                        int32_t unwrap, int32_t remove_com, int64_t atom_begin, int64_t atom_end,
                        const double *com_dev /* device [F][3] or NULL */, double *sumsq_dev /* device [S][W], += */);
int amof_msd_com_dev(amof_ctx *ctx, const amof_traj *traj, int64_t frame_begin, int64_t frame_end,
                     double *com_dev /* device [F][3] */);

/*
 * Atom-sharded window MSD around ONE all-reduce (one process per GPU; the element-parallel split this stands in for:
 * amof/msd.py:252-256).  Every rank holds the whole device-resident trajectory and owns the atoms [atom_begin, atom_end):
 *   amof_msd_shard_begin  reads ITS atoms once: csum_dev[F][3] (device, overwritten) = sum over its atoms of m_a p_a(k),
 *                         its share of the centre of mass of every frame (amof/msd.py:235-237), and keeps the segment
 *                         sums of its atoms' wrapped displacements in the context's scratch;
 *   the caller sums csum_dev over the ranks (one all-reduce of 24 F bytes);
 *   amof_msd_shard_finish reads its atoms a second time and ADDS their sums of squared displacements (amof_msd_window's
 *                         definition) into sumsq_dev[S][W] (device), which the caller all-reduces next.
 * finish must be the next call on the context after its begin, with the same arguments.  begin returns AMOF_EUNSUPPORTED
 * (nothing done) where this form does not apply -- general (non-diagonal) cells, host positions, windows that are not
 * w * d (16 <= d, W <= 32) -- and the caller then uses amof_msd_com_dev + amof_msd_window_dev.
 */
int amof_msd_shard_begin(amof_ctx *ctx, const amof_traj *traj, const int32_t *windows, int32_t n_windows,
                         int64_t atom_begin, int64_t atom_end, double *csum_dev /* device [F][3] */);
int amof_msd_shard_finish(amof_ctx *ctx, const amof_traj *traj, const int32_t *windows, int32_t n_windows,
                          int64_t atom_begin, int64_t atom_end, const double *csum_dev /* device [F][3], summed over the ranks */,
                          double *sumsq_dev /* device [S][W], += */);

/*
 * Direct MSD with running unwrap, orthogonal cells only (deprecated in the reference).
 * Replaces DirectMsd.compute_species_msd (amof/msd.py:83-107) for every species at once:
 *   msd[t*(S+1) + 0]     = sum over all atoms  |r_i(t) - r_i(0)|^2 / N     (column 'X')
 *   msd[t*(S+1) + 1 + s] = the same over the atoms of species s / N_s
 * with r_i(t) = r_i(t-1) + fold(pos_i(t) - (r_i(t-1) % a_t)) per axis, a_t = cell_t[j][j].
 */
int amof_msd_direct(amof_ctx *ctx, const amof_traj *traj, double *msd /* host [F][S+1] */);

/*
 * Trajectory ingest (host only).
 * Replaces, for the packed path, ase.io.read(filename, index, format='xyz') as driven by
 * Trajectory.from_traj / read_lammps_traj / read_cp2k_traj (amof/trajectory.py:37-60,193-228)
 * and np.genfromtxt on the CP2K cell log (amof/trajectory.py:217).
 *   amof_xyz_scan: number of frames and atoms per frame (all frames must agree).
 *   amof_xyz_read: frames first, first+step, ... (count of them) into pos[count][N][3], N = n_atoms as
 *       amof_xyz_scan reported it (AMOF_EINVAL if the file no longer agrees: nothing is written then);
 *       symbols[N][4] (NUL padded) from the first frame read; lattice[count][9] (may be NULL)
 *       receives extended-XYZ Lattice="..." when every frame carries one (*has_lattice = 1).
 *       n_threads <= 0: all hardware threads.  Numbers are parsed correctly rounded.
 *   amof_cp2k_cell_read: columns [2:-1] (Ax..Cz) of every data row into cell[rows][9];
 *       cell == NULL only counts rows.
 *   amof_xyz_open / amof_xyz_read_frames / amof_xyz_close: the same reader on an OPEN file -- mapping and frame index
 *       are built once and serve any number of batch reads (streamed analyses: amof_amd/stream.py); a handle may be
 *       read from several threads at once.  pos == NULL with lattice != NULL reads the Lattice of the frames only.
 * Errors: negative code, message via amof_ingest_last_error() (thread local).
 */
typedef struct amof_xyz_file amof_xyz_file;
int amof_xyz_open(const char *path, amof_xyz_file **out, int64_t *n_frames, int64_t *n_atoms);
int amof_xyz_read_frames(amof_xyz_file *file, int64_t first, int64_t count, int64_t step, int64_t n_atoms, double *pos,
                         char *symbols, double *lattice, int32_t *has_lattice, int32_t n_threads);
void amof_xyz_close(amof_xyz_file *file);
int amof_xyz_scan(const char *path, int64_t *n_frames, int64_t *n_atoms);
int amof_xyz_read(const char *path, int64_t first, int64_t count, int64_t step, int64_t n_atoms, double *pos,
                  char *symbols, double *lattice, int32_t *has_lattice, int32_t n_threads);
int amof_cp2k_cell_read(const char *path, int64_t max_rows, double *cell, int64_t *n_rows);
const char *amof_ingest_last_error(void);

/*
 * Packing a list of frames (host only).
 * Replaces the per-frame Python walk over a list of ase.Atoms (amof/trajectory.py:27-35,56-59; amof/rdf.py:88-93,
 * amof/msd.py:218-242) for the packed path: frame_pos[k] points at frame k's positions ([N][3] float64, C-contiguous --
 * ase.Atoms.positions); amof_pack_frames copies them into dst[F][N][3] on n_threads threads (<= 0: all hardware threads)
 * and, when checksums != NULL, fingerprints each frame's bytes; amof_frames_checksum fingerprints without copying (is a
 * list that was packed before still the same?).  Equal bytes give equal checksums; the hash is not cryptographic.
 */
int amof_pack_frames(const double *const *frame_pos, int64_t n_frames, int64_t n_atoms, double *dst, uint64_t *checksums,
                     int32_t n_threads);
int amof_frames_checksum(const double *const *frame_pos, int64_t n_frames, int64_t n_atoms, uint64_t *checksums,
                         int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif /* AMOF_HIP_H */
