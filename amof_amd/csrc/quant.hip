// Fixed-point folding + slab counting sort shared by the RDF and neighbour fast paths.
#include <math.h>

#include <algorithm>

#include "amof_internal.h"

namespace amof {

// One workgroup per (species, frame): fold the atoms into the cell, quantise to 2^-32 of
// the cell vectors, and counting-sort the species segment into 256 slabs along cell axis
// `axis` (order inside a slab is arbitrary -- every result downstream is an integer count,
// independent of the order).  slab_start (optional) [nf][S][SLABS+1]: offset, relative to
// the species segment, of the first atom of every slab.
constexpr int QUANT_THREADS = 1024;

__global__ __launch_bounds__(QUANT_THREADS) void quantize_kernel(const double *__restrict__ pos,
                                                       const double *__restrict__ geom, int n_cells,
                                                       const int32_t *__restrict__ perm,
                                                       const int64_t *__restrict__ sp_first, int S, int64_t N,
                                                       int f0, int axis, QAtom *__restrict__ Q,
                                                       uint32_t *__restrict__ slab_start, int32_t *flag, int cache_cap,
                                                       int ax0, int ax1, const double *__restrict__ fold,
                                                       unsigned long long used_mask)
{
    if (blockIdx.x < 64 && !((used_mask >> blockIdx.x) & 1ull)) return;    // (a species nobody reads: neither sorted nor written)
    // species segments of up to cache_cap atoms keep their quantised records in LDS between the
    // counting pass and the placement pass, so the positions are read from HBM once
    extern __shared__ __align__(16) unsigned char qcache_raw[];
    QAtom *cache = reinterpret_cast<QAtom *>(qcache_raw);
    __shared__ unsigned cnt[QSLABS];
    __shared__ unsigned wsum[4];
    const int sp = blockIdx.x, fl = blockIdx.y, tid = threadIdx.x;
    const int f = f0 + fl;
    // stored order: (ax0, ax1, axis)
    const double *__restrict__ g = geom + (size_t)(n_cells == 1 ? 0 : f) * GEOM_STRIDE;
    const double *__restrict__ fo = fold ? fold + (size_t)(n_cells == 1 ? 0 : f) * 2 : nullptr;
    const int64_t k0 = sp_first[sp], k1 = sp_first[sp + 1];
    const bool cached = k1 - k0 <= (int64_t)cache_cap;
    if (tid < QSLABS) cnt[tid] = 0u;
    __syncthreads();
    for (int64_t k = k0 + tid; k < k1; k += QUANT_THREADS) {
        QAtom q = quantize_atom(pos, g, N, f, perm[k], ax0, ax1, axis, flag);
        if (fo) fold_atom(q, fo);
        if (cached) cache[k - k0] = q;
        atomicAdd(&cnt[q.uz >> 24], 1u);
    }
    __syncthreads();
    // exclusive scan of the 256 counters (one per thread of the first four waves)
    unsigned v = tid < QSLABS ? cnt[tid] : 0u, incl = v;
    const int lane = tid & 63, wv = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    if (lane == 63 && wv < 4) wsum[wv] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wv && w < 4; w++) base += wsum[w];
    __syncthreads();
    const unsigned excl = base + incl - v;
    if (tid < QSLABS) {
        cnt[tid] = excl;
        if (slab_start) {
            uint32_t *st = slab_start + ((size_t)fl * S + sp) * (QSLABS + 1);
            st[tid] = excl;
            if (tid == QSLABS - 1) st[QSLABS] = excl + v;
        }
    }
    __syncthreads();
    QAtom *__restrict__ Qf = Q + (size_t)fl * N + k0;
    for (int64_t k = k0 + tid; k < k1; k += QUANT_THREADS) {
        QAtom q;
        if (cached) {
            q = cache[k - k0];
        } else {
            q = quantize_atom(pos, g, N, f, perm[k], ax0, ax1, axis, flag);
            if (fo) fold_atom(q, fo);
        }
        const unsigned slot = atomicAdd(&cnt[q.uz >> 24], 1u);
        Qf[slot] = q;
    }
}

// One workgroup per FRAME, all species (round 5).  quantize_kernel above runs one workgroup per (species, frame) and
// gathers its species' atoms through the permutation: the species are interleaved in the frame (a ZIF-4 supercell repeats
// a 272-atom cell), so every 128-byte line of the frame is fetched by up to S workgroups -- 3.40 GB of traffic for 1.96 GB
// of positions read + records written (profiles/traffic.json, round 4).  Here the frame is read ONCE, in atom order
// (coalesced), a thread keeps its <= QF_APT atoms' fixed-point coordinates in registers between the counting and the
// placement pass, and the 256 slab counters exist once per species in LDS; the placement goes species by species through an
// LDS stage, so that the records leave as whole lines.  Same output: Q[fl][species segments][slab order],
// slab_start[fl][S][257]; the order inside a slab is arbitrary, as before.
constexpr int QF_APT = 12;          // atoms per thread: frames of up to 12 288 atoms
constexpr int QF_MAXS = 8;          // species with counters in LDS

// species of every atom as a byte, in atom order (once per launch_quantize: the per-frame workgroups used to rebuild it in
// LDS from the permutation -- ten dependent L2 round trips per thread at the head of a workgroup that lives ~35 us)
__global__ __launch_bounds__(256) void species_bytes_kernel(const int32_t *__restrict__ perm, const int64_t *__restrict__ sp_first,
                                                            int S, int64_t N, uint8_t *__restrict__ spec)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    int sp = 0;
    while (sp + 1 < S && k >= sp_first[sp + 1]) sp++;
    spec[perm[k]] = (uint8_t)sp;
}

// (86 registers: one workgroup per CU.  Held to 64 for two, the compiler spills 96 bytes per lane and the kernel is slower
//  than before, 0.83 against 0.57 ms for quantisation + finalisation of 5000 frames: profiles/r05/quantize_experiments.txt)
__global__ __launch_bounds__(QUANT_THREADS) void quantize_frame_kernel(
    const double *__restrict__ pos, const double *__restrict__ geom, int n_cells, const uint8_t *__restrict__ spec,
    const int64_t *__restrict__ sp_first, int S, int64_t N, int f0, int axis, QAtom *__restrict__ Q, uint32_t *__restrict__ slab_start,
    int32_t *flag, int ax0, int ax1, const double *__restrict__ fold, unsigned long long used_mask, int stage_cap)
{
    extern __shared__ __align__(16) unsigned char qf_raw[];
    unsigned *cnt = reinterpret_cast<unsigned *>(qf_raw);                                // [S][QSLABS], then stage[stage_cap]
    __shared__ unsigned wsum[QUANT_THREADS / 64];
    const int fl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = f0 + fl;
    const double *__restrict__ g = geom + (size_t)(n_cells == 1 ? 0 : f) * GEOM_STRIDE;
    const double *__restrict__ fo = fold ? fold + (size_t)(n_cells == 1 ? 0 : f) * 2 : nullptr;
    for (int i = tid; i < S * QSLABS; i += QUANT_THREADS) cnt[i] = 0u;
    // the species of this thread's atoms, four bits each (15: no atom, or a species nobody reads)
    unsigned long long spk = 0ull;
#pragma unroll
    for (int j = 0; j < QF_APT; j++) {
        const int64_t a = tid + (int64_t)j * QUANT_THREADS;
        unsigned sp = a < N ? (unsigned)spec[a] : 15u;
        if (sp < 15u && !((used_mask >> sp) & 1ull)) sp = 15u;
        spk |= (unsigned long long)sp << (4 * j);
    }
    __syncthreads();
    uint32_t ux[QF_APT], uy[QF_APT], uz[QF_APT];
#pragma unroll
    for (int j = 0; j < QF_APT; j++) {
        const int64_t a = tid + (int64_t)j * QUANT_THREADS;
        ux[j] = uy[j] = uz[j] = 0u;
        const unsigned sp = (unsigned)(spk >> (4 * j)) & 15u;
        if (sp < 15u) {
            QAtom q = quantize_atom(pos, g, N, f, a, ax0, ax1, axis, flag);
            if (fo) fold_atom(q, fo);
            ux[j] = q.ux; uy[j] = q.uy; uz[j] = q.uz;
            atomicAdd(&cnt[sp * QSLABS + (q.uz >> 24)], 1u);
        }
    }
    __syncthreads();
    // exclusive scans of the S x 256 counters: four species per round, one per group of four waves
    for (int s0 = 0; s0 < S; s0 += QUANT_THREADS / QSLABS) {
        const int grp = tid / QSLABS, sp = s0 + grp, t = tid % QSLABS;
        const unsigned v = sp < S ? cnt[sp * QSLABS + t] : 0u;
        unsigned incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned n = __shfl_up(incl, off, 64);
            if (lane >= off) incl += n;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 4 * grp; w < wv; w++) base += wsum[w];
        const unsigned excl = base + incl - v;
        if (sp < S) {
            cnt[sp * QSLABS + t] = excl;
            if (slab_start) {
                uint32_t *st = slab_start + ((size_t)fl * S + sp) * (QSLABS + 1);
                st[t] = excl;
                if (t == QSLABS - 1) st[QSLABS] = excl + v;
            }
        }
        __syncthreads();
    }
    // placement, species by species: the records of a species segment are put in slab order in LDS and leave as whole
    // lines (scattered from the threads, 16 bytes each over a frame's 157 kB, lines left the L2 partly written: 1.76 GB
    // written for 0.78 GB of records); a segment longer than the stage (stage_cap records) is scattered directly
    QAtom *__restrict__ Qf = Q + (size_t)fl * N;
    QAtom *stage = reinterpret_cast<QAtom *>(cnt + (size_t)S * QSLABS);
    for (int s = 0; s < S; s++) {
        if (!((used_mask >> s) & 1ull)) continue;
        const int64_t k0 = sp_first[s];
        const int count = (int)(sp_first[s + 1] - k0);
        const bool staged = count <= stage_cap;
#pragma unroll
        for (int j = 0; j < QF_APT; j++) {
            if (((unsigned)(spk >> (4 * j)) & 15u) == (unsigned)s) {
                const unsigned slot = atomicAdd(&cnt[s * QSLABS + (uz[j] >> 24)], 1u);
                QAtom q;
                q.ux = ux[j]; q.uy = uy[j]; q.uz = uz[j]; q.idx = (uint32_t)(tid + j * QUANT_THREADS);
                if (staged) stage[slot] = q;
                else Qf[k0 + slot] = q;
            }
        }
        if (staged) {
            __syncthreads();
            for (int k = tid; k < count; k += QUANT_THREADS) Qf[k0 + k] = stage[k];
            __syncthreads();
        }
    }
}

// 3-D cell variant for the neighbour kernels (cutoffs far below the cell size): one workgroup per (species,
// frame) counting-sorts the species segment by cell = (cz ny + cy) nx + cx (x fastest) with the counters in LDS,
// positions read once (the quantised records wait in LDS between the counting and the placement pass, as in
// quantize_kernel).  start3[fl][sp * ncell + c] = position, inside the frame's sorted array, of the first atom of
// species sp in cell c; start3[fl][S * ncell] = N.  Record idx = species << CELL_SPECIES_SHIFT | atom.
// Components are stored in the cell's own axis order (ux, uy, uz) = axes (0, 1, 2).
__global__ __launch_bounds__(QUANT_THREADS) void quantize_cells_kernel(const double *__restrict__ pos,
                                                                       const double *__restrict__ geom, int n_cells,
                                                                       const int32_t *__restrict__ perm,
                                                                       const int64_t *__restrict__ sp_first, int S,
                                                                       int64_t N, int f0, int nx, int ny, int nz,
                                                                       QAtom *__restrict__ Q, uint32_t *__restrict__ start3,
                                                                       int32_t *flag, int cache_cap,
                                                                       unsigned long long used_mask)
{
    extern __shared__ __align__(16) unsigned char qcache_raw[];
    QAtom *cache = reinterpret_cast<QAtom *>(qcache_raw);                   // [cache_cap]
    unsigned *cnt = reinterpret_cast<unsigned *>(cache + cache_cap);        // [ncell]
    __shared__ unsigned wsum[QUANT_THREADS / 64];
    // blockIdx.x counts the species that take part in the search (used_mask: those with a cutoff to any species);
    // the others are neither sorted nor read by the neighbour kernels
    int sp = 0;
    {
        unsigned long long m = used_mask;
        for (unsigned k = 0; k < blockIdx.x; k++) m &= m - 1ull;
        sp = __ffsll((long long)m) - 1;
    }
    const int fl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = f0 + fl;
    const int ncell = nx * ny * nz;
    const double *__restrict__ g = geom + (size_t)(n_cells == 1 ? 0 : f) * GEOM_STRIDE;
    const int64_t k0 = sp_first[sp], k1 = sp_first[sp + 1];
    const bool cached = k1 - k0 <= (int64_t)cache_cap;
    auto cell_of = [&](const QAtom &q) {
        return (int)((__umulhi(q.uz, (unsigned)nz) * (unsigned)ny + __umulhi(q.uy, (unsigned)ny)) * (unsigned)nx +
                     __umulhi(q.ux, (unsigned)nx));
    };
    for (int c = tid; c < ncell; c += QUANT_THREADS) cnt[c] = 0u;
    __syncthreads();
    for (int64_t k = k0 + tid; k < k1; k += QUANT_THREADS) {
        QAtom q = quantize_atom(pos, g, N, f, perm[k], 0, 1, 2, flag);
        q.idx |= (uint32_t)sp << CELL_SPECIES_SHIFT;
        if (cached) cache[k - k0] = q;
        atomicAdd(&cnt[cell_of(q)], 1u);
    }
    __syncthreads();
    // exclusive scan of the cell counters: one contiguous chunk per thread, chunk totals scanned by waves
    const int chunk = (ncell + QUANT_THREADS - 1) / QUANT_THREADS;
    const int c0 = min(tid * chunk, ncell), c1 = min(c0 + chunk, ncell);
    unsigned s = 0;
    for (int c = c0; c < c1; c++) s += cnt[c];
    unsigned incl = s;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned run = incl - s;
    for (int w = 0; w < wv; w++) run += wsum[w];
    uint32_t *st = start3 + (size_t)fl * ((size_t)S * ncell + 1) + (size_t)sp * ncell;
    for (int c = c0; c < c1; c++) {
        const unsigned v = cnt[c];
        cnt[c] = run;                       // (cursor of the placement pass)
        st[c] = (uint32_t)k0 + run;
        run += v;
    }
    // The neighbour kernels take the upper bound of a species' LAST cell from the entry behind its table, i.e. cell 0 of
    // the next species (or the closing entry): write it here, because the next species may not be sorted at all (it has
    // no block when it carries no cutoff) and the table is scratch that nobody clears.  When the next species IS
    // sorted its block stores the same value (k0' + 0 = k1): a benign same-value race.
    if (tid == 0) st[ncell] = (uint32_t)k1;
    if (blockIdx.x == gridDim.x - 1 && tid == 0) start3[(size_t)fl * ((size_t)S * ncell + 1) + (size_t)S * ncell] = (uint32_t)N;
    __syncthreads();
    QAtom *__restrict__ Qf = Q + (size_t)fl * N + k0;
    for (int64_t k = k0 + tid; k < k1; k += QUANT_THREADS) {
        QAtom q;
        if (cached) {
            q = cache[k - k0];
        } else {
            q = quantize_atom(pos, g, N, f, perm[k], 0, 1, 2, flag);
            q.idx |= (uint32_t)sp << CELL_SPECIES_SHIFT;
        }
        const unsigned slot = atomicAdd(&cnt[cell_of(q)], 1u);
        Qf[slot] = q;
    }
}

// Two-level variant for small cutoffs in big cells: nz coarse slabs along `axis_z` (each at
// least one cutoff thick) x 256 fine bins along `axis_y`.  Key = slab * 256 + ybin; the species
// segment is counting-sorted by key and start2[(fl*S + sp)*(nz*256+1) + key] gives the offset
// of every (slab, ybin) cell inside the segment -- a 2-D cell list.  Stored component order:
// (.ux, .uy, .uz) = (remaining axis, axis_y, axis_z).
__global__ __launch_bounds__(256) void quantize2_kernel(const double *__restrict__ pos,
                                                        const double *__restrict__ geom, int n_cells,
                                                        const int32_t *__restrict__ perm,
                                                        const int64_t *__restrict__ sp_first, int S, int64_t N,
                                                        int f0, int axis_z, int axis_y, int nz,
                                                        QAtom *__restrict__ Q, uint32_t *__restrict__ start2,
                                                        int32_t *flag)
{
    extern __shared__ unsigned cnt2[];          // [nz * 256]
    __shared__ unsigned wsum[4];
    const int sp = blockIdx.x, fl = blockIdx.y, tid = threadIdx.x;
    const int f = f0 + fl;
    const int axis_x = 3 - axis_z - axis_y;
    const int nkeys = nz * 256;
    const double *__restrict__ g = geom + (size_t)(n_cells == 1 ? 0 : f) * GEOM_STRIDE;
    const int64_t k0 = sp_first[sp], k1 = sp_first[sp + 1];
    for (int k = tid; k < nkeys; k += 256) cnt2[k] = 0u;
    __syncthreads();
    for (int64_t k = k0 + tid; k < k1; k += 256) {
        const QAtom q = quantize_atom(pos, g, N, f, perm[k], axis_x, axis_y, axis_z, flag);
        const unsigned key = __umulhi(q.uz, (unsigned)nz) * 256u + (q.uy >> 24);
        atomicAdd(&cnt2[key], 1u);
    }
    __syncthreads();
    // exclusive scan: thread t owns the nz consecutive counters [t*nz, (t+1)*nz)
    unsigned s = 0;
    for (int k = 0; k < nz; k++) s += cnt2[tid * nz + k];
    unsigned incl = s;
    const int lane = tid & 63, wv = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned run = incl - s;
    for (int w = 0; w < wv; w++) run += wsum[w];
    uint32_t *st = start2 + ((size_t)fl * S + sp) * (size_t)(nkeys + 1);
    for (int k = 0; k < nz; k++) {
        const unsigned c = cnt2[tid * nz + k];
        cnt2[tid * nz + k] = run;
        st[tid * nz + k] = run;
        run += c;
    }
    if (tid == 255) st[nkeys] = run;
    __syncthreads();
    QAtom *__restrict__ Qf = Q + (size_t)fl * N + k0;
    for (int64_t k = k0 + tid; k < k1; k += 256) {
        const QAtom q = quantize_atom(pos, g, N, f, perm[k], axis_x, axis_y, axis_z, flag);
        const unsigned key = __umulhi(q.uz, (unsigned)nz) * 256u + (q.uy >> 24);
        const unsigned slot = atomicAdd(&cnt2[key], 1u);
        Qf[slot] = q;
    }
}

// ---- 3-D cell sort (rdf_cell_kernel) -------------------------------------------------------
__device__ __forceinline__ uint32_t cell_key(const QAtom &q, int nx, int ny, int nz, int S, int sp)
{
    const uint32_t cx = __umulhi(q.ux, (unsigned)nx), cy = __umulhi(q.uy, (unsigned)ny), cz = __umulhi(q.uz, (unsigned)nz);
    return ((cz * (uint32_t)ny + cy) * (uint32_t)nx + cx) * (uint32_t)S + (uint32_t)sp;
}

__global__ __launch_bounds__(256) void cell_key_kernel(const double *__restrict__ pos, const double *__restrict__ geom,
                                                       int n_cells, const int32_t *__restrict__ species, int S,
                                                       int64_t N, int f0, int nx, int ny, int nz,
                                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ counts,
                                                       int32_t *flag)
{
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= N) return;
    const int fl = blockIdx.y, f = f0 + fl;
    const double *__restrict__ g = geom + (size_t)(n_cells == 1 ? 0 : f) * GEOM_STRIDE;
    const QAtom q = quantize_atom(pos, g, N, f, a, 0, 1, 2, flag);
    const uint32_t key = cell_key(q, nx, ny, nz, S, species[a]);
    keys[(size_t)fl * N + a] = key;
    atomicAdd(&counts[(size_t)fl * ((size_t)nx * ny * nz * S + 1) + key], 1u);
}

// one workgroup per frame: exclusive scan of the key counts, in place (start3), and a copy for the scatter
__global__ __launch_bounds__(256) void cell_scan_kernel(uint32_t *__restrict__ start3, uint32_t *__restrict__ cursor,
                                                        int nkeys)
{
    __shared__ unsigned wsum[4];
    const int fl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t *st = start3 + (size_t)fl * (nkeys + 1);
    uint32_t *cu = cursor + (size_t)fl * nkeys;
    const int chunk = (nkeys + 255) / 256;
    const int k0 = min(tid * chunk, nkeys), k1 = min(k0 + chunk, nkeys);
    unsigned s = 0;
    for (int k = k0; k < k1; k++) s += st[k];
    unsigned incl = s;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned run = incl - s;
    for (int w = 0; w < wv; w++) run += wsum[w];
    for (int k = k0; k < k1; k++) {
        const unsigned c = st[k];
        st[k] = run;
        cu[k] = run;
        run += c;
    }
    if (tid == 255) st[nkeys] = run;     // (the last thread's chunk ends at nkeys, possibly empty)
}

__global__ __launch_bounds__(256) void cell_scatter_kernel(const double *__restrict__ pos, const double *__restrict__ geom,
                                                           int n_cells, const int32_t *__restrict__ species, int64_t N,
                                                           int f0, int nkeys, const uint32_t *__restrict__ keys,
                                                           uint32_t *__restrict__ cursor, QAtom *__restrict__ Q3)
{
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= N) return;
    const int fl = blockIdx.y, f = f0 + fl;
    const double *__restrict__ g = geom + (size_t)(n_cells == 1 ? 0 : f) * GEOM_STRIDE;
    int32_t dummy = 0;
    QAtom q = quantize_atom(pos, g, N, f, a, 0, 1, 2, &dummy);
    const uint32_t key = keys[(size_t)fl * N + a];
    const unsigned slot = atomicAdd(&cursor[(size_t)fl * nkeys + key], 1u);
    q.idx = ((uint32_t)species[a] << CELL_SPECIES_SHIFT) | (uint32_t)a;
    Q3[(size_t)fl * N + slot] = q;
}

// The same sort in ONE kernel, one workgroup per frame (round 5): the key counters live in LDS as 16-bit halves (configs[4]:
// 45 360 keys = 89 kB), so the three-kernel form's key array (written, read), its global count and cursor atomics and its
// strided one-workgroup scan of the table in global memory are gone: positions are read and quantised twice (counting pass,
// placement pass), the table is written once, the records once.  A thread scans a contiguous run of counter words; an atom's
// place is its run's absolute start (32 bits, cbase) + a 16-bit cursor inside the run -- which requires that no run of ~45
// adjacent keys holds more than 65 535 atoms: checked in the scan; a frame that violates it gets an EMPTY table (every cell
// empty, nothing placed) and raises the flag, and the caller falls back as for far-away atoms.
constexpr int CSF_THREADS = 1024;
constexpr int CSF_AHEAD = 4;        // atoms a thread has in flight

__global__ __launch_bounds__(CSF_THREADS) void cell_sort_frame_kernel(const double *__restrict__ pos, const double *__restrict__ geom,
                                                                      int n_cells, const int32_t *__restrict__ species, int S,
                                                                      int64_t N, int f0, int nx, int ny, int nz,
                                                                      QAtom *__restrict__ Q3, uint32_t *__restrict__ start3,
                                                                      int32_t *flag)
{
    extern __shared__ __align__(16) unsigned char csf_raw[];
    unsigned *cnt = reinterpret_cast<unsigned *>(csf_raw);           // [nwords] two 16-bit counters / cursors per word
    __shared__ unsigned cbase[CSF_THREADS];                          // absolute start of thread t's run of words
    __shared__ unsigned wsum[CSF_THREADS / 64];
    __shared__ int overflow;
    const int fl = blockIdx.x, f = f0 + fl, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nkeys = nx * ny * nz * S, nwords = (nkeys + 1) >> 1;
    const double *__restrict__ g = geom + (size_t)(n_cells == 1 ? 0 : f) * GEOM_STRIDE;
    for (int k = tid; k < nwords; k += CSF_THREADS) cnt[k] = 0u;
    if (tid == 0) overflow = 0;
    __syncthreads();
    // (a workgroup is alone on its CU with 16 waves: CSF_AHEAD atoms per thread are loaded before the first is used, or
    //  every trip waits a whole memory latency -- 1.28 ms per 106 624-atom frame instead of 0.3)
    const double *__restrict__ pf = pos + (size_t)f * (size_t)N * 3;
    auto quant3 = [&](double x, double y, double z, int32_t *fl_out) {
        QAtom q;
        uint32_t u[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {               // (quantize_atom's arithmetic, operation for operation)
            double sx = fma(z, g[15 + c], fma(y, g[12 + c], x * g[9 + c]));
            if (!(fabs(sx) < 1.0e4)) *fl_out = 1;
            sx = sx - floor(sx);
            const double t = sx * 4294967296.0;
            u[c] = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
        }
        q.ux = u[0]; q.uy = u[1]; q.uz = u[2]; q.idx = 0u;
        return q;
    };
    for (int64_t a0 = tid; a0 < N; a0 += (int64_t)CSF_THREADS * CSF_AHEAD) {
        double xs[CSF_AHEAD], ys[CSF_AHEAD], zs[CSF_AHEAD];
        int32_t sps[CSF_AHEAD];
#pragma unroll
        for (int u = 0; u < CSF_AHEAD; u++) {
            const int64_t a = std::min<int64_t>(a0 + (int64_t)u * CSF_THREADS, N - 1);
            xs[u] = pf[3 * a]; ys[u] = pf[3 * a + 1]; zs[u] = pf[3 * a + 2];
            sps[u] = species[a];
        }
#pragma unroll
        for (int u = 0; u < CSF_AHEAD; u++) {
            if (a0 + (int64_t)u * CSF_THREADS < N) {
                const QAtom q = quant3(xs[u], ys[u], zs[u], flag);
                const uint32_t key = cell_key(q, nx, ny, nz, S, sps[u]);
                atomicAdd(&cnt[key >> 1], 1u << ((key & 1u) * 16u));
            }
        }
    }
    __syncthreads();
    // exclusive scan: a contiguous run of words per thread (odd length: consecutive threads on different banks)
    const int chunk = ((nwords + CSF_THREADS - 1) / CSF_THREADS) | 1;
    const int w0 = min(tid * chunk, nwords), w1 = min(w0 + chunk, nwords);
    unsigned s = 0;
    for (int w = w0; w < w1; w++) {
        const unsigned c = cnt[w];
        s += (c & 0xffffu) + (c >> 16);
    }
    // (a half that counted past 65 535 has carried into its neighbour or wrapped: the total of the frame then differs from N)
    if (s > 65535u) overflow = 1;
    unsigned incl = s;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    unsigned run = incl - s;
    for (int w = 0; w < wv; w++) run += wsum[w];
    if (tid == CSF_THREADS - 1 && run + s != (unsigned)N) overflow = 1;
    __syncthreads();
    const bool bad = overflow != 0;
    uint32_t *st = start3 + (size_t)fl * ((size_t)nkeys + 1);
    cbase[tid] = run;
    unsigned local = 0;
    for (int w = w0; w < w1; w++) {
        const unsigned c = cnt[w];
        const unsigned lo = c & 0xffffu, hi = c >> 16;
        cnt[w] = local | ((local + lo) << 16);                       // cursors of the placement pass
        if (2 * w < nkeys) st[2 * w] = bad ? 0u : run + local;
        if (2 * w + 1 < nkeys) st[2 * w + 1] = bad ? 0u : run + local + lo;
        local += lo + hi;
    }
    if (tid == 0) st[nkeys] = bad ? 0u : (uint32_t)N;
    if (bad) {
        if (tid == 0) *flag = 1;
        return;
    }
    __syncthreads();
    QAtom *__restrict__ Qf = Q3 + (size_t)fl * N;
    int32_t dummy = 0;
    for (int64_t a0 = tid; a0 < N; a0 += (int64_t)CSF_THREADS * CSF_AHEAD) {
        double xs[CSF_AHEAD], ys[CSF_AHEAD], zs[CSF_AHEAD];
        int32_t sps[CSF_AHEAD];
#pragma unroll
        for (int u = 0; u < CSF_AHEAD; u++) {
            const int64_t a = std::min<int64_t>(a0 + (int64_t)u * CSF_THREADS, N - 1);
            xs[u] = pf[3 * a]; ys[u] = pf[3 * a + 1]; zs[u] = pf[3 * a + 2];
            sps[u] = species[a];
        }
#pragma unroll
        for (int u = 0; u < CSF_AHEAD; u++) {
            const int64_t a = a0 + (int64_t)u * CSF_THREADS;
            if (a < N) {
                QAtom q = quant3(xs[u], ys[u], zs[u], &dummy);
                const uint32_t sp = (uint32_t)sps[u];
                const uint32_t key = cell_key(q, nx, ny, nz, S, sp);
                const unsigned w = key >> 1, sh = (key & 1u) * 16u;
                const unsigned old = atomicAdd(&cnt[w], 1u << sh);
                const unsigned slot = cbase[w / (unsigned)chunk] + ((old >> sh) & 0xffffu);
                q.idx = (sp << CELL_SPECIES_SHIFT) | (uint32_t)a;
                if (slot < (unsigned)N) Qf[slot] = q;
            }
        }
    }
}

int launch_cell_sort(amof_ctx *ctx, const double *pos_dev, const double *d_geom, int n_cells, const int32_t *d_species,
                     int S, int64_t N, int f0, int nf, int nx, int ny, int nz, QAtom *d_Q3, uint32_t *d_start3,
                     uint32_t *d_keys, uint32_t *d_cursor, int32_t *d_flag)
{
    if (nf <= 0 || N <= 0) return AMOF_OK;
    if (nf > 65535) return fail(ctx, AMOF_ECAPACITY, "frame batch too large");
    const int64_t nkeys64 = (int64_t)nx * ny * nz * S;
    if (nx < 1 || ny < 1 || nz < 1 || nkeys64 > 0x3fffffff || N >= (1ll << CELL_SPECIES_SHIFT) || S > 64)
        return fail(ctx, AMOF_EINVAL, "bad cell grid");
    const int nkeys = (int)nkeys64;
    // one workgroup per frame with the counters in LDS where they fit (and a frame has enough atoms to feed 1024 threads)
    const size_t lds_frame = (size_t)((nkeys + 1) / 2) * sizeof(unsigned);
    if (lds_frame <= 144 * 1024 && N >= 2048 && !getenv("AMOF_CELL_SORT_3K")) {
        AMOF_HIP_TRY(ctx, allow_max_lds((const void *)cell_sort_frame_kernel));
        hipLaunchKernelGGL(cell_sort_frame_kernel, dim3((unsigned)nf), dim3(CSF_THREADS), lds_frame, ctx->stream, pos_dev, d_geom,
                           n_cells, d_species, S, N, f0, nx, ny, nz, d_Q3, d_start3, d_flag);
        AMOF_HIP_TRY(ctx, hipGetLastError());
        return AMOF_OK;
    }
    AMOF_HIP_TRY(ctx, hipMemsetAsync(d_start3, 0, (size_t)nf * (nkeys + 1) * sizeof(uint32_t), ctx->stream));
    dim3 agrid((unsigned)((N + 255) / 256), (unsigned)nf);
    hipLaunchKernelGGL(cell_key_kernel, agrid, dim3(256), 0, ctx->stream, pos_dev, d_geom, n_cells, d_species, S, N, f0,
                       nx, ny, nz, d_keys, d_start3, d_flag);
    hipLaunchKernelGGL(cell_scan_kernel, dim3((unsigned)nf), dim3(256), 0, ctx->stream, d_start3, d_cursor, nkeys);
    hipLaunchKernelGGL(cell_scatter_kernel, agrid, dim3(256), 0, ctx->stream, pos_dev, d_geom, n_cells, d_species, N, f0,
                       nkeys, (const uint32_t *)d_keys, d_cursor, d_Q3);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    return AMOF_OK;
}

int launch_quantize2(amof_ctx *ctx, const double *pos_dev, const double *d_geom, int n_cells, const int32_t *d_perm,
                     const int64_t *d_spfirst, int S, int64_t N, int f0, int nf, int axis_z, int axis_y, int nz,
                     QAtom *d_Q, uint32_t *d_start2, int32_t *d_flag)
{
    if (nf <= 0 || S <= 0) return AMOF_OK;
    if (nf > 65535) return fail(ctx, AMOF_ECAPACITY, "frame batch too large");
    if (nz < 1 || nz > 64 || axis_z == axis_y) return fail(ctx, AMOF_EINVAL, "bad two-level grid");
    dim3 qgrid((unsigned)S, (unsigned)nf);
    hipLaunchKernelGGL(quantize2_kernel, qgrid, dim3(256), (size_t)nz * 256 * sizeof(unsigned), ctx->stream, pos_dev,
                       d_geom, n_cells, d_perm, d_spfirst, S, N, f0, axis_z, axis_y, nz, d_Q, d_start2, d_flag);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    return AMOF_OK;
}

int launch_quantize_cells(amof_ctx *ctx, const double *pos_dev, const double *d_geom, int n_cells, const int32_t *d_perm,
                          const int64_t *d_spfirst, int S, int64_t N, int f0, int nf, int nx, int ny, int nz, QAtom *d_Q,
                          uint32_t *d_start3, int32_t *d_flag, int64_t max_species_atoms, unsigned long long used_mask)
{
    if (nf <= 0 || S <= 0) return AMOF_OK;
    if (S < 64) used_mask &= (1ull << S) - 1ull;
    const int n_used = __builtin_popcountll(used_mask);
    if (n_used == 0) return AMOF_OK;
    if (nf > 65535) return fail(ctx, AMOF_ECAPACITY, "frame batch too large");
    const int64_t ncell = (int64_t)nx * ny * nz;
    if (nx < 1 || ny < 1 || nz < 1 || ncell > CELL_LDS_MAX || N >= (1ll << CELL_SPECIES_SHIFT) || S > 64)
        return fail(ctx, AMOF_EINVAL, "bad cell grid");
    // LDS record cache sized for the largest species segment (up to 4608 records = 72 KiB): with the cell counters
    // beside it, a tight cache is what lets two workgroups share a CU
    const int cache_cap = (int)std::min<int64_t>(std::max<int64_t>(max_species_atoms, 1), 4608);
    const size_t lds = (size_t)cache_cap * sizeof(QAtom) + (size_t)ncell * sizeof(unsigned);
    AMOF_HIP_TRY(ctx, allow_max_lds((const void *)quantize_cells_kernel));
    hipLaunchKernelGGL(quantize_cells_kernel, dim3((unsigned)n_used, (unsigned)nf), dim3(QUANT_THREADS), lds, ctx->stream, pos_dev,
                       d_geom, n_cells, d_perm, d_spfirst, S, N, f0, nx, ny, nz, d_Q, d_start3, d_flag, cache_cap, used_mask);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    return AMOF_OK;
}

int launch_quantize(amof_ctx *ctx, const double *pos_dev, const double *d_geom, int n_cells, const int32_t *d_perm,
                    const int64_t *d_spfirst, int S, int64_t N, int f0, int nf, int axis, QAtom *d_Q,
                    uint32_t *d_slab_start, int32_t *d_flag, int ax0, int ax1, const double *d_fold, unsigned long long used_mask)
{
    if (ax0 < 0 || ax1 < 0) { ax0 = (axis + 1) % 3; ax1 = (axis + 2) % 3; }
    if (nf <= 0 || S <= 0) return AMOF_OK;
    if (nf > 65535) return fail(ctx, AMOF_ECAPACITY, "frame batch too large");
    if (N <= (int64_t)QF_APT * QUANT_THREADS && S <= QF_MAXS && !getenv("AMOF_QUANT_PER_SPECIES")) {
        // one workgroup per frame: the frame is read once, in atom order
        void *d_spec;
        AMOF_TRY(ensure(ctx, SLOT_QSPEC, (size_t)N, &d_spec));
        hipLaunchKernelGGL(species_bytes_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_perm, d_spfirst, S, N,
                           (uint8_t *)d_spec);
        AMOF_HIP_TRY(ctx, hipGetLastError());
        // the stage: 4608 records (72 KiB, as quantize_kernel's cache; the callers do not pass the species counts: N is
        // the bound); AMOF_QUANT_NOSTAGE (experiments): every record scattered from its thread
        const int stage_cap = getenv("AMOF_QUANT_NOSTAGE") ? 0 : (int)std::min<int64_t>(N, 4608);
        const size_t lds = (size_t)S * QSLABS * sizeof(unsigned) + (size_t)stage_cap * sizeof(QAtom);
        AMOF_HIP_TRY(ctx, allow_max_lds((const void *)quantize_frame_kernel));
        hipLaunchKernelGGL(quantize_frame_kernel, dim3((unsigned)nf), dim3(QUANT_THREADS), lds, ctx->stream, pos_dev, d_geom, n_cells,
                           (const uint8_t *)d_spec, d_spfirst, S, N, f0, axis, d_Q, d_slab_start, d_flag, ax0, ax1, d_fold, used_mask,
                           stage_cap);
        AMOF_HIP_TRY(ctx, hipGetLastError());
        return AMOF_OK;
    }
    dim3 qgrid((unsigned)S, (unsigned)nf);
    // LDS record cache: up to 4608 atoms per species segment (72 KiB: two workgroups per CU); longer segments
    // (or species counts unknown here: the cap is only a capacity) take the two-read path
    const int cache_cap = (int)std::min<int64_t>(N, 4608);
    const size_t lds = (size_t)cache_cap * sizeof(QAtom);
    AMOF_HIP_TRY(ctx, allow_max_lds((const void *)quantize_kernel));
    hipLaunchKernelGGL(quantize_kernel, qgrid, dim3(QUANT_THREADS), lds, ctx->stream, pos_dev, d_geom, n_cells, d_perm, d_spfirst,
                       S, N, f0, axis, d_Q, d_slab_start, d_flag, cache_cap, ax0, ax1, d_fold, used_mask);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    return AMOF_OK;
}

}  // namespace amof
