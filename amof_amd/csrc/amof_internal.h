// Internal declarations shared by the translation units of libamofhip.so.
// gfx950 only; compiled with -ffp-contract=off: every fused multiply-add in the
// pair arithmetic is an explicit fma() so that integer results are reproducible
// bit for bit on any conforming IEEE-754 implementation (the CPU oracle).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/amof_hip.h"
#include "guard_math.h"

namespace amof {

// ---------------------------------------------------------------- context --
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

enum Slot {
    SLOT_POS = 0,   // staged positions (host input)
    SLOT_GEOM,      // per-frame geometry records
    SLOT_IMG,       // per-frame extra image vectors
    SLOT_NIMG,      // per-frame extra image counts
    SLOT_PERM,      // species-sorted atom permutation
    SLOT_TILES,     // tile descriptors
    SLOT_PAIRS,     // tile-pair work list
    SLOT_HISTU,     // unordered-key histograms (u64)
    SLOT_SELF,      // self-image histogram (u64)
    SLOT_SPEC,      // species index per atom
    SLOT_QSPEC,     // species of every atom as a byte (quantize_frame_kernel)
    SLOT_AUX0,
    SLOT_AUX1,
    SLOT_AUX2,
    SLOT_AUX3,
    SLOT_AUX4,
    SLOT_AUX5,
    SLOT_AUX6,
    SLOT_AUX7,
    SLOT_AUX8,
    SLOT_AUX9,
    SLOT_OUT0,
    SLOT_OUT1,
    SLOT_FLAGS,
    SLOT_COUNT
};

}  // namespace amof

struct amof_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // host -> device staging that overlaps the kernels of the previous batch
    hipEvent_t ev_copy = nullptr;
    hipEvent_t ev_order = nullptr;       // amof_ctx_wait_stream
    hipEvent_t ev_all0 = nullptr, ev_all1 = nullptr, ev_dom0 = nullptr, ev_dom1 = nullptr;
    bool ev_valid = false;
    int64_t dom_launches = 0;
    std::string err;
    const char *last_path = "";   // kernel family of the last dominant launch (static strings)
    amof::DevBuf buf[amof::SLOT_COUNT];
    // pinned staging ring for the small per-call tables (tile lists, geometry records, work lists ...): a pageable
    // hipMemcpyAsync blocks the host ~40 us per table; from pinned memory it is queued in ~5 us.  The ring is reused
    // from the start after every synchronisation of the context's stream (sync_stream): nothing queued before a
    // synchronisation still reads it.
    unsigned char *pin = nullptr;
    size_t pin_cap = 0, pin_off = 0;
    // page-locked landing area for what the library reads back (flags, result tables): see fetch()
    unsigned char *rb = nullptr;
    size_t rb_cap = 0;
    // amof_msd_shard_begin leaves scratch for amof_msd_shard_finish: valid while no other call ran on the context
    int64_t calls = 0;            // entry points that started device work (timing_begin)
    int64_t progress = 0;         // 2 calls + (the call's dominant kernel is queued): read by OTHER threads (amof_ctx_follow), atomically
    int64_t shard_ticket = 0;     // `calls` right after a begin; 0 = none pending
    int64_t shard_key[7] = {0, 0, 0, 0, 0, 0, 0};
};

namespace amof {

int fail(amof_ctx *ctx, int code, const char *fmt, ...);
// hipStreamSynchronize(ctx->stream) + reuse of the pinned staging ring
hipError_t sync_stream(amof_ctx *ctx);
int ensure(amof_ctx *ctx, Slot s, size_t bytes, void **out);

#define AMOF_HIP_TRY(ctx, expr)                                                              \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return amof::fail((ctx), AMOF_EHIP, "%s failed: %s (%s:%d)", #expr,              \
                              hipGetErrorString(_e), __FILE__, __LINE__);                    \
    } while (0)

#define AMOF_TRY(expr)            \
    do {                          \
        int _rc = (expr);         \
        if (_rc != AMOF_OK) return _rc; \
    } while (0)

// Raise a kernel's dynamic-LDS cap to everything a workgroup may have besides the kernel's static
// LDS.  The value depends on the kernel (and the device) only, not on the launch, so contexts used from different
// threads cannot undercut each other by setting it for the same kernel -- and it is set once per kernel and device:
// the three runtime calls cost ~10 us, per launch, in front of kernels that one rank of eight runs for 9 ms.
// *static_bytes (optional): the kernel's static LDS.
inline hipError_t allow_max_lds(const void *kern, size_t *static_bytes = nullptr)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> done;
    int dev = 0, per_block = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = done.find(std::make_pair(kern, dev));
        if (it != done.end()) {
            if (static_bytes) *static_bytes = it->second;
            return hipSuccess;
        }
    }
    e = hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    if (e != hipSuccess) return e;
    hipFuncAttributes attr;
    e = hipFuncGetAttributes(&attr, kern);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, per_block - (int)attr.sharedSizeBytes);
    if (e != hipSuccess) return e;
    if (static_bytes) *static_bytes = attr.sharedSizeBytes;
    std::lock_guard<std::mutex> lock(mu);
    done[std::make_pair(kern, dev)] = attr.sharedSizeBytes;
    return hipSuccess;
}

// the same for a kernel that takes its dynamic LDS to start at LDS address 0 (rdf_tile_kernel_fast forms histogram addresses
// from integers: bin_count): refused -- loudly, hipErrorInvalidValue -- if the kernel has static LDS in front of it
inline hipError_t allow_max_lds_from_zero(const void *kern)
{
    size_t static_bytes = 0;
    hipError_t e = allow_max_lds(kern, &static_bytes);
    if (e != hipSuccess) return e;
    return static_bytes == 0 ? hipSuccess : hipErrorInvalidValue;
}

// --------------------------------------------------------------- geometry --
// One record per distinct cell: 24 doubles.
//   [0..8]  cell rows          [9..17] cell^-1 with non-periodic columns zeroed
//   [18..20] perpendicular heights   [21] volume   [22] ortho flag   [23] pad
constexpr int GEOM_STRIDE = 24;

struct HostGeom {
    std::vector<double> rec;      // [n_cells][GEOM_STRIDE]
    std::vector<double> invfull;  // [n_cells][9] (host-side use)
    bool all_ortho = true;
    double volume_sum = 0.0;      // over FRAMES (constant cell counted F times)
};

int validate_traj(amof_ctx *ctx, const amof_traj *t, bool need_masses);
int build_geometry(amof_ctx *ctx, const amof_traj *t, HostGeom &g);
// extra periodic images for cutoff R: per cell record a list of lattice vectors
int build_images(amof_ctx *ctx, const amof_traj *t, const HostGeom &g, double R,
                 std::vector<double> &img /* [n_cells][max_img][3] */, std::vector<int32_t> &nimg,
                 int &max_img);

// Error model of the f32 candidate distance of the fast paths (rdf.hip, nbr.hip):
//   q~ = v_sqrt_f32( sum_c d_c^2 ),  d_c = sum_k float(i_k) * s_kc   (s = cell rows * 2^-32 [/ dr])
// With u = 2^-24:  |q~ - q| <= (5 kappa + 3.06) u q, where kappa bounds how much larger than the
// pair vector the summands of d_c can be (cancellation in sheared cells):
//   sum_k |f_k| |C_kc| <= (|d| P)_c,  P = |C^-1| |C|,  kappa = sqrt(||P||_1 ||P||_inf) >= ||P||_2
// (kappa = 1 for a diagonal cell, whose kernels multiply instead of summing: 4.5u + root).
// v_sqrt_f32 on gfx950: <= 1 ulp from correctly rounded, relative error <= 1.56 u, measured
// exhaustively (profiles/tools/sqrt_ulp.hip).  Both bounds carry a further 10 % margin.
// sq_scaled: the diagonal-cell kernel squares the converted differences first and multiplies by the
// squared scales (cvt 1u, square 3u, scale^2 1u, product/fma 5-6-7u on t): 3.5u + root.
inline double fast_guard_rel(const HostGeom &g, int64_t n_cells, bool sq_scaled = false)
{
    const double u = 1.0 / 16777216.0;
    if (g.all_ortho) return 1.1 * ((sq_scaled ? 3.5 : 4.5) + 1.56) * u;
    double kappa = 1.0;
    for (int64_t k = 0; k < n_cells; k++) {
        const double *c = g.rec.data() + (size_t)k * GEOM_STRIDE, *inv = c + 9;
        double P[3][3], n1 = 0.0, ninf = 0.0;
        for (int r = 0; r < 3; r++)
            for (int x = 0; x < 3; x++) {
                P[r][x] = 0.0;
                for (int m = 0; m < 3; m++) P[r][x] += fabs(inv[3 * r + m]) * fabs(c[3 * m + x]);
            }
        for (int r = 0; r < 3; r++) {
            ninf = std::max(ninf, P[r][0] + P[r][1] + P[r][2]);
            n1 = std::max(n1, P[0][r] + P[1][r] + P[2][r]);
        }
        kappa = std::max(kappa, sqrt(n1 * ninf));
    }
    return 1.1 * (5.0 * kappa + 3.06) * u;
}

// relative error bound of the RDF fast paths' f32 chain: diagonal cells as fast_guard_rel(sq_scaled); general cells
// (5 kappa + 3.06) u with kappa of the lower factor, the largest over the six stored axis orders a kernel may pick
inline double fast_guard_rel_rdf(const HostGeom &g, const double *cells, int64_t n_cells)
{
    const double u = 1.0 / 16777216.0;
    if (g.all_ortho) return 1.1 * (3.5 + 1.56) * u;
    double kappa = 1.0;
    static const int perms[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
    for (int64_t k = 0; k < n_cells; k++)
        for (int pm = 0; pm < 6; pm++) {
            const int *ord = perms[pm];
            double rows[9], L[9];
            for (int q = 0; q < 3; q++)
                for (int x = 0; x < 3; x++) rows[3 * q + x] = cells[9 * k + 3 * ord[q] + x];
            lower_factor(rows, L);
            kappa = std::max(kappa, kappa_lower(L));
        }
    return 1.1 * (5.0 * kappa + 3.06) * u;
}

// species-sorted tiling of the atoms
struct Tile {
    int32_t start;    // offset into perm
    int32_t count;
    int32_t species;
    int32_t _pad;
};
struct HostTiles {
    std::vector<int32_t> perm;     // atoms sorted by species (stable)
    std::vector<Tile> tiles;
    std::vector<int64_t> nsp;      // atoms per species
    std::vector<int32_t> sp_first_tile, sp_ntiles;
};
void build_tiles(const amof_traj *t, int tile, HostTiles &out, int granule = 0);

// staging of the position array (host -> device) or pass-through
int stage_positions(amof_ctx *ctx, const amof_traj *t, const double **pos_dev);
// Pipelined variant for kernels that walk the trajectory in frame batches: the device array is
// allocated at once, frames are copied on the context's copy stream only when a batch needs them
// (stager_need), and the compute stream waits on an event -- so the PCIe copy of batch k+1 runs
// while batch k is being computed.  Device-resident and short trajectories degenerate to
// stage_positions.
struct Stager {
    amof_ctx *ctx = nullptr;
    const amof_traj *t = nullptr;
    double *dev = nullptr;
    int64_t upto = 0;       // frames [0, upto) are on the device (or queued ahead of the compute stream)
    bool lazy = false;
};
int stager_begin(amof_ctx *ctx, const amof_traj *t, bool allow_lazy, Stager &st);
int stager_need(Stager &st, int64_t f1);
int upload(amof_ctx *ctx, Slot s, const void *src, size_t bytes, void **out);
// Device -> host of a flag word or a result table, complete on return (the stream is synchronised).  Through page-locked
// memory: with a pageable destination -- a stack variable, the caller's numpy array -- the runtime pins or stages on the
// fly, and the 4-byte flag read after the RDF launch showed as a 0.36 ms copy in the round-5 trace of bench.py.
int fetch(amof_ctx *ctx, void *dst, const void *src_dev, size_t bytes);

// Several small tables in ONE device buffer with ONE host-to-device copy: however small, a copy costs ~5 us of queue time and
// a dozen of them in front of a 0.25 ms kernel were a third of a call.  add() the pieces, upload_pack(), then ptr<T>(i).
struct UploadPack {
    struct Piece {
        const void *src;
        size_t bytes, off;
    };
    std::vector<Piece> pieces;
    size_t total = 0;
    void *base = nullptr;
    int add(const void *src, size_t bytes)
    {
        pieces.push_back(Piece{src, bytes, total});
        total += (bytes + 255) & ~(size_t)255;
        return (int)pieces.size() - 1;
    }
    template <typename T> const T *ptr(int i) const
    {
        return reinterpret_cast<const T *>(static_cast<const unsigned char *>(base) + pieces[(size_t)i].off);
    }
};
int upload_pack(amof_ctx *ctx, Slot s, UploadPack &pk);

// fixed-point atom record of the fast paths: fractional coordinates * 2^32 in the stored
// axis order (slab axis last), original atom index
struct QAtom {
    uint32_t ux, uy, uz, idx;
};
constexpr int QSLABS = 256;

// fold an atom into the cell and quantise its fractional coordinates to 2^-32 (shared by every fast path)
__device__ __forceinline__ QAtom quantize_atom(const double *__restrict__ pos, const double *__restrict__ g,
                                                int64_t N, int f, int64_t a, int ax0, int ax1, int ax2,
                                                int32_t *flag)
{
    const double *__restrict__ p = pos + ((size_t)f * N + a) * 3;
    const double x = p[0], y = p[1], z = p[2];
    uint32_t u[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double s = fma(z, g[15 + c], fma(y, g[12 + c], x * g[9 + c]));
        if (!(fabs(s) < 1.0e4)) *flag = 1;   // absurdly far from the cell (or NaN): caller falls back
        s = s - floor(s);
        double t = s * 4294967296.0;
        u[c] = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    }
    // components are stored in the order (ax0, ax1, ax2): the host puts the slab axis last
    QAtom q;
    q.ux = u[ax0]; q.uy = u[ax1]; q.uz = u[ax2]; q.idx = (uint32_t)a;
    return q;
}

// "Triangular" tile kernel (rdf.hip, TRI): the first two stored coordinates carry the slab coordinate's share of the
// shear, x'' = x + kx z, y' = y + ky z (mod 1; fold = {kx, ky} of the frame's cell), so that the u32 differences of a pair
// are the components of its vector along the orthogonalised lattice directions (see RdfTri in rdf.hip).
__device__ __forceinline__ void fold_atom(QAtom &q, const double *__restrict__ fold)
{
    const double z = (double)q.uz;
    q.ux += (uint32_t)(long long)rint(z * fold[0]);
    q.uy += (uint32_t)(long long)rint(z * fold[1]);
}

// ax0, ax1: the stored order of the two axes that are not the slab axis; d_fold (optional, device [n_cells][2]): fold_atom;
// used_mask: bit s clear = species s (< 64) is neither sorted nor written
int launch_quantize(amof_ctx *ctx, const double *pos_dev, const double *d_geom, int n_cells, const int32_t *d_perm,
                    const int64_t *d_spfirst, int S, int64_t N, int f0, int nf, int axis, QAtom *d_Q,
                    uint32_t *d_slab_start, int32_t *d_flag, int ax0 = -1, int ax1 = -1, const double *d_fold = nullptr,
                    unsigned long long used_mask = ~0ull);

int launch_quantize2(amof_ctx *ctx, const double *pos_dev, const double *d_geom, int n_cells, const int32_t *d_perm,
                     const int64_t *d_spfirst, int S, int64_t N, int f0, int nf, int axis_z, int axis_y, int nz,
                     QAtom *d_Q, uint32_t *d_start2, int32_t *d_flag);

// 3-D cell sort for the cell-list RDF kernel: all atoms of a frame sorted by
// key = ((cz * ny + cy) * nx + cx) * S + species (x fastest),
// d_start3[nf][nkeys + 1] = offsets.
// Q3[.].idx = species << CELL_SPECIES_SHIFT | atom index.  Scratch: d_keys u32 [nf][N],
// d_cursor u32 [nf][nkeys].
constexpr int CELL_SPECIES_SHIFT = 26;
int launch_cell_sort(amof_ctx *ctx, const double *pos_dev, const double *d_geom, int n_cells, const int32_t *d_species,
                     int S, int64_t N, int f0, int nf, int nx, int ny, int nz, QAtom *d_Q3, uint32_t *d_start3,
                     uint32_t *d_keys, uint32_t *d_cursor, int32_t *d_flag);

// 3-D cell sort per species segment for the neighbour kernels (cell counters in LDS: at most CELL_LDS_MAX cells):
// d_Q[nf][N] sorted by (species, cell), d_start3[nf][S * ncell + 1]
constexpr int CELL_LDS_MAX = 16384;
int launch_quantize_cells(amof_ctx *ctx, const double *pos_dev, const double *d_geom, int n_cells, const int32_t *d_perm,
                          const int64_t *d_spfirst, int S, int64_t N, int f0, int nf, int nx, int ny, int nz, QAtom *d_Q,
                          uint32_t *d_start3, int32_t *d_flag, int64_t max_species_atoms, unsigned long long used_mask);

void timing_begin(amof_ctx *ctx);
void timing_end(amof_ctx *ctx);
void timing_dom_begin(amof_ctx *ctx, const char *path);
void timing_dom_end(amof_ctx *ctx, int64_t launches);

// ---------------------------------------------------------------- LDS-DMA --
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// One wave copies 16 B per ACTIVE lane from per-lane global addresses to LDS at dst_wave + 16 * lane
// (dst_wave wave-uniform), without passing through registers (global_load_lds_dwordx4).  Completion is
// tracked by vmcnt: s_waitcnt vmcnt(0) + a barrier before anyone reads the data.
__device__ __forceinline__ void dma16(const void *src_lane, void *dst_wave)
{
    __builtin_amdgcn_global_load_lds((gptr_t)src_lane, (lptr_t)dst_wave, 16, 0, 0);
}

// ------------------------------------------------------- device arithmetic --
// Canonical minimum-image pair vector (see oracle/amof_oracle.c header and
// DESIGN.md "canonical arithmetic").  g points at a geometry record.
template <bool ORTHO>
__device__ __forceinline__ void pair_base(const double *__restrict__ g, double d0x, double d0y,
                                          double d0z, double &dx, double &dy, double &dz)
{
    if (ORTHO) {
        // off-diagonal terms are exact zeros: identical bits to the general form
        double n0 = rint(d0x * g[9]);
        double n1 = rint(d0y * g[13]);
        double n2 = rint(d0z * g[17]);
        dx = fma(-n0, g[0], d0x);
        dy = fma(-n1, g[4], d0y);
        dz = fma(-n2, g[8], d0z);
    } else {
        double s0 = fma(d0z, g[15], fma(d0y, g[12], d0x * g[9]));
        double s1 = fma(d0z, g[16], fma(d0y, g[13], d0x * g[10]));
        double s2 = fma(d0z, g[17], fma(d0y, g[14], d0x * g[11]));
        double n0 = rint(s0), n1 = rint(s1), n2 = rint(s2);
        dx = fma(-n2, g[6], fma(-n1, g[3], fma(-n0, g[0], d0x)));
        dy = fma(-n2, g[7], fma(-n1, g[4], fma(-n0, g[1], d0y)));
        dz = fma(-n2, g[8], fma(-n1, g[5], fma(-n0, g[2], d0z)));
    }
}

__device__ __forceinline__ double norm2(double dx, double dy, double dz)
{
    return fma(dz, dz, fma(dy, dy, dx * dx));
}

}  // namespace amof
