// Window-averaged MSD kernels (gfx950).
//
// Replaces amof.trajectory.get_delta_pos (amof/trajectory.py:285-303, i.e.
// ase wrap_positions(center=0)), WindowMsd.compute_msd_of_m
// (amof/msd.py:185-205) and the centre-of-mass / unwrap preamble of
// amof/msd.py:222-237.
//
// Data flow (all float64):
//   frame-major pos[F][N][3]  --com_kernel-->      com[F][3]
//   pos, com  --delta_transpose_kernel-->          D_T[3N][Fp]   wrapped
//       frame-to-frame displacements, TRANSPOSED to atom-major through an LDS
//       tile so that a column (one coordinate of one atom over time) is
//       contiguous;  Fp = F rounded up to 32
//   D_T --msd_group_kernel-->  partial[group][W]: per column the whole time
//       series sits in LDS (F*8 bytes), is prefix-summed in place (running
//       position u) and every window m reduces sum_k (u[k+m]-u[k])^2 from LDS.
//   partial --msd_reduce_kernel--> sumsq[S][W]  (fixed order: deterministic)
// HBM traffic: pos read twice (com + delta), D_T written once and read once.
//
// Round 4, diagonal cells (the 2-pass form): the centre of mass is folded into the transposition.  wrap() is a shift by
// whole cell vectors, so  wrap((p_k - c_k) - (p_k-1 - c_k-1)) = wrap(wrap(p_k - p_k-1) - (c_k - c_k-1)):  pass 1 reads pos
// ONCE, writes the wrapped RAW differences to D_T and the mass-weighted coordinate sums of its 64-atom tile per frame
// (cpart, 1.5 % of the traffic; summed in fixed order by com_finish_kernel into dc[3][Fp] = c_k - c_k-1); the window
// kernels finish the columns.  Wrapping again changes an entry only when the raw difference lies within |dc| of half the
// cell (an atom that moves half a box in one frame): for every other column  sum_j wrap(raw_j - dc_j) = U_raw[k] - C[k],
// C[k] = c_k (up to a constant that cancels in every window) -- the kernels scan the RAW column and subtract C[k] where an entry enters the registers (no extra
// pass); the scan's first loop checks |raw| against  L/2 - max|dc|,  and a column with such an entry is loaded again,
// corrected entry by entry with the wrap arithmetic (dcom_column) and scanned once more (C replaced by zeros).
// pos read once, D_T written once and read once.
#include <math.h>

#include <algorithm>
#include <vector>

#include "amof_internal.h"

namespace amof {

constexpr int MSD_THREADS = 256;
constexpr int MSD_GEOM = 24;  // cell[9], full inverse[9], pbc[3], pad
constexpr int MSD_GROUP = 4;  // atoms per msd workgroup

// ase.geometry.wrap_positions(d, cell, center=(0,0,0), eps=1e-7) ([3P-memory],
// call site amof/trajectory.py:302): fractional = d.cell^-1 - shift with
// shift = -0.5 - eps; periodic axes: fractional %= 1; fractional += shift;
// result = fractional . cell
__device__ __forceinline__ void wrap_delta(const double *__restrict__ g, double dx, double dy, double dz,
                                           double &ox, double &oy, double &oz)
{
    const double shift = 0.0 - 0.5 - 1e-7;
    double fr[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double s = dx * g[9 + k] + dy * g[12 + k] + dz * g[15 + k];
        if (g[18 + k] != 0.0) {
            double t = s - shift;
            t = t - floor(t);
            s = t + shift;
        }
        fr[k] = s;
    }
    ox = fr[0] * g[0] + fr[1] * g[3] + fr[2] * g[6];
    oy = fr[0] * g[1] + fr[1] * g[4] + fr[2] * g[7];
    oz = fr[0] * g[2] + fr[1] * g[5] + fr[2] * g[8];
}

__device__ __forceinline__ double block_sum(double v, double *red)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < MSD_THREADS / 64; k++) s += red[k];
    return s;  // every thread holds the total
}

// mass-weighted centre of mass of every frame (ase get_center_of_mass:
// masses @ positions / masses.sum(); amof/msd.py:236)
__global__ __launch_bounds__(MSD_THREADS) void com_kernel(const double *__restrict__ pos,
                                                          const double *__restrict__ masses, int64_t N,
                                                          double total_mass, double *__restrict__ com)
{
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    const int f = blockIdx.x;
    const double *__restrict__ p = pos + (size_t)f * (size_t)N * 3;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += MSD_THREADS) {
        double m = masses[i];
        sx += m * p[3 * i];
        sy += m * p[3 * i + 1];
        sz += m * p[3 * i + 2];
    }
    sx = block_sum(sx, red);
    sy = block_sum(sy, red);
    sz = block_sum(sz, red);
    if (threadIdx.x == 0) {
        com[3 * f] = sx / total_mass;
        com[3 * f + 1] = sy / total_mass;
        com[3 * f + 2] = sz / total_mass;
    }
}

// D_T[3a+c][k] = wrap((pos[k][a]-com[k]) - (pos[k-1][a]-com[k-1]); cell[k-1]),  D_T[.][0] = 0
// for the atoms a in [a_begin, a_end) only (atom-sharded calls transpose just their share).
// Tile = TF frames x TA atoms.  A thread owns one atom and FPT = TF TA / THREADS CONSECUTIVE frames: it loads
// the FPT + 1 rows it needs once (every load in flight before the first use), removes the centre of mass once per
// row, wraps the FPT differences and parks them in the LDS tile; the tile leaves transposed, two frames (16 B) per
// lane, TF frames of a column as one contiguous run.  ORTHO: every cell is diagonal -- the zero terms of the
// general formula are dropped, which leaves the bits unchanged (x + (+-0) = x).
template <bool ORTHO>
__device__ __forceinline__ void wrap_delta_t(const double *__restrict__ g, double dx, double dy, double dz,
                                             double &ox, double &oy, double &oz)
{
    if (!ORTHO) {
        wrap_delta(g, dx, dy, dz, ox, oy, oz);
        return;
    }
    const double shift = 0.0 - 0.5 - 1e-7;
    double fr[3] = {dx * g[9], dy * g[13], dz * g[17]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (g[18 + k] != 0.0) {
            double t = fr[k] - shift;
            t = t - floor(t);
            fr[k] = t + shift;
        }
    }
    ox = fr[0] * g[0];
    oy = fr[1] * g[4];
    oz = fr[2] * g[8];
}

// Sum of a double over the 64 lanes of a wave, valid in lane 63; fixed order.  Rows of 16 lanes by DPP moves (row_shr
// 1, 2, 4, 8 with zero fill: vector-ALU instructions -- __shfl_down goes through the LDS crossbar twice per double and
// cost the transposition 0.12 ms), the four row totals by readlane.
__device__ __forceinline__ double wave_sum_dpp(double x)
{
#define AMOF_DPP_STEP(CTRL)                                                                                     \
    {                                                                                                           \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);                  \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);                  \
        x += __hiloint2double(hi, lo);                                                                           \
    }
    AMOF_DPP_STEP(0x111)    // row_shr:1
    AMOF_DPP_STEP(0x112)    // row_shr:2
    AMOF_DPP_STEP(0x114)    // row_shr:4
    AMOF_DPP_STEP(0x118)    // row_shr:8
#undef AMOF_DPP_STEP
    auto lane = [&](int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
    };
    return ((lane(15) + lane(31)) + lane(47)) + x;      // (lane 63 holds the last row's total)
}

// 2-pass form: what a window kernel needs to finish a column of raw wrapped differences (dcT == nullptr: nothing to do)
struct Dcom {
    const double *dcT;      // [3][Fp] centre-of-mass step c_k - c_k-1 per coordinate (entry 0 = 0)
    const double *CT;       // [4][Fp] C[k] = c_k per coordinate (only its differences matter); row 3 = zeros (columns corrected entry by entry)
    const unsigned long long *dcmax;   // [3] bits of max_k |dc_k| per coordinate
    const double *geom;     // [n_cells][MSD_GEOM]
    double lhalf[3];        // smallest L_c (0.5 - 1e-6) over the frames: an entry beyond lhalf - max|dc| may wrap again
    int64_t Fp;
    int32_t n_cells;
    int32_t _pad;
};

// threshold of the raw entries of coordinate comp (see above); +inf without the 2-pass form
__device__ __forceinline__ double dcom_thr(const Dcom &dc, int comp)
{
    return dc.dcT ? dc.lhalf[comp] - __longlong_as_double((long long)dc.dcmax[comp]) : __builtin_inf();
}

// entry k of coordinate `comp`: wrap(x - dc_k) with the cell of frame k - 1, the arithmetic of wrap_delta_t<true>
__device__ __forceinline__ double dcom_fix(const Dcom &dc, int comp, int k, double x)
{
    const double *__restrict__ g = dc.geom + (size_t)(dc.n_cells == 1 || k == 0 ? 0 : k - 1) * MSD_GEOM;
    double v = x - dc.dcT[(size_t)comp * dc.Fp + k];
    if (g[18 + comp] != 0.0) {
        const double shift = 0.0 - 0.5 - 1e-7;
        double t = v * g[9 + 4 * comp] - shift;
        t = t - floor(t);
        v = (t + shift) * g[4 * comp];
    }
    return v;
}

// a column that arrived in LDS by DMA: one coalesced pass (the caller's barrier follows)
template <int T>
__device__ __forceinline__ void dcom_column(const Dcom &dc, int comp, double *u, int F)
{
    for (int k = threadIdx.x; k < F; k += T) u[k] = dcom_fix(dc, comp, k, u[k]);
}

template <int TF, int TA, int THREADS, bool ORTHO>
__global__ __launch_bounds__(THREADS) void delta_transpose_kernel(const double *__restrict__ pos,
                                                                  const double *__restrict__ com,
                                                                  const double *__restrict__ geom,
                                                                  int n_cells, int64_t N, int F, int64_t Fp,
                                                                  int64_t a_begin, int64_t a_end,
                                                                  double *__restrict__ DT,
                                                                  const double *__restrict__ masses, double *__restrict__ cpart)
{
    constexpr int KG = THREADS / TA;      // frame groups of the workgroup
    constexpr int FPT = TF / KG;          // consecutive frames per thread
    static_assert(THREADS % TA == 0 && TF % KG == 0 && TF % 2 == 0, "tile shape");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double (*tile)[3 * TA + 1] = reinterpret_cast<double (*)[3 * TA + 1]>(lds_raw);   // [TF][3 TA + 1]
    const int k0 = blockIdx.y * TF;
    const int64_t a0 = a_begin + (int64_t)blockIdx.x * TA;
    const int al = threadIdx.x % TA, kg = threadIdx.x / TA;
    const int64_t a = a0 + al;
    const int kb = k0 + kg * FPT;         // this thread's frames: kb .. kb + FPT - 1 (row kb - 1 is needed too)
    const bool atom_ok = a < a_end;
    double px[FPT + 1], py[FPT + 1], pz[FPT + 1];
#pragma unroll
    for (int j = 0; j <= FPT; j++) {
        const int k = kb - 1 + j;
        px[j] = py[j] = pz[j] = 0.0;
        if (atom_ok && k >= 0 && k < F) {
            const double *p = pos + ((size_t)k * N + a) * 3;
            px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
        }
    }
    if (com) {
#pragma unroll
        for (int j = 0; j <= FPT; j++) {
            const int k = kb - 1 + j;
            if (k >= 0 && k < F) {
                px[j] -= com[3 * k]; py[j] -= com[3 * k + 1]; pz[j] -= com[3 * k + 2];
            }
        }
    }
    if (cpart) {
        // 2-pass form: the tile's share of sum_a m_a p_a for this thread's own frames (a wave = the TA = 64 atoms of one
        // frame group): lanes by shuffles in fixed order, one row of cpart[tile][F][3] per frame
        static_assert(TA == 64, "one wave per frame group");
        const double m = atom_ok ? masses[a] : 0.0;
#pragma unroll
        for (int j = 1; j <= FPT; j++) {
            const int k = kb - 1 + j;
            const double sx = wave_sum_dpp(m * px[j]), sy = wave_sum_dpp(m * py[j]), sz = wave_sum_dpp(m * pz[j]);
            if (al == 63 && k < F) {
                double *o = cpart + ((size_t)blockIdx.x * (size_t)F + (size_t)k) * 3;
                o[0] = sx; o[1] = sy; o[2] = sz;
            }
        }
    }
#pragma unroll
    for (int j = 1; j <= FPT; j++) {
        const int k = kb - 1 + j;
        double dx = 0.0, dy = 0.0, dz = 0.0;
        if (atom_ok && k >= 1 && k < F) {
            const double *g = geom + (size_t)(n_cells == 1 ? 0 : k - 1) * MSD_GEOM;
            wrap_delta_t<ORTHO>(g, px[j] - px[j - 1], py[j] - py[j - 1], pz[j] - pz[j - 1], dx, dy, dz);
        }
        const int kl = kg * FPT + j - 1;
        tile[kl][3 * al] = dx;
        tile[kl][3 * al + 1] = dy;
        tile[kl][3 * al + 2] = dz;
    }
    __syncthreads();
#pragma unroll 3
    for (int idx = threadIdx.x; idx < (TF / 2) * 3 * TA; idx += THREADS) {
        const int cl = idx / (TF / 2), kl = 2 * (idx % (TF / 2));
        const int64_t col = 3 * a0 + cl;
        // (k0 + kl is even and Fp a multiple of 32: the pair is inside the column or wholly outside)
        if (col < 3 * a_end && k0 + kl < Fp)
            *reinterpret_cast<double2 *>(DT + (size_t)col * Fp + k0 + kl) = make_double2(tile[kl][cl], tile[kl + 1][cl]);
    }
}

// 2-pass form: c_k = (sum over the atom tiles of cpart[t][k]) / M in tile order, dc[comp][k] = c_k - c_k-1 (dc[.][0] = 0)
constexpr int COMF_FRAMES = 64, COMF_GROUPS = MSD_THREADS / COMF_FRAMES;
__global__ __launch_bounds__(MSD_THREADS) void com_finish_kernel(const double *__restrict__ cpart, int ntiles, int F, int64_t Fp,
                                                                 double total_mass, double *__restrict__ dcT,
                                                                 double *__restrict__ CT, unsigned long long *__restrict__ dcmax)
{
    // a block = 64 frames (and the one before them); four thread groups share the tiles (contiguous quarters, eight loads
    // in flight per thread: a serial walk over 153 tiles paid a memory latency per tile, 0.15 ms on 20 workgroups)
    __shared__ double part[COMF_GROUPS][COMF_FRAMES + 1][3];
    __shared__ double cs[COMF_FRAMES + 1][3];
    const int k0 = blockIdx.x * COMF_FRAMES;
    const int g = threadIdx.x / COMF_FRAMES, kk = threadIdx.x % COMF_FRAMES;
    const int per = (ntiles + COMF_GROUPS - 1) / COMF_GROUPS, t0 = min(g * per, ntiles), t1 = min(t0 + per, ntiles);
    for (int q = kk; q < COMF_FRAMES + 1; q += COMF_FRAMES) {
        const int k = k0 - 1 + q;       // frames k0 - 1 .. k0 + 63
        double sx = 0.0, sy = 0.0, sz = 0.0;
        if (k >= 0 && k < F) {
            for (int t = t0; t < t1; t += 8) {
                double v[8][3];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const double *__restrict__ c = cpart + ((size_t)min(t + e, t1 - 1) * (size_t)F + (size_t)k) * 3;
                    v[e][0] = c[0]; v[e][1] = c[1]; v[e][2] = c[2];
                }
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (t + e < t1) { sx += v[e][0]; sy += v[e][1]; sz += v[e][2]; }
            }
        }
        part[g][q][0] = sx; part[g][q][1] = sy; part[g][q][2] = sz;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < (COMF_FRAMES + 1) * 3; q += MSD_THREADS) {
        const int f = q / 3, c = q % 3;
        double sum = 0.0;
        for (int gg = 0; gg < COMF_GROUPS; gg++) sum += part[gg][f][c];
        cs[f][c] = sum / total_mass;
    }
    __syncthreads();
    if (threadIdx.x < COMF_FRAMES) {
        const int k = k0 + threadIdx.x;
        if (k < F) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double d = k == 0 ? 0.0 : cs[threadIdx.x + 1][c] - cs[threadIdx.x][c];
                dcT[(size_t)c * Fp + k] = d;
                CT[(size_t)c * Fp + k] = cs[threadIdx.x + 1][c];      // (only differences C[k + m] - C[k] enter the sums)
                atomicMax(&dcmax[c], (unsigned long long)__double_as_longlong(fabs(d)));      // (non-negative doubles order as integers)
            }
            CT[(size_t)3 * Fp + k] = 0.0;
        }
    }
}

// in-place inclusive prefix sum of u[0..F) held in LDS by the whole workgroup: every thread
// owns one contiguous chunk (serial sum, then serial rewrite), the 256 chunk totals are scanned
// with wave shuffles -- two barriers per column instead of two per 256 elements.
// thr / evt (2-pass form): evt is set for the whole workgroup when some entry of the column exceeds thr in magnitude
__device__ __forceinline__ void lds_scan(double *u, int F, double *wtot, double carry_init, double thr = __builtin_inf(),
                                         bool *evt = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // odd chunk length: lanes then start an odd number of doubles apart, so the strided LDS accesses of a
    // wave spread over all banks (an even stride such as 20 doubles is an 8-way conflict)
    const int chunk = ((F + MSD_THREADS - 1) / MSD_THREADS) | 1;
    const int k0 = min(tid * chunk, F), k1 = min(k0 + chunk, F);
    double s = 0.0;
    bool big = false;
    for (int k = k0; k < k1; k++) {
        const double x = u[k];
        big |= fabs(x) > thr;
        s += x;
    }
    const bool wbig = evt ? __any(big) != 0 : false;
    double v = s;                                   // inclusive scan of the chunk totals
    for (int off = 1; off < 64; off <<= 1) {
        double n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    __syncthreads();
    if (lane == 63) {
        wtot[wv] = v;
        if (evt) wtot[MSD_THREADS / 64 + wv] = wbig ? 1.0 : 0.0;       // (the flags ride behind the totals: no barrier of their own)
    }
    __syncthreads();
    if (evt) {
        double any = 0.0;
        for (int q = 0; q < MSD_THREADS / 64; q++) any += wtot[MSD_THREADS / 64 + q];
        *evt = any > 0.0;
    }
    double run = carry_init + (v - s);              // exclusive prefix of this thread's chunk
    for (int q = 0; q < wv; q++) run += wtot[q];
    for (int k = k0; k < k1; k++) {
        run += u[k];
        u[k] = run;
    }
    __syncthreads();
}

template <int T>
__device__ __forceinline__ void lds_scan_t(double *u, int F, double *wtot, double thr = __builtin_inf(), bool *evt = nullptr);

// prefix sums of a column that sits in LDS (barrier done).  2-pass form: the raw column is scanned and C[k] subtracted --
// unless an entry could wrap again under the centre-of-mass step (rare: see the header); then `reload` brings the raw
// column back (its own barriers) and it is corrected entry by entry before the scan.
template <int T, typename Reload>
__device__ __forceinline__ void scan_column(const Dcom &dc, int comp, double *u, int F, double *red, Reload &&reload)
{
    if (!dc.dcT) {
        lds_scan_t<T>(u, F, red, __builtin_inf(), nullptr);
        return;
    }
    bool evt = false;
    lds_scan_t<T>(u, F, red, dcom_thr(dc, comp), &evt);
    if (evt) {
        reload();
        dcom_column<T>(dc, comp, u, F);
        __syncthreads();
        lds_scan_t<T>(u, F, red, __builtin_inf(), nullptr);
    } else {
        const double *__restrict__ cb = dc.CT + (size_t)comp * dc.Fp;
        for (int k = threadIdx.x; k < F; k += T) u[k] -= cb[k];
        __syncthreads();
    }
}

struct MsdGroup {
    int32_t start;    // into perm
    int32_t count;    // atoms
    int32_t species;
    int32_t _pad;
};

// one workgroup = one group of <= MSD_GROUP atoms of one species.
// WREG > 0: at most WREG windows, per-thread partial sums stay in registers across all the
// columns of the group and are reduced over the workgroup once per window;
// WREG == 0: any number of windows, one workgroup reduction per (column, window).
template <int WREG>
__global__ __launch_bounds__(MSD_THREADS) void msd_group_kernel(const double *__restrict__ DT, int64_t Fp, int F,
                                                                const int32_t *__restrict__ perm,
                                                                const MsdGroup *__restrict__ groups,
                                                                const int32_t *__restrict__ windows, int W,
                                                                double *__restrict__ partial, Dcom dc)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *u = reinterpret_cast<double *>(lds_raw);  // [F]
    double *wsum = u + F;                             // [W]
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    const int tid = threadIdx.x;
    const MsdGroup gr = groups[blockIdx.x];
    double acc[WREG > 0 ? WREG : 1];
#pragma unroll
    for (int w = 0; w < (WREG > 0 ? WREG : 1); w++) acc[w] = 0.0;
    if (WREG == 0)
        for (int w = tid; w < W; w += MSD_THREADS) wsum[w] = 0.0;
    for (int c = 0; c < 3 * gr.count; c++) {
        const int64_t atom = perm[gr.start + c / 3];
        const double *__restrict__ col = DT + (size_t)(3 * atom + c % 3) * Fp;
        __syncthreads();
        // columns start 256-B aligned and are padded to a multiple of 32 frames: 16-B loads
        auto load_raw = [&]() {
            for (int k = 2 * tid; k < F; k += 2 * MSD_THREADS) {
                const double2 v2 = *reinterpret_cast<const double2 *>(col + k);
                u[k] = v2.x;
                if (k + 1 < F) u[k + 1] = v2.y;
            }
            __syncthreads();
        };
        load_raw();
        scan_column<MSD_THREADS>(dc, c % 3, u, F, red, [&]() { __syncthreads(); load_raw(); });
        if (WREG > 0) {
#pragma unroll
            for (int w = 0; w < WREG; w++) {
                if (w < W) {
                    const int m = windows[w];
                    double a = acc[w];
                    for (int k = 1 + tid; k + m < F; k += MSD_THREADS) {
                        double d = u[k + m] - u[k];
                        a = fma(d, d, a);
                    }
                    acc[w] = a;
                }
            }
        } else {
            for (int w = 0; w < W; w++) {
                const int m = windows[w];
                double a = 0.0;
                for (int k = 1 + tid; k + m < F; k += MSD_THREADS) {
                    double d = u[k + m] - u[k];
                    a = fma(d, d, a);
                }
                a = block_sum(a, red);
                if (tid == 0) wsum[w] += a;
            }
        }
    }
    if (WREG > 0) {
#pragma unroll
        for (int w = 0; w < WREG; w++) {
            if (w < W) {
                const double tot = block_sum(acc[w], red);
                if (tid == 0) partial[(size_t)blockIdx.x * W + w] = tot;
            }
        }
    } else {
        __syncthreads();
        for (int w = tid; w < W; w += MSD_THREADS) partial[(size_t)blockIdx.x * W + w] = wsum[w];
    }
}

// Windows in arithmetic progression m_w = w * d (what WindowMsd always passes: window =
// arange(0, max, delta_m), amof/msd.py:180).  The pairs (k, k + w d) of one residue class
// r = k mod d live on the "comb" u[r], u[r + d], u[r + 2d], ...: a thread loads COMB_B + WT - 1
// consecutive comb entries once and forms every pair (j, j + w), j in its COMB_B bases, w < WT,
// from registers -- about 7 terms per LDS read instead of one term per two reads, which is what
// bounded msd_group_kernel (LDS bandwidth).  Windows w >= W of the template bucket WT are
// computed and dropped.  Same column load and prefix sum as msd_group_kernel.
constexpr int COMB_B = 10;

// One comb task: the COMB_B base entries u[r + d (J0 + i)] and their WT - 1 successors, every pair (i, e = i + w),
// w = 1 .. WT - 1, from registers.  `lim` = number of valid comb entries from J0 on; m0 = 0 for the task that holds
// time origin k = 0, which the reference never evaluates (amof/msd.py:200: k starts at m + 1), else 1.
// (Measured and rejected on the MI355X: keeping the lane-mask compares next to their uses and splitting off the
// origin task -- 22 % fewer instructions, yet 10 % slower: the kernel is bound by dependency stalls at 2-3 waves per
// SIMD, not by issue; 512-thread workgroups whose halves split the windows -- 4 waves per SIMD but spills, 0.84 ms;
// forcing 4 waves per SIMD by launch bounds (128 VGPRs, spills): 0.90 ms; COMB_B = 8 / 6 / 5: 0.77 / 1.00 / 0.88 ms.)
template <int WT>
__device__ __forceinline__ void comb_task(const double *__restrict__ ub, int d, int lim, double m0, double (&acc)[WT])
{
    constexpr int NV = COMB_B + WT - 1;
    double v[NV];
#pragma unroll
    for (int e = 0; e < NV; e++) v[e] = e < lim ? ub[(size_t)d * e] : 0.0;
#pragma unroll
    for (int e = 1; e < NV; e++) {
        if (e < lim) {
#pragma unroll
            for (int i = (e - WT + 1 > 0 ? e - WT + 1 : 0); i <= (e - 1 < COMB_B - 1 ? e - 1 : COMB_B - 1); i++) {
                const double dd = v[e] - v[i];
                acc[e - i] = fma(i == 0 ? dd * m0 : dd, dd, acc[e - i]);
            }
        }
    }
}

template <int WT>
__global__ __launch_bounds__(MSD_THREADS) void msd_comb_kernel(const double *__restrict__ DT, int64_t Fp, int F,
                                                               const int32_t *__restrict__ perm,
                                                               const MsdGroup *__restrict__ groups, int d, int W,
                                                               int Wstride, double *__restrict__ partial, Dcom dc)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *u = reinterpret_cast<double *>(lds_raw);  // [F]
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    const int tid = threadIdx.x;
    const MsdGroup gr = groups[blockIdx.x];
    const int nq = (F + d - 1) / d;                    // longest comb
    const int ntask = d * ((nq + COMB_B - 1) / COMB_B);
    double acc[WT];
#pragma unroll
    for (int w = 0; w < WT; w++) acc[w] = 0.0;
    for (int c = 0; c < 3 * gr.count; c++) {
        const int64_t atom = perm[gr.start + c / 3];
        const double *__restrict__ col = DT + (size_t)(3 * atom + c % 3) * Fp;
        __syncthreads();
        auto load_raw = [&]() {
            for (int k = 2 * tid; k < F; k += 2 * MSD_THREADS) {
                const double2 v2 = *reinterpret_cast<const double2 *>(col + k);
                u[k] = v2.x;
                if (k + 1 < F) u[k + 1] = v2.y;
            }
            __syncthreads();
        };
        load_raw();
        scan_column<MSD_THREADS>(dc, c % 3, u, F, red, [&]() { __syncthreads(); load_raw(); });
        for (int t = tid; t < ntask; t += MSD_THREADS) {
            const int r = t % d, J0 = (t / d) * COMB_B;
            const int lim = (F - r + d - 1) / d - J0;   // valid comb entries of this task (from J0 on)
            comb_task<WT>(u + r + (size_t)d * J0, d, lim, t == 0 ? 0.0 : 1.0, acc);
        }
    }
#pragma unroll
    for (int w = 0; w < WT; w++) {
        if (w < W) {
            const double tot = block_sum(acc[w], red);
            if (tid == 0) partial[(size_t)blockIdx.x * Wstride + w] = tot;
        }
    }
}

// The same with the column loads off the critical path: two LDS buffers, the next column streams in by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave instruction, no registers) while the current one is scanned and
// combed -- one barrier per column hands the buffers over.  Used when two columns fit (2 Fp doubles).
template <int WT>
__global__ __launch_bounds__(MSD_THREADS) void msd_comb_db_kernel(const double *__restrict__ DT, int64_t Fp, int F,
                                                                  const int32_t *__restrict__ perm,
                                                                  const MsdGroup *__restrict__ groups, int d, int W,
                                                                  int Wstride, double *__restrict__ partial, Dcom dc)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *ubuf = reinterpret_cast<double *>(lds_raw);  // [2][Fp]
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const MsdGroup gr = groups[blockIdx.x];
    const int nq = (F + d - 1) / d;                    // longest comb
    const int ntask = d * ((nq + COMB_B - 1) / COMB_B);
    const int ncol = 3 * gr.count;
    const int ngran = (int)(Fp / 2);                   // 16-byte granules per column (Fp is a multiple of 32)
    // column c -> buffer b: wave w moves the KiB blocks w, w + 4, ...; lanes beyond the column stay idle
    auto issue = [&](int c, int b) {
        const int64_t atom = perm[gr.start + c / 3];
        const double *__restrict__ col = DT + (size_t)(3 * atom + c % 3) * Fp;
        unsigned char *dst = reinterpret_cast<unsigned char *>(ubuf + (size_t)b * Fp);
        for (int blk = wave; blk * 64 < ngran; blk += MSD_THREADS / 64) {
            const int gran = blk * 64 + lane;
            if (gran < ngran) dma16(col + 2 * gran, dst + (size_t)blk * 1024);
        }
    };
    double acc[WT];
#pragma unroll
    for (int w = 0; w < WT; w++) acc[w] = 0.0;
    if (ncol > 0) issue(0, 0);
    for (int c = 0; c < ncol; c++) {
        double *u = ubuf + (size_t)(c & 1) * Fp;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of column c has landed
        __syncthreads();                                    // everyone's has; column c - 1 is fully consumed
        if (c + 1 < ncol) issue(c + 1, (c + 1) & 1);        // streams in behind the scan and the comb arithmetic
        scan_column<MSD_THREADS>(dc, c % 3, u, F, red, [&]() {
            __syncthreads();
            issue(c, c & 1);                                // (rare) the raw column again, behind the next one
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        });
        for (int t = tid; t < ntask; t += MSD_THREADS) {
            const int r = t % d, J0 = (t / d) * COMB_B;
            const int lim = (F - r + d - 1) / d - J0;   // valid comb entries of this task (from J0 on)
            comb_task<WT>(u + r + (size_t)d * J0, d, lim, t == 0 ? 0.0 : 1.0, acc);
        }
    }
#pragma unroll
    for (int w = 0; w < WT; w++) {
        if (w < W) {
            const double tot = block_sum(acc[w], red);
            if (tid == 0) partial[(size_t)blockIdx.x * Wstride + w] = tot;
        }
    }
}

// Streaming form of the comb arithmetic for window spacings d of 64 .. 256 frames (WindowMsd's default is 100).
// One thread owns one residue class r of the column and walks its comb x_e = u[r + d e], e = 0, 1, ...: the last L
// entries live in a register ring, every new entry forms its L pairs (lags 1 .. L) against them -- one LDS read per
// L terms, no lane masks in the steady state (the block form above spends 45 % of its vector instructions on masks,
// addresses and scalar-register spills: PMC, profiles/r02/pmc_msd.json), every thread the same amount of work.
// Workgroup = roundup64(d) threads and ONE column buffer (F doubles): four workgroups per CU overlap each other's
// column load, scan and arithmetic.  The time origin k = 0 (r = 0, e = 0) is left out as in the reference
// (amof/msd.py:200); lags beyond W - 1 are computed and dropped.
template <int T>
__device__ __forceinline__ void lds_scan_t(double *u, int F, double *wtot, double thr, bool *evt)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chunk = ((F + T - 1) / T) | 1;        // odd: strided LDS accesses of a wave spread over the banks
    const int k0 = min(tid * chunk, F), k1 = min(k0 + chunk, F);
    double s = 0.0;
    bool big = false;
    for (int k = k0; k < k1; k++) {
        const double x = u[k];
        big |= fabs(x) > thr;
        s += x;
    }
    const bool wbig = evt ? __any(big) != 0 : false;
    double v = s;
    for (int off = 1; off < 64; off <<= 1) {
        double n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    __syncthreads();
    if (lane == 63) {
        wtot[wv] = v;
        if (evt) wtot[T / 64 + wv] = wbig ? 1.0 : 0.0;
    }
    __syncthreads();
    if (evt) {
        double any = 0.0;
        for (int q = 0; q < T / 64; q++) any += wtot[T / 64 + q];
        *evt = any > 0.0;
    }
    double run = v - s;
    for (int q = 0; q < wv; q++) run += wtot[q];
    for (int k = k0; k < k1; k++) {
        run += u[k];
        u[k] = run;
    }
    __syncthreads();
}

// one block of L comb entries (e = e0 .. e0 + L - 1; ring slot of entry e is e mod L, e0 a multiple of L).
// BLK = 0: the block that starts the comb (e0 = 0): entry e only has e predecessors; BLK = 1: the second block
// (e0 = L); BLK = 2: any later one.  A pair with entry 0 -- (0, s) in the first block, (0, L) at the start of the
// second -- carries the factor m0 (0 for the residue class that holds the time origin).  GUARD: entries >= nq do not exist.
template <int L, int BLK, bool GUARD>
__device__ __forceinline__ void stream_block(const double *__restrict__ ub, int d, int e0, int nq, double m0,
                                             double (&ring)[L], double (&acc)[L + 1])
{
#pragma unroll
    for (int s = 0; s < L; s++) {
        if (!GUARD || e0 + s < nq) {
            const double x = ub[(size_t)d * (e0 + s)];
#pragma unroll
            for (int w = 1; w <= L; w++) {
                if (!(BLK == 0 && w > s)) {                     // (compile time)
                    const double dd = x - ring[(s - w + 4 * L) % L];
                    const bool with_origin = (BLK == 0 && w == s) || (BLK == 1 && s == 0 && w == L);
                    acc[w] = fma(with_origin ? dd * m0 : dd, dd, acc[w]);
                }
            }
            ring[s] = x;
        }
    }
}

// The scan in registers (round 4).  Stamps inside the kernel (profiles/r04/msd_experiments.txt) showed the in-place scan at
// 5 - 6 us of a 20 us column: its second loop reads and writes the same LDS array, and every iteration waited for the LDS.
// Here a thread reads its chunk (<= CH entries) at once, adds it up in registers and writes it back at once -- with, in the
// 2-pass form, the centre-of-mass step subtracted on the way: v[] arrives holding the thread's chunk of dc (loaded from
// global memory while the column's DMA is in flight; the sums are linear: scanning raw - dc gives U_raw - C), one register
// array for both.  Returns "some entry could wrap again under the centre-of-mass step" (then u is left RAW).
template <int T, int CH, bool FOLD>
__device__ __forceinline__ bool reg_scan(double *u, int F, int chunk, double *wtot, double thr, double (&v)[CH])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k0 = min(tid * chunk, F), n = min(k0 + chunk, F) - k0;
    bool big = false;
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const double x = i < n ? u[k0 + i] : 0.0;
        if (FOLD) {
            big |= fabs(x) > thr;
            v[i] = x - v[i];
        } else {
            v[i] = x;
        }
    }
#pragma unroll
    for (int i = 1; i < CH; i++) v[i] += v[i - 1];
    const double s = v[CH - 1];
    const bool wbig = FOLD ? __any(big) != 0 : false;
    double incl = s;
    for (int off = 1; off < 64; off <<= 1) {
        const double nb = __shfl_up(incl, off, 64);
        if (lane >= off) incl += nb;
    }
    __syncthreads();
    if (lane == 63) {
        wtot[wv] = incl;
        if (FOLD) wtot[T / 64 + wv] = wbig ? 1.0 : 0.0;
    }
    __syncthreads();
    bool evt = false;
    if (FOLD) {
        double any = 0.0;
        for (int q = 0; q < T / 64; q++) any += wtot[T / 64 + q];
        evt = any > 0.0;
    }
    double off = incl - s;
    for (int q = 0; q < wv; q++) off += wtot[q];
    if (!evt) {
#pragma unroll
        for (int i = 0; i < CH; i++)
            if (i < n) u[k0 + i] = v[i] + off;
    }
    __syncthreads();
    return evt;
}
constexpr int STREAM_CH = 42;       // register scan: chunks of up to 42 entries per thread (F <= 41 T: 5248 frames at 128 threads)

template <int L, int T, bool FOLD>
__global__ __launch_bounds__(T, 2) void msd_stream_kernel(const double *__restrict__ DT, int64_t Fp, int F,
                                                       const int32_t *__restrict__ perm,
                                                       const MsdGroup *__restrict__ groups, int d, int W, int Wstride,
                                                       double *__restrict__ partial, Dcom dc)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *u = reinterpret_cast<double *>(lds_raw);  // [Fp]
    __shared__ double red[2 * (T / 64)];     // (wave totals | event flags of the scan)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const MsdGroup gr = groups[blockIdx.x];
    const int ncol = 3 * gr.count;
    const int ngran = (int)(Fp / 2);                   // 16-byte granules per column
    const bool active = tid < d;                       // thread = residue class
    const int r = active ? tid : 0;
    const int nq = active ? (F - r + d - 1) / d : 0;   // entries of this thread's comb
    const int nq_min = F / d;                          // every comb has at least this many
    const double m0 = r == 0 ? 0.0 : 1.0;
    const int chunk = ((F + T - 1) / T) | 1;           // entries per thread in the scan (odd: LDS banks)
    const bool regscan = chunk <= STREAM_CH;
    double acc[L + 1], ring[L];
#pragma unroll
    for (int w = 0; w <= L; w++) acc[w] = 0.0;
    // (measured and rejected: the next column prefetched through 40 registers per thread while this one is combed --
    //  253 VGPRs, one wave per SIMD, 0.63 instead of 0.55 ms)
    for (int c = 0; c < ncol; c++) {
        const int64_t atom = perm[gr.start + c / 3];
        const double *__restrict__ col = DT + (size_t)(3 * atom + c % 3) * Fp;
        __syncthreads();                               // the previous column is fully consumed
        for (int blk = wave; blk * 64 < ngran; blk += T / 64) {
            const int gran = blk * 64 + lane;
            if (gran < ngran) dma16(col + 2 * gran, reinterpret_cast<unsigned char *>(u) + (size_t)blk * 1024);
        }
        // (2-pass form) the thread's chunk of the centre-of-mass steps of this coordinate: in flight together with the column
        double cc[STREAM_CH];
        if (FOLD && regscan) {
            const double *__restrict__ crow = dc.dcT + (size_t)(c % 3) * dc.Fp;
            const int k0c = min(tid * chunk, F), nc = min(k0c + chunk, F) - k0c;
#pragma unroll
            for (int i = 0; i < STREAM_CH; i++) cc[i] = i < nc ? crow[k0c + i] : 0.0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (regscan) {
            const bool evt = reg_scan<T, STREAM_CH, FOLD>(u, F, chunk, red, FOLD ? dcom_thr(dc, c % 3) : 0.0, cc);
            if (FOLD && evt) {      // (rare) an entry could wrap again under the centre-of-mass step: the raw column (still in
                dcom_column<T>(dc, c % 3, u, F);        //  LDS) entry by entry with the wrap arithmetic, then the plain scan
                __syncthreads();
                lds_scan_t<T>(u, F, red);
            }
        } else {
            lds_scan_t<T>(u, F, red);       // (FOLD is only launched when the register scan applies)
        }
        if (active) {
            const double *__restrict__ ub = u + r;
#pragma unroll
            for (int k = 0; k < L; k++) ring[k] = 0.0;
            int e0 = 0;
            if (nq_min >= 2 * L) {
                stream_block<L, 0, false>(ub, d, 0, nq, m0, ring, acc);
                stream_block<L, 1, false>(ub, d, L, nq, m0, ring, acc);
                for (e0 = 2 * L; e0 + L <= nq_min; e0 += L) stream_block<L, 2, false>(ub, d, e0, nq, m0, ring, acc);
                for (; e0 < nq; e0 += L) stream_block<L, 2, true>(ub, d, e0, nq, m0, ring, acc);
            } else {
                stream_block<L, 0, true>(ub, d, 0, nq, m0, ring, acc);
                if (L < nq) stream_block<L, 1, true>(ub, d, L, nq, m0, ring, acc);
                for (e0 = 2 * L; e0 < nq; e0 += L) stream_block<L, 2, true>(ub, d, e0, nq, m0, ring, acc);
            }
        }
    }
    // fixed order: lanes by shuffles, waves in sequence (deterministic)
#pragma unroll
    for (int w = 1; w <= L; w++) {
        double v = acc[w];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0 && w < W) {
            double tot = 0.0;
            for (int q = 0; q < T / 64; q++) tot += red[q];
            partial[(size_t)blockIdx.x * Wstride + w] = tot;
        }
    }
    if (tid == 0) partial[(size_t)blockIdx.x * Wstride] = 0.0;      // lag 0
}

// Further passes for W > 32: windows w0 .. w0 + Wn - 1 (Wn <= WT).  A task holds its COMB_B base
// entries and the COMB_B + WT - 1 partner entries that start w0 comb steps later.
template <int WT>
__global__ __launch_bounds__(MSD_THREADS) void msd_comb_hi_kernel(const double *__restrict__ DT, int64_t Fp, int F,
                                                                  const int32_t *__restrict__ perm,
                                                                  const MsdGroup *__restrict__ groups, int d, int w0,
                                                                  int Wn, int Wstride, double *__restrict__ partial, Dcom dc)
{
    constexpr int NP = COMB_B + WT - 1;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *u = reinterpret_cast<double *>(lds_raw);  // [F]
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    const int tid = threadIdx.x;
    const MsdGroup gr = groups[blockIdx.x];
    const int nq = (F + d - 1) / d;
    const int ntask = d * ((nq + COMB_B - 1) / COMB_B);
    double acc[WT];
#pragma unroll
    for (int w = 0; w < WT; w++) acc[w] = 0.0;
    for (int c = 0; c < 3 * gr.count; c++) {
        const int64_t atom = perm[gr.start + c / 3];
        const double *__restrict__ col = DT + (size_t)(3 * atom + c % 3) * Fp;
        __syncthreads();
        auto load_raw = [&]() {
            for (int k = 2 * tid; k < F; k += 2 * MSD_THREADS) {
                const double2 v2 = *reinterpret_cast<const double2 *>(col + k);
                u[k] = v2.x;
                if (k + 1 < F) u[k + 1] = v2.y;
            }
            __syncthreads();
        };
        load_raw();
        scan_column<MSD_THREADS>(dc, c % 3, u, F, red, [&]() { __syncthreads(); load_raw(); });
        for (int t = tid; t < ntask; t += MSD_THREADS) {
            const int r = t % d, J0 = (t / d) * COMB_B;
            const int limp = (F - r + d - 1) / d - J0 - w0;   // valid partner entries of this task
            if (limp <= 0) continue;
            const double m0 = t == 0 ? 0.0 : 1.0;              // origin k = 0 is skipped
            double vb[COMB_B], vp[NP];
            const double *ub = u + r + (size_t)d * J0;
#pragma unroll
            for (int i = 0; i < COMB_B; i++) vb[i] = i < limp + w0 ? ub[(size_t)d * i] : 0.0;
#pragma unroll
            for (int e = 0; e < NP; e++) vp[e] = e < limp ? ub[(size_t)d * (w0 + e)] : 0.0;
#pragma unroll
            for (int e = 0; e < NP; e++) {
                if (e < limp) {
#pragma unroll
                    for (int i = (e - WT + 1 > 0 ? e - WT + 1 : 0); i <= (e < COMB_B - 1 ? e : COMB_B - 1); i++) {
                        const double dd = vp[e] - vb[i];
                        acc[e - i] = fma(i == 0 ? dd * m0 : dd, dd, acc[e - i]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < WT; w++) {
        if (w < Wn) {
            const double tot = block_sum(acc[w], red);
            if (tid == 0) partial[(size_t)blockIdx.x * Wstride + w0 + w] = tot;
        }
    }
}

// The same comb arithmetic for trajectories too long for LDS (F + W > 19 200): the columns are prefix-summed in
// global memory first (scan_column_kernel); a workgroup then walks each column in batches of R residue classes,
// staging only their comb entries -- T[j][rr] = U[rb + rr + d j], nq x R doubles, R contiguous doubles per global
// segment -- and forms the pairs from registers as above.  One launch per 32 windows.
template <int WT>
__global__ __launch_bounds__(MSD_THREADS) void msd_comb_global_kernel(const double *__restrict__ UT, int64_t Fp, int F,
                                                                      const int32_t *__restrict__ perm,
                                                                      const MsdGroup *__restrict__ groups, int d, int R,
                                                                      int w0, int Wn, int Wstride,
                                                                      double *__restrict__ partial)
{
    constexpr int NP = COMB_B + WT - 1;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *T = reinterpret_cast<double *>(lds_raw);  // [nq][R]
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    const int tid = threadIdx.x;
    const MsdGroup gr = groups[blockIdx.x];
    const int nq = (F + d - 1) / d;
    const int nblk = (nq + COMB_B - 1) / COMB_B;
    double acc[WT];
#pragma unroll
    for (int w = 0; w < WT; w++) acc[w] = 0.0;
    for (int c = 0; c < 3 * gr.count; c++) {
        const int64_t atom = perm[gr.start + c / 3];
        const double *__restrict__ col = UT + (size_t)(3 * atom + c % 3) * Fp;
        for (int rb = 0; rb < d; rb += R) {
            const int Rn = min(R, d - rb);
            __syncthreads();
            for (int idx = tid; idx < nq * R; idx += MSD_THREADS) {
                const int j = idx / R, rr = idx - j * R;
                const int k = rb + rr + d * j;
                T[idx] = (rr < Rn && k < F) ? col[k] : 0.0;
            }
            __syncthreads();
            const int ntask = Rn * nblk;
            for (int t = tid; t < ntask; t += MSD_THREADS) {
                const int jb = t / Rn, rr = t - jb * Rn;
                const int J0 = jb * COMB_B, r = rb + rr;
                const int limp = (F - r + d - 1) / d - J0 - w0;   // valid partner entries of this task
                if (limp <= 0) continue;
                const double m0 = (r == 0 && J0 == 0) ? 0.0 : 1.0;   // origin k = 0 is skipped
                double vb[COMB_B], vp[NP];
                const double *tb = T + (size_t)J0 * R + rr;
#pragma unroll
                for (int i = 0; i < COMB_B; i++) vb[i] = i < limp + w0 ? tb[(size_t)R * i] : 0.0;
#pragma unroll
                for (int e = 0; e < NP; e++) vp[e] = e < limp ? tb[(size_t)R * (w0 + e)] : 0.0;
#pragma unroll
                for (int e = 0; e < NP; e++) {
                    if (e < limp) {
#pragma unroll
                        for (int i = (e - WT + 1 > 0 ? e - WT + 1 : 0); i <= (e < COMB_B - 1 ? e : COMB_B - 1); i++) {
                            const double dd = vp[e] - vb[i];
                            acc[e - i] = fma(i == 0 ? dd * m0 : dd, dd, acc[e - i]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < WT; w++) {
        if (w < Wn) {
            const double tot = block_sum(acc[w], red);
            if (tid == 0) partial[(size_t)blockIdx.x * Wstride + w0 + w] = tot;
        }
    }
}

// sumsq[s][w] = sum over the groups of species s, in a fixed order (deterministic):
// one workgroup per (s, w); groups are species-sorted, so species s owns [g0, g1)
__global__ __launch_bounds__(MSD_THREADS) void msd_reduce_kernel(const double *__restrict__ partial,
                                                                 const int32_t *__restrict__ sp_group_first, int W,
                                                                 double *__restrict__ sumsq)
{
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    const int s = blockIdx.x / W, w = blockIdx.x % W;
    const int g0 = sp_group_first[s], g1 = sp_group_first[s + 1];
    double acc = 0.0;
    for (int g = g0 + threadIdx.x; g < g1; g += MSD_THREADS) acc += partial[(size_t)g * W + w];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) sumsq[blockIdx.x] = acc;
}

// ---- unwrap path (amof/msd.py:222-230) in atom-major layout ----
// OUT[col][k] = x0[col] + sum_{j<=k} IN[col][j]   (x0 == nullptr: plain prefix sum; IN may alias OUT).
// Any F: the column is scanned in LDS segments with a running carry.
constexpr int SCAN_SEG = 8192;
__global__ __launch_bounds__(MSD_THREADS) void scan_column_kernel(const double *IN, const double *__restrict__ x0,
                                                                  int64_t Fp, int F, double *OUT)
{
    __shared__ double u[SCAN_SEG];
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    __shared__ double carry_s;
    const size_t col = blockIdx.x;
    double carry = x0 ? x0[col] : 0.0;
    for (int base = 0; base < F; base += SCAN_SEG) {
        const int n = min(SCAN_SEG, F - base);
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += MSD_THREADS) u[k] = IN[col * Fp + base + k];
        __syncthreads();
        lds_scan(u, n, red, carry);
        for (int k = threadIdx.x; k < n; k += MSD_THREADS) OUT[col * Fp + base + k] = u[k];
        if (threadIdx.x == 0) carry_s = u[n - 1];
        __syncthreads();
        carry = carry_s;
    }
}

// Long trajectories (F + W beyond the LDS-resident limit): window sums straight from the
// prefix-summed columns in global memory.  Workgroup (g, c) owns the windows w = c, c + C, ...
// of group g, so every partial[g][w] has exactly one writer (deterministic).
__global__ __launch_bounds__(MSD_THREADS) void msd_group_kernel_global(const double *__restrict__ UT, int64_t Fp, int F,
                                                                       const int32_t *__restrict__ perm,
                                                                       const MsdGroup *__restrict__ groups,
                                                                       const int32_t *__restrict__ windows, int W,
                                                                       double *__restrict__ partial)
{
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    const MsdGroup gr = groups[blockIdx.x];
    for (int w = blockIdx.y; w < W; w += gridDim.y) {
        const int m = windows[w];
        double tot = 0.0;
        for (int c = 0; c < 3 * gr.count; c++) {
            const int64_t atom = perm[gr.start + c / 3];
            const double *__restrict__ u = UT + (size_t)(3 * atom + c % 3) * Fp;
            double a = 0.0;
            for (int k = 1 + threadIdx.x; k + m < F; k += MSD_THREADS) {
                const double d = u[k + m] - u[k];
                a = fma(d, d, a);
            }
            tot += block_sum(a, red);
        }
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * W + w] = tot;
    }
}

// centre of mass from the atom-major layout, in two deterministic stages: partial sums over
// blocks of COMT_BLK atoms (thread = frame k, coalesced along k), then the blocks in order
constexpr int COMT_BLK = 128;
__global__ __launch_bounds__(MSD_THREADS) void com_T_partial_kernel(const double *__restrict__ UT,
                                                                    const double *__restrict__ masses, int64_t N,
                                                                    int64_t Fp, int F, double *__restrict__ part)
{
    const int k = blockIdx.x * MSD_THREADS + threadIdx.x;
    const int c = blockIdx.y;
    const int64_t i0 = (int64_t)blockIdx.z * COMT_BLK, i1 = min(i0 + COMT_BLK, N);
    if (k >= F) return;
    double s = 0.0;
    for (int64_t i = i0; i < i1; i++) s += masses[i] * UT[(size_t)(3 * i + c) * Fp + k];
    part[((size_t)blockIdx.z * 3 + c) * Fp + k] = s;
}

__global__ __launch_bounds__(MSD_THREADS) void com_T_final_kernel(const double *__restrict__ part, int nblk,
                                                                  int64_t Fp, int F, double total_mass,
                                                                  double *__restrict__ com)
{
    const int k = blockIdx.x * MSD_THREADS + threadIdx.x;
    const int c = blockIdx.y;
    if (k >= F) return;
    double s = 0.0;
    for (int b = 0; b < nblk; b++) s += part[((size_t)b * 3 + c) * Fp + k];
    com[3 * k + c] = s / total_mass;
}

__global__ __launch_bounds__(MSD_THREADS) void delta_T_kernel(const double *__restrict__ UT,
                                                              const double *__restrict__ com,
                                                              const double *__restrict__ geom, int n_cells,
                                                              int64_t N, int64_t Fp, int F, int64_t a_begin,
                                                              double *__restrict__ DT)
{
    const int k = blockIdx.y * MSD_THREADS + threadIdx.x;
    const size_t a = (size_t)a_begin + blockIdx.x;
    if (k >= F) return;
    double dx = 0.0, dy = 0.0, dz = 0.0;
    if (k >= 1) {
        double x1 = UT[(3 * a) * Fp + k], y1 = UT[(3 * a + 1) * Fp + k], z1 = UT[(3 * a + 2) * Fp + k];
        double x0 = UT[(3 * a) * Fp + k - 1], y0 = UT[(3 * a + 1) * Fp + k - 1], z0 = UT[(3 * a + 2) * Fp + k - 1];
        if (com) {
            x1 -= com[3 * k]; y1 -= com[3 * k + 1]; z1 -= com[3 * k + 2];
            x0 -= com[3 * (k - 1)]; y0 -= com[3 * (k - 1) + 1]; z0 -= com[3 * (k - 1) + 2];
        }
        const double *g = geom + (size_t)(n_cells == 1 ? 0 : k - 1) * MSD_GEOM;
        wrap_delta(g, x1 - x0, y1 - y0, z1 - z0, dx, dy, dz);
    }
    DT[(3 * a) * Fp + k] = dx;
    DT[(3 * a + 1) * Fp + k] = dy;
    DT[(3 * a + 2) * Fp + k] = dz;
}

// ---- DirectMsd (deprecated in the reference, orthogonal cells only; amof/msd.py:83-107) ----
// One thread per coordinate column walks the frames: r_t = r_{t-1} + wrap(pos_t - (r_{t-1} % a)),
// with Python's float modulo (result in [0, a)) and the reference's +-a/2 fold.
__global__ __launch_bounds__(256) void direct_walk_kernel(const double *__restrict__ pos,
                                                          const double *__restrict__ cell, int n_cells, int64_t N,
                                                          int F, double *__restrict__ sq)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= 3 * N) return;
    const int j = (int)(c % 3);
    const double r0 = pos[c];
    double r = r0;
    sq[c] = 0.0;
    for (int t = 1; t < F; t++) {
        const double a = cell[(size_t)(n_cells == 1 ? 0 : t) * 9 + 4 * j];
        double m = fmod(r, a);
        if (m != 0.0 && ((a < 0.0) != (m < 0.0))) m += a;      // numpy's % on floats
        double dr = pos[(size_t)t * 3 * N + c] - m;
        if (dr > a / 2) dr -= a;
        else if (dr < -a / 2) dr += a;
        r = dr + r;
        const double d = r - r0;
        sq[(size_t)t * 3 * N + c] = d * d;
    }
}

// msd[t][0] = sum over all atoms / N ; msd[t][1+s] = sum over species s / N_s  (fixed order)
__global__ __launch_bounds__(MSD_THREADS) void direct_reduce_kernel(const double *__restrict__ sq,
                                                                    const int32_t *__restrict__ perm,
                                                                    const int64_t *__restrict__ sp_first, int S,
                                                                    int64_t N, double *__restrict__ msd)
{
    __shared__ double red[2 * (MSD_THREADS / 64)];     // (wave totals | event flags of the scan)
    const int t = blockIdx.x;
    const double *__restrict__ row = sq + (size_t)t * 3 * N;
    double total = 0.0;
    for (int s = 0; s < S; s++) {
        const int64_t k0 = sp_first[s], k1 = sp_first[s + 1];
        double acc = 0.0;
        for (int64_t k = k0 + threadIdx.x; k < k1; k += MSD_THREADS) {
            const int64_t a = perm[k];
            acc += row[3 * a] + row[3 * a + 1] + row[3 * a + 2];
        }
        acc = block_sum(acc, red);
        total += acc;
        if (threadIdx.x == 0) msd[(size_t)t * (S + 1) + 1 + s] = k1 > k0 ? acc / (double)(k1 - k0) : 0.0;
    }
    if (threadIdx.x == 0) msd[(size_t)t * (S + 1)] = N > 0 ? total / (double)N : 0.0;
}

}  // namespace amof

using namespace amof;

__global__ void add_f64_kernel(double *dst, const double *src, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// sumsq (host, overwritten) or sumsq_dev (device, accumulated into) receives the [S][W] sums; com_ext: optional
// precomputed centre of mass of every frame (device [F][3]; frame-sharded ranks compute their rows with
// amof_msd_com_dev and all-reduce them), not with unwrap (the unwrapped centre of mass is a different quantity)
static int msd_window_run(amof_ctx *ctx, const amof_traj *t, const int32_t *windows, int32_t W, int32_t unwrap,
                          int32_t remove_com, int64_t atom_begin, int64_t atom_end, const double *com_ext, double *sumsq,
                          double *sumsq_dev)
{
    AMOF_TRY(validate_traj(ctx, t, remove_com != 0));
    const int S = t->n_species;
    const int64_t N = t->n_atoms, F = t->n_frames;
    if (W < 0 || (W > 0 && !windows) || (!sumsq && !sumsq_dev)) return fail(ctx, AMOF_EINVAL, "NULL argument");
    if (atom_begin < 0 || atom_end > N || atom_begin > atom_end) return fail(ctx, AMOF_EINVAL, "bad atom range");
    if (com_ext && unwrap) return fail(ctx, AMOF_EINVAL, "a precomputed centre of mass cannot be combined with unwrap");
    for (int w = 0; w < W; w++)
        if (windows[w] < 0 || (F > 0 && windows[w] >= F)) return fail(ctx, AMOF_EINVAL, "window %d out of range", windows[w]);
    if (sumsq)
        for (int k = 0; k < S * W; k++) sumsq[k] = 0.0;
    if (F == 0 || N == 0 || W == 0 || atom_begin == atom_end) return AMOF_OK;
    if (F > 0x7fffffffLL) return fail(ctx, AMOF_EINVAL, "too many frames");
    const size_t lds_need = ((size_t)F + (size_t)W) * sizeof(double);
    const bool lds_resident = lds_need <= 150 * 1024;   // whole time series + window sums fit in LDS

    // geometry records with the FULL inverse (wrap_positions semantics)
    HostGeom hg;
    AMOF_TRY(build_geometry(ctx, t, hg));
    std::vector<double> grec((size_t)t->n_cells * MSD_GEOM, 0.0);
    for (int64_t k = 0; k < t->n_cells; k++) {
        for (int q = 0; q < 9; q++) grec[(size_t)k * MSD_GEOM + q] = t->cell[9 * k + q];
        for (int q = 0; q < 9; q++) grec[(size_t)k * MSD_GEOM + 9 + q] = hg.invfull[(size_t)k * 9 + q];
        for (int q = 0; q < 3; q++) grec[(size_t)k * MSD_GEOM + 18 + q] = t->pbc[q] ? 1.0 : 0.0;
    }
    // species-sorted groups of the selected atoms
    std::vector<int32_t> perm;
    std::vector<MsdGroup> groups;
    std::vector<int32_t> sp_group_first(S + 1, 0);
    for (int s = 0; s < S; s++) {
        sp_group_first[s] = (int32_t)groups.size();
        size_t first = perm.size();
        for (int64_t i = atom_begin; i < atom_end; i++)
            if (t->species[i] == s) perm.push_back((int32_t)i);
        for (size_t off = first; off < perm.size(); off += MSD_GROUP) {
            MsdGroup g;
            g.start = (int32_t)off;
            g.count = (int32_t)std::min<size_t>(MSD_GROUP, perm.size() - off);
            g.species = s;
            g._pad = 0;
            groups.push_back(g);
        }
    }
    sp_group_first[S] = (int32_t)groups.size();
    double total_mass = 0.0;
    if (remove_com)
        for (int64_t i = 0; i < N; i++) total_mass += t->masses[i];

    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    timing_begin(ctx);
    const double *pos_dev = nullptr;
    AMOF_TRY(stage_positions(ctx, t, &pos_dev));
    const int64_t Fp = (F + 31) / 32 * 32;
    const size_t dt_bytes = (size_t)3 * N * Fp * sizeof(double);
    const void *d_geom, *d_perm, *d_groups, *d_win, *d_mass = nullptr, *d_sgf;
    void *d_com = nullptr, *d_DT, *d_UT = nullptr, *d_part, *d_out;
    // the call's small tables travel in ONE copy (a copy costs ~5 us of queue time however small: six of them were a tenth
    // of the pipeline)
    UploadPack pk;
    const int i_geom = pk.add(grec.data(), grec.size() * sizeof(double));
    const int i_perm = pk.add(perm.data(), perm.size() * sizeof(int32_t));
    const int i_groups = pk.add(groups.data(), groups.size() * sizeof(MsdGroup));
    const int i_win = pk.add(windows, (size_t)W * sizeof(int32_t));
    const int i_sgf = pk.add(sp_group_first.data(), sp_group_first.size() * sizeof(int32_t));
    const int i_mass = remove_com ? pk.add(t->masses, (size_t)N * sizeof(double)) : -1;
    AMOF_TRY(upload_pack(ctx, SLOT_GEOM, pk));
    d_geom = pk.ptr<double>(i_geom); d_perm = pk.ptr<int32_t>(i_perm); d_groups = pk.ptr<MsdGroup>(i_groups);
    d_win = pk.ptr<int32_t>(i_win); d_sgf = pk.ptr<int32_t>(i_sgf);
    if (remove_com) {
        d_mass = pk.ptr<double>(i_mass);
        AMOF_TRY(ensure(ctx, SLOT_AUX2, (size_t)F * 3 * sizeof(double), &d_com));
    }
    AMOF_TRY(ensure(ctx, SLOT_AUX3, dt_bytes, &d_DT));
    AMOF_TRY(ensure(ctx, SLOT_AUX5, groups.size() * (size_t)W * sizeof(double), &d_part));
    AMOF_TRY(ensure(ctx, SLOT_OUT0, (size_t)S * W * sizeof(double), &d_out));

    // transposition tile: 32 frames x 64 atoms, 1024 threads (two consecutive frames per thread).  Measured on the
    // MI355X for the headline shape (profiles/r02/msd_transpose_shapes.txt): 0.52 ms = 4.5 TB/s of read + write; a
    // flat copy of the same bytes (torch) takes 0.46 ms.  Four or eight frames per thread: 0.57 / 1.18 ms.
    constexpr int TR_TA = 64;
    auto transpose = [&](const double *com_dev, int64_t a0, int64_t a1, double *cpart) -> hipError_t {
        constexpr int TF = 32, TA = TR_TA, TH = 1024;
        auto go = [&](auto kern) -> hipError_t {
            const size_t lds = (size_t)TF * (3 * TA + 1) * sizeof(double);
            hipError_t e = allow_max_lds((const void *)kern);
            if (e != hipSuccess) return e;
            dim3 grid((unsigned)((a1 - a0 + TA - 1) / TA), (unsigned)((F + TF - 1) / TF));
            hipLaunchKernelGGL(kern, grid, dim3(TH), lds, ctx->stream, pos_dev, com_dev, (const double *)d_geom,
                               (int)t->n_cells, N, (int)F, Fp, a0, a1, (double *)d_DT, (const double *)d_mass, cpart);
            return hipGetLastError();
        };
        return hg.all_ortho ? go(delta_transpose_kernel<TF, TA, TH, true>) : go(delta_transpose_kernel<TF, TA, TH, false>);
    };
    // windows in arithmetic progression from 0 (the only thing WindowMsd produces): comb kernels
    int comb_d = 0;
    if (W >= 2 && W <= (lds_resident ? 128 : 256) && windows[0] == 0 && windows[1] > 0 && !getenv("AMOF_MSD_NOCOMB")) {
        comb_d = windows[1];
        for (int w = 0; w < W; w++)
            if ((int64_t)windows[w] != (int64_t)w * comb_d) comb_d = 0;
    }
    // 2-pass form (see the header): diagonal cells, the whole system in one call, the series LDS-resident
    Dcom dcom = {nullptr, nullptr, nullptr, (const double *)d_geom, {0.0, 0.0, 0.0}, Fp, (int32_t)t->n_cells, 0};
    // (the streaming window kernel subtracts C in its register scan: chunks of at most STREAM_CH entries per thread)
    const bool stream = lds_resident && comb_d >= 64 && comb_d <= 256 && W <= 32 && (size_t)Fp * sizeof(double) <= 150 * 1024 &&
                        !getenv("AMOF_MSD_NOSTREAM");
    const int streamT = comb_d <= 128 ? 128 : 256;
    const bool stream_regscan = (((F + streamT - 1) / streamT) | 1) <= STREAM_CH;
    const bool fold = !unwrap && remove_com && !com_ext && hg.all_ortho && lds_resident && atom_begin == 0 && atom_end == N &&
                      !(stream && !stream_regscan) && !getenv("AMOF_MSD_NOFOLD");
    if (fold) {
        const int ntiles = (int)((N + TR_TA - 1) / TR_TA);
        void *d_cpart, *d_dcT;
        AMOF_TRY(ensure(ctx, SLOT_AUX6, (size_t)ntiles * (size_t)F * 3 * sizeof(double), &d_cpart));
        // dc [3][Fp] | C [4][Fp] (row 3 = zeros) | max |dc| [3] (bits)
        AMOF_TRY(ensure(ctx, SLOT_AUX4, (size_t)7 * Fp * sizeof(double) + 32, &d_dcT));
        double *d_CT = (double *)d_dcT + (size_t)3 * Fp;
        unsigned long long *d_dcmax = (unsigned long long *)(d_CT + (size_t)4 * Fp);
        AMOF_HIP_TRY(ctx, hipMemsetAsync(d_dcmax, 0, 3 * sizeof(unsigned long long), ctx->stream));
        AMOF_HIP_TRY(ctx, transpose((const double *)nullptr, 0, N, (double *)d_cpart));
        hipLaunchKernelGGL(com_finish_kernel, dim3((unsigned)((F + COMF_FRAMES - 1) / COMF_FRAMES)), dim3(MSD_THREADS), 0,
                           ctx->stream, (const double *)d_cpart, ntiles, (int)F, Fp, total_mass, (double *)d_dcT, d_CT, d_dcmax);
        dcom.dcT = (const double *)d_dcT;
        dcom.CT = d_CT;
        dcom.dcmax = d_dcmax;
        for (int c = 0; c < 3; c++) {
            double lmin = 1e300;
            for (int64_t k = 0; k < t->n_cells; k++) lmin = std::min(lmin, fabs(t->cell[9 * k + 4 * c]));
            dcom.lhalf[c] = t->pbc[c] ? lmin * (0.5 - 1e-6) : 1e300;       // (a non-periodic axis never wraps)
        }
    } else if (!unwrap) {
        if (remove_com && !com_ext) {
            hipLaunchKernelGGL(com_kernel, dim3((unsigned)F), dim3(MSD_THREADS), 0, ctx->stream, pos_dev,
                               (const double *)d_mass, N, total_mass, (double *)d_com);
        }
        // only the atoms of this call's range are transposed (atom-sharded ranks each do their share)
        AMOF_HIP_TRY(ctx, transpose(remove_com && com_ext ? com_ext : (const double *)d_com, atom_begin, atom_end, nullptr));
    } else {
        // the unwrapped centre of mass needs every atom: all columns are transposed and scanned
        AMOF_TRY(ensure(ctx, SLOT_AUX4, dt_bytes, &d_UT));
        AMOF_HIP_TRY(ctx, transpose((const double *)nullptr, 0, N, nullptr));
        hipLaunchKernelGGL(scan_column_kernel, dim3((unsigned)(3 * N)), dim3(MSD_THREADS), 0, ctx->stream,
                           (const double *)d_DT, pos_dev, Fp, (int)F, (double *)d_UT);
        if (remove_com) {
            const int nblk = (int)((N + COMT_BLK - 1) / COMT_BLK);
            void *d_cpart;
            AMOF_TRY(ensure(ctx, SLOT_AUX6, (size_t)nblk * 3 * Fp * sizeof(double), &d_cpart));
            hipLaunchKernelGGL(com_T_partial_kernel, dim3((unsigned)((F + MSD_THREADS - 1) / MSD_THREADS), 3, (unsigned)nblk),
                               dim3(MSD_THREADS), 0, ctx->stream, (const double *)d_UT, (const double *)d_mass, N, Fp,
                               (int)F, (double *)d_cpart);
            hipLaunchKernelGGL(com_T_final_kernel, dim3((unsigned)((F + MSD_THREADS - 1) / MSD_THREADS), 3),
                               dim3(MSD_THREADS), 0, ctx->stream, (const double *)d_cpart, nblk, Fp, (int)F, total_mass,
                               (double *)d_com);
        }
        hipLaunchKernelGGL(delta_T_kernel, dim3((unsigned)(atom_end - atom_begin), (unsigned)((F + MSD_THREADS - 1) / MSD_THREADS)),
                           dim3(MSD_THREADS), 0, ctx->stream, (const double *)d_UT, (const double *)d_com,
                           (const double *)d_geom, (int)t->n_cells, N, Fp, (int)F, atom_begin, (double *)d_DT);
    }
    AMOF_HIP_TRY(ctx, hipGetLastError());
    timing_dom_begin(ctx, "msd_global");
    if (lds_resident) {
        auto launch = [&](auto kern) -> hipError_t {
            hipError_t e = allow_max_lds((const void *)kern);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3((unsigned)groups.size()), dim3(MSD_THREADS), lds_need, ctx->stream,
                               (const double *)d_DT, Fp, (int)F, (const int32_t *)d_perm, (const MsdGroup *)d_groups,
                               (const int32_t *)d_win, (int)W, (double *)d_part, dcom);
            return hipGetLastError();
        };
        // two column buffers (the next column streams in behind the arithmetic) when they fit twice per CU
        const bool db = 2 * (size_t)Fp * sizeof(double) <= 80 * 1024 && !getenv("AMOF_MSD_NODB");
        auto launch_comb = [&](auto kern, auto kern_db) -> hipError_t {
            const size_t lds = db ? 2 * (size_t)Fp * sizeof(double) : (size_t)F * sizeof(double);
            hipError_t e = db ? allow_max_lds((const void *)kern_db) : allow_max_lds((const void *)kern);
            if (e != hipSuccess) return e;
            if (db)
                hipLaunchKernelGGL(kern_db, dim3((unsigned)groups.size()), dim3(MSD_THREADS), lds, ctx->stream,
                                   (const double *)d_DT, Fp, (int)F, (const int32_t *)d_perm, (const MsdGroup *)d_groups,
                                   comb_d, (int)std::min(W, 32), (int)W, (double *)d_part, dcom);
            else
                hipLaunchKernelGGL(kern, dim3((unsigned)groups.size()), dim3(MSD_THREADS), lds, ctx->stream,
                                   (const double *)d_DT, Fp, (int)F, (const int32_t *)d_perm, (const MsdGroup *)d_groups,
                                   comb_d, (int)std::min(W, 32), (int)W, (double *)d_part, dcom);
            return hipGetLastError();
        };
        // windows 32 .. W-1 in further passes of up to 32 (each re-reads the columns)
        auto launch_comb_hi = [&](auto kern, int w0, int wn) -> hipError_t {
            const size_t lds = (size_t)F * sizeof(double);
            hipError_t e = allow_max_lds((const void *)kern);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3((unsigned)groups.size()), dim3(MSD_THREADS), lds, ctx->stream,
                               (const double *)d_DT, Fp, (int)F, (const int32_t *)d_perm, (const MsdGroup *)d_groups,
                               comb_d, w0, wn, (int)W, (double *)d_part, dcom);
            return hipGetLastError();
        };
        ctx->last_path = comb_d > 0 ? "msd_comb" : "msd_group";
        hipError_t e;
        // streaming comb kernel: one thread per residue class (window spacing 64 .. 256 frames, <= 32 windows)
        auto launch_stream = [&](auto kern) -> hipError_t {
            hipError_t e2 = allow_max_lds((const void *)kern);
            if (e2 != hipSuccess) return e2;
            hipLaunchKernelGGL(kern, dim3((unsigned)groups.size()), dim3(streamT), (size_t)Fp * sizeof(double), ctx->stream,
                               (const double *)d_DT, Fp, (int)F, (const int32_t *)d_perm, (const MsdGroup *)d_groups,
                               comb_d, (int)W, (int)W, (double *)d_part, dcom);
            ctx->last_path = "msd_stream";
            return hipGetLastError();
        };
#define AMOF_STREAM(L)                                                                                                     \
    (fold ? (streamT == 128 ? launch_stream(msd_stream_kernel<L, 128, true>) : launch_stream(msd_stream_kernel<L, 256, true>))  \
          : (streamT == 128 ? launch_stream(msd_stream_kernel<L, 128, false>) : launch_stream(msd_stream_kernel<L, 256, false>)))
        if (stream && W <= 8) e = AMOF_STREAM(7);
        else if (stream && W <= 16) e = AMOF_STREAM(15);
        else if (stream && W <= 24) e = AMOF_STREAM(23);
        else if (stream && W <= 25) e = AMOF_STREAM(24);
        else if (stream) e = AMOF_STREAM(31);
#undef AMOF_STREAM
        else if (comb_d > 0 && W <= 4) e = launch_comb(msd_comb_kernel<4>, msd_comb_db_kernel<4>);
        else if (comb_d > 0 && W <= 8) e = launch_comb(msd_comb_kernel<8>, msd_comb_db_kernel<8>);
        else if (comb_d > 0 && W <= 12) e = launch_comb(msd_comb_kernel<12>, msd_comb_db_kernel<12>);
        else if (comb_d > 0 && W <= 16) e = launch_comb(msd_comb_kernel<16>, msd_comb_db_kernel<16>);
        else if (comb_d > 0 && W <= 20) e = launch_comb(msd_comb_kernel<20>, msd_comb_db_kernel<20>);
        else if (comb_d > 0 && W <= 24) e = launch_comb(msd_comb_kernel<24>, msd_comb_db_kernel<24>);
        else if (comb_d > 0 && W <= 28) e = launch_comb(msd_comb_kernel<28>, msd_comb_db_kernel<28>);
        else if (comb_d > 0) {
            e = launch_comb(msd_comb_kernel<32>, msd_comb_db_kernel<32>);
            for (int w0 = 32; w0 < W && e == hipSuccess; w0 += 32) {
                const int wn = std::min(32, (int)W - w0);
                if (wn <= 8) e = launch_comb_hi(msd_comb_hi_kernel<8>, w0, wn);
                else if (wn <= 16) e = launch_comb_hi(msd_comb_hi_kernel<16>, w0, wn);
                else e = launch_comb_hi(msd_comb_hi_kernel<32>, w0, wn);
            }
        }
        else if (W <= 8) e = launch(msd_group_kernel<8>);
        else if (W <= 32) e = launch(msd_group_kernel<32>);
        else e = launch(msd_group_kernel<0>);
        AMOF_HIP_TRY(ctx, e);
    } else {
        // long trajectory: prefix-sum every column in place, then reduce the windows from global memory
        hipLaunchKernelGGL(scan_column_kernel, dim3((unsigned)(3 * N)), dim3(MSD_THREADS), 0, ctx->stream,
                           (const double *)d_DT, (const double *)nullptr, Fp, (int)F, (double *)d_DT);
        const int64_t nq = comb_d > 0 ? (F + comb_d - 1) / comb_d : 0;
        const int Rres = comb_d > 0 && nq > 0 ? (int)std::min<int64_t>(std::min<int64_t>(comb_d, 32), 16384 / nq) : 0;
        if (Rres >= 1) {
            // comb kernel on the scanned columns, 32 windows per launch
            ctx->last_path = "msd_comb_global";
            const size_t lds = (size_t)nq * Rres * sizeof(double);
            AMOF_HIP_TRY(ctx, allow_max_lds((const void *)msd_comb_global_kernel<32>));
            for (int w0 = 0; w0 < W; w0 += 32) {
                hipLaunchKernelGGL(msd_comb_global_kernel<32>, dim3((unsigned)groups.size()), dim3(MSD_THREADS), lds,
                                   ctx->stream, (const double *)d_DT, Fp, (int)F, (const int32_t *)d_perm,
                                   (const MsdGroup *)d_groups, comb_d, Rres, w0, std::min(32, (int)W - w0), (int)W,
                                   (double *)d_part);
            }
            AMOF_HIP_TRY(ctx, hipGetLastError());
        } else {
        const unsigned wchunks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(W, (4096 + (int64_t)groups.size() - 1) /
                                                                                    (int64_t)groups.size()));
        hipLaunchKernelGGL(msd_group_kernel_global, dim3((unsigned)groups.size(), std::min(wchunks, 65535u)),
                           dim3(MSD_THREADS), 0, ctx->stream, (const double *)d_DT, Fp, (int)F, (const int32_t *)d_perm,
                           (const MsdGroup *)d_groups, (const int32_t *)d_win, (int)W, (double *)d_part);
        AMOF_HIP_TRY(ctx, hipGetLastError());
        }
    }
    timing_dom_end(ctx, 1);
    hipLaunchKernelGGL(msd_reduce_kernel, dim3((unsigned)(S * W)), dim3(MSD_THREADS), 0, ctx->stream,
                       (const double *)d_part, (const int32_t *)d_sgf, (int)W, (double *)d_out);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    if (sumsq_dev) {
        hipLaunchKernelGGL(add_f64_kernel, dim3((unsigned)((S * W + 255) / 256)), dim3(256), 0, ctx->stream, sumsq_dev,
                           (const double *)d_out, S * (int)W);
        AMOF_HIP_TRY(ctx, hipGetLastError());
    }
    timing_end(ctx);
    if (sumsq)
        AMOF_HIP_TRY(ctx, hipMemcpyAsync(sumsq, d_out, (size_t)S * W * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}

extern "C" int amof_msd_window(amof_ctx *ctx, const amof_traj *t, const int32_t *windows, int32_t W,
                               int32_t unwrap, int32_t remove_com, int64_t atom_begin, int64_t atom_end,
                               double *sumsq)
{
    if (!ctx) return AMOF_EINVAL;
    if (!sumsq) return fail(ctx, AMOF_EINVAL, "NULL argument");
    return msd_window_run(ctx, t, windows, W, unwrap, remove_com, atom_begin, atom_end, nullptr, sumsq, nullptr);
}

extern "C" int amof_msd_window_dev(amof_ctx *ctx, const amof_traj *t, const int32_t *windows, int32_t W,
                                   int32_t unwrap, int32_t remove_com, int64_t atom_begin, int64_t atom_end,
                                   const double *com_dev, double *sumsq_dev)
{
    if (!ctx) return AMOF_EINVAL;
    if (!sumsq_dev) return fail(ctx, AMOF_EINVAL, "NULL argument");
    return msd_window_run(ctx, t, windows, W, unwrap, remove_com, atom_begin, atom_end, com_dev, nullptr, sumsq_dev);
}

extern "C" int amof_msd_com_dev(amof_ctx *ctx, const amof_traj *t, int64_t frame_begin, int64_t frame_end, double *com_dev)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_TRY(validate_traj(ctx, t, true));
    const int64_t N = t->n_atoms, F = t->n_frames;
    if (!com_dev) return fail(ctx, AMOF_EINVAL, "NULL argument");
    if (frame_begin < 0 || frame_end > F || frame_begin > frame_end) return fail(ctx, AMOF_EINVAL, "bad frame range");
    if (frame_begin == frame_end || N == 0) return AMOF_OK;
    double total_mass = 0.0;
    for (int64_t i = 0; i < N; i++) total_mass += t->masses[i];
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    timing_begin(ctx);
    const double *range_dev = nullptr;      // first frame OF THE RANGE on the device
    if (t->pos_on_device) {
        range_dev = t->pos + (size_t)frame_begin * (size_t)N * 3;
    } else {        // host input: only the frames of the range travel
        amof_traj sub = *t;
        sub.pos = t->pos + (size_t)frame_begin * (size_t)N * 3;
        sub.n_frames = frame_end - frame_begin;
        AMOF_TRY(stage_positions(ctx, &sub, &range_dev));
    }
    void *d_mass;
    AMOF_TRY(upload(ctx, SLOT_AUX1, t->masses, (size_t)N * sizeof(double), &d_mass));
    timing_dom_begin(ctx, "msd_com");
    hipLaunchKernelGGL(com_kernel, dim3((unsigned)(frame_end - frame_begin)), dim3(MSD_THREADS), 0, ctx->stream,
                       range_dev, (const double *)d_mass, N, total_mass, com_dev + 3 * frame_begin);
    timing_dom_end(ctx, 1);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    timing_end(ctx);
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}

extern "C" int amof_msd_direct(amof_ctx *ctx, const amof_traj *t, double *msd)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_TRY(validate_traj(ctx, t, false));
    if (!msd) return fail(ctx, AMOF_EINVAL, "msd is NULL");
    const int S = t->n_species;
    const int64_t N = t->n_atoms, F = t->n_frames;
    if (F == 0) return AMOF_OK;
    if (F > 0x7fffffffLL) return fail(ctx, AMOF_EINVAL, "too many frames");
    for (int64_t k = 0; k < t->n_cells; k++)
        for (int j = 0; j < 3; j++)
            if (!(t->cell[9 * k + 4 * j] > 0.0)) return fail(ctx, AMOF_EINVAL, "DirectMsd needs positive cell diagonals");
    HostTiles tiles;
    build_tiles(t, 256, tiles);
    std::vector<int64_t> sp_first(S + 1, 0);
    for (int x = 0; x < S; x++) sp_first[x + 1] = sp_first[x] + tiles.nsp[x];
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    timing_begin(ctx);
    const double *pos_dev = nullptr;
    AMOF_TRY(stage_positions(ctx, t, &pos_dev));
    void *d_cell, *d_perm, *d_spf, *d_sq, *d_out;
    AMOF_TRY(upload(ctx, SLOT_GEOM, t->cell, (size_t)t->n_cells * 9 * sizeof(double), &d_cell));
    AMOF_TRY(upload(ctx, SLOT_PERM, tiles.perm.data(), tiles.perm.size() * sizeof(int32_t), &d_perm));
    AMOF_TRY(upload(ctx, SLOT_AUX0, sp_first.data(), sp_first.size() * sizeof(int64_t), &d_spf));
    AMOF_TRY(ensure(ctx, SLOT_AUX3, (size_t)F * 3 * N * sizeof(double), &d_sq));
    AMOF_TRY(ensure(ctx, SLOT_OUT0, (size_t)F * (S + 1) * sizeof(double), &d_out));
    if (N > 0) {
        timing_dom_begin(ctx, "msd_direct");
        hipLaunchKernelGGL(direct_walk_kernel, dim3((unsigned)((3 * N + 255) / 256)), dim3(256), 0, ctx->stream, pos_dev,
                           (const double *)d_cell, (int)t->n_cells, N, (int)F, (double *)d_sq);
        timing_dom_end(ctx, 1);
        AMOF_HIP_TRY(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(direct_reduce_kernel, dim3((unsigned)F), dim3(MSD_THREADS), 0, ctx->stream, (const double *)d_sq,
                       (const int32_t *)d_perm, (const int64_t *)d_spf, S, N, (double *)d_out);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    timing_end(ctx);
    AMOF_HIP_TRY(ctx, hipMemcpyAsync(msd, d_out, (size_t)F * (S + 1) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}
