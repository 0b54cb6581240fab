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
#include <type_traits>
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

// The same sum with the four row totals combined by DPP row broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into
// rows 2 and 3) instead of readlanes: no scalar round trip (a v_readlane of a value a vector instruction has just written
// waits, and three double sums per frame are the inner loop of msd_seg_kernel).  Valid in lane 63; fixed order.
__device__ __forceinline__ double wave_sum_bcast(double x)
{
#define AMOF_DPP_STEP2(CTRL, ROWMASK)                                                                                  \
    {                                                                                                                   \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWMASK, 0xf, false);                     \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWMASK, 0xf, false);                     \
        x += __hiloint2double(hi, lo);                                                                                   \
    }
    AMOF_DPP_STEP2(0x111, 0xf)    // row_shr:1
    AMOF_DPP_STEP2(0x112, 0xf)    // row_shr:2
    AMOF_DPP_STEP2(0x114, 0xf)    // row_shr:4
    AMOF_DPP_STEP2(0x118, 0xf)    // row_shr:8   (lane 15 of every row: the row's total)
    AMOF_DPP_STEP2(0x142, 0xa)    // row_bcast:15 -> rows 1 and 3 add the total of the row before
    AMOF_DPP_STEP2(0x143, 0xc)    // row_bcast:31 -> rows 2 and 3 add lane 31 (rows 0 + 1)
#undef AMOF_DPP_STEP2
    return x;
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

// ---- fused form (round 5): no transposed copy of the trajectory --------------------------------------------------
// The transposed forms above write D_T (1.18 GB at the headline size) and read it back; their window kernel holds a whole
// column in LDS, loads / scans / multiplies it one phase after the other, and a workgroup's columns one after the other
// (profiles/r04/msd_experiments.txt: 24 % of the HBM peak; with the atoms in blocks, 8 x slower per atom --
// profiles/r05/msd_blocks_experiment.txt).  For windows m_w = w d the pairs (k, k + w d) only couple frames of the same
// residue r = k mod d, so a column is cut into SEGMENTS of d frames, e = 0 .. nq - 1 (nq = ceil(F / d)), and at "step" r
// every segment contributes one comb entry  x_e(r) = U[e d + r]  (U = the running position):
//   pass 1  msd_seg_kernel    one thread per (atom, segment): the segment's sum of wrapped raw differences RS[e][col] and,
//                             per (64-atom tile, frame), the mass-weighted coordinate sums (cpart) -- pos read ONCE, 12 MB written
//           com_tiles / com_steps (centre of mass, its steps dc, max |dc|), seg_scan (RS -> exclusive prefix P over e)
//   pass 2  msd_fused_kernel  a workgroup = 24 columns (8 atoms) x all segments; a thread owns 5 consecutive segments of one
//                             column and walks r = 0 .. d-1: one coalesced load per segment and step (pos read a SECOND time,
//                             192 contiguous bytes per frame row and workgroup), x_e(r) = P_e + sum of raw differences
//                             - c_k, published through LDS; every pair (e, e - w), w = 1 .. L, is formed from registers
//                             (own entries) or ONE LDS read per partner entry (28 reads for 120 products)
//           msd_colreduce     per-column sums -> sumsq[S][W] in species order (fixed order: deterministic)
// HBM traffic 2 x pos + ~30 MB (3.64 GB before), no LDS-resident column (any F), all loads coalesced along the atoms.
// The centre of mass enters as in the 2-pass form: wrap(raw - dc) = raw - dc unless |raw| comes within max|dc| of half the
// cell; pass 2 checks every raw difference and raises a flag, and the call is then answered by the transposed forms.
// Shape of pass 2's workgroups (256 threads, <= 128 VGPRs, 10 kB of LDS): what ONE retiring workgroup of the RDF tile
// kernel frees on a CU, so that they are placed beside a running RDF launch (second lane: amof_amd/_lazy.py).
constexpr int FU_EB = 5;          // comb entries (segments) per thread
#ifndef AMOF_FU_CPW
#define AMOF_FU_CPW 24
#endif
// columns (= 8 atoms) per workgroup; whole 128-byte lines measured slower: 32 (5-wave workgroups) 0.64, 16 (2.5-wave
// workgroups) 0.57 against 0.47 ms (-DAMOF_FU_CPW=..: experiments)
constexpr int FU_CPW = AMOF_FU_CPW;
constexpr int FU_MAX_TPC = 21;    // threads per column: nq <= 105 (workgroups of <= 512 threads: 168 registers per lane)

// TPL atoms per lane (atoms a, a + 64, ...: a "tile" of cpart is 64 TPL atoms): their m p are added in the lane before the
// wave sum -- the three double sums over the 64 lanes are ~60 of the ~130 instructions of a frame at TPL = 1.
// CONSTCELL: one cell for all frames -- its nine numbers sit in registers.  (Loaded per frame they sat behind a
// `s_waitcnt vmcnt(0)` in the frame's body: global loads retire in order, so every frame waited for ALL the positions
// prefetched for later frames -- the kernel ran at 3.3 TB/s whatever its shape: profiles/r05/msd_fused_experiments.txt.)
template <int TPL, int U, bool CONSTCELL>
__global__ __launch_bounds__(256) void msd_seg_kernel(const double *__restrict__ pos, const double *__restrict__ geom, int n_cells,
                                                      int64_t N, int F, int d, int nq, int64_t a_begin, int64_t a_end,
                                                      const double *__restrict__ masses, double *__restrict__ RS,
                                                      int64_t rs_stride, double *__restrict__ cpart)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = blockIdx.y * 4 + wv;
    if (e >= nq) return;                     // (a whole wave; the kernel has no barrier)
    const int k0 = e * d, k1 = min(k0 + d, F);
    const size_t N3 = (size_t)N * 3;
    bool ok[TPL];
    double m[TPL], px[TPL], py[TPL], pz[TPL], rx[TPL], ry[TPL], rz[TPL];
    const double *__restrict__ p[TPL];
#pragma unroll
    for (int t = 0; t < TPL; t++) {
        const int64_t a = a_begin + ((int64_t)blockIdx.x * TPL + t) * 64 + lane;
        ok[t] = a < a_end;
        const int64_t aa = ok[t] ? a : a_begin;     // (idle lanes read an atom that exists and keep nothing)
        m[t] = ok[t] ? masses[a] : 0.0;
        p[t] = pos + (size_t)k0 * N3 + (size_t)aa * 3;
        px[t] = py[t] = pz[t] = rx[t] = ry[t] = rz[t] = 0.0;
        if (k0 >= 1) {
            px[t] = p[t][-(ptrdiff_t)N3]; py[t] = p[t][1 - (ptrdiff_t)N3]; pz[t] = p[t][2 - (ptrdiff_t)N3];
        }
    }
    double gc[MSD_GEOM];        // (the entries wrap_delta_t<true> reads: diagonal of the cell and of its inverse, periodic flags)
#pragma unroll
    for (int q = 0; q < MSD_GEOM; q++) gc[q] = 0.0;
    if (CONSTCELL) {
#pragma unroll
        for (int c = 0; c < 3; c++) { gc[4 * c] = geom[4 * c]; gc[9 + 4 * c] = geom[9 + 4 * c]; gc[18 + c] = geom[18 + c]; }
    }
    // U frames per round, the next round's loads issued before this round's arithmetic
    double qx[U][TPL], qy[U][TPL], qz[U][TPL], nx[U][TPL], ny[U][TPL], nz[U][TPL];
#pragma unroll
    for (int j = 0; j < U; j++) {
        const size_t off = (size_t)(min(k0 + j, k1 - 1) - k0) * N3;
#pragma unroll
        for (int t = 0; t < TPL; t++) { qx[j][t] = p[t][off]; qy[j][t] = p[t][off + 1]; qz[j][t] = p[t][off + 2]; }
    }
    for (int kb = k0; kb < k1; kb += U) {
#pragma unroll
        for (int j = 0; j < U; j++) {
            const size_t off = (size_t)(min(kb + U + j, k1 - 1) - k0) * N3;
#pragma unroll
            for (int t = 0; t < TPL; t++) { nx[j][t] = p[t][off]; ny[j][t] = p[t][off + 1]; nz[j][t] = p[t][off + 2]; }
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int k = kb + j;
            if (k < k1) {
                double mx = 0.0, my = 0.0, mz = 0.0;
#pragma unroll
                for (int t = 0; t < TPL; t++) {
                    if (k >= 1) {
                        double dx, dy, dz;
                        if (CONSTCELL) {
                            wrap_delta_t<true>(gc, qx[j][t] - px[t], qy[j][t] - py[t], qz[j][t] - pz[t], dx, dy, dz);
                        } else {
                            const double *g = geom + (size_t)(k - 1) * MSD_GEOM;
                            wrap_delta_t<true>(g, qx[j][t] - px[t], qy[j][t] - py[t], qz[j][t] - pz[t], dx, dy, dz);
                        }
                        rx[t] += dx; ry[t] += dy; rz[t] += dz;
                    }
                    mx += m[t] * qx[j][t]; my += m[t] * qy[j][t]; mz += m[t] * qz[j][t];
                    px[t] = qx[j][t]; py[t] = qy[j][t]; pz[t] = qz[j][t];
                }
                const double sx = wave_sum_bcast(mx), sy = wave_sum_bcast(my), sz = wave_sum_bcast(mz);
                if (lane == 63) {
                    double *o = cpart + ((size_t)blockIdx.x * (size_t)F + (size_t)k) * 3;
                    o[0] = sx; o[1] = sy; o[2] = sz;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
#pragma unroll
            for (int t = 0; t < TPL; t++) { qx[j][t] = nx[j][t]; qy[j][t] = ny[j][t]; qz[j][t] = nz[j][t]; }
        }
    }
#pragma unroll
    for (int t = 0; t < TPL; t++) {
        if (ok[t]) {
            const int64_t a = a_begin + ((int64_t)blockIdx.x * TPL + t) * 64 + lane;
            double *o = RS + (size_t)e * rs_stride + 3 * (a - a_begin);
            o[0] = rx[t]; o[1] = ry[t]; o[2] = rz[t];
        }
    }
}

// csum[k][c] = sum over this call's atom tiles of cpart[t][k][c], in tile order (four thread groups share the tiles, their
// partial sums are added in group order: deterministic)
__global__ __launch_bounds__(MSD_THREADS) void com_tiles_kernel(const double *__restrict__ cpart, int ntiles, int F,
                                                                double *__restrict__ csum)
{
    __shared__ double part[COMF_GROUPS][COMF_FRAMES][3];
    const int k = blockIdx.x * COMF_FRAMES + threadIdx.x % COMF_FRAMES, g = threadIdx.x / COMF_FRAMES;
    const int per = (ntiles + COMF_GROUPS - 1) / COMF_GROUPS, t0 = min(g * per, ntiles), t1 = min(t0 + per, ntiles);
    double sx = 0.0, sy = 0.0, sz = 0.0;
    if (k < F) {
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
    part[g][threadIdx.x % COMF_FRAMES][0] = sx; part[g][threadIdx.x % COMF_FRAMES][1] = sy; part[g][threadIdx.x % COMF_FRAMES][2] = sz;
    __syncthreads();
    for (int q = threadIdx.x; q < COMF_FRAMES * 3; q += MSD_THREADS) {
        const int f = q / 3, c = q % 3, kk = blockIdx.x * COMF_FRAMES + f;
        double sum = 0.0;
        for (int gg = 0; gg < COMF_GROUPS; gg++) sum += part[gg][f][c];
        if (kk < F) csum[(size_t)kk * 3 + c] = sum;
    }
}

// c_k = csum_k / M (csum complete over ALL atoms: after the ranks' all-reduce in an atom-sharded run);
// dcT[c][k] = c_k - c_k-1 (0 at k = 0), CT[c][k] = c_k, CT[3][k] = 0, dcmax[c] = max_k |dc| (bits)
__global__ __launch_bounds__(MSD_THREADS) void com_steps_kernel(const double *__restrict__ csum, int F, int64_t Fp, double total_mass,
                                                                double *__restrict__ dcT, double *__restrict__ CT,
                                                                unsigned long long *__restrict__ dcmax)
{
    const int k = blockIdx.x * MSD_THREADS + threadIdx.x;
    double mx[3] = {0.0, 0.0, 0.0};
    if (k < F) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double ck = csum[(size_t)k * 3 + c] / total_mass;
            const double dd = k == 0 ? 0.0 : ck - csum[(size_t)(k - 1) * 3 + c] / total_mass;
            dcT[(size_t)c * Fp + k] = dd;
            CT[(size_t)c * Fp + k] = ck;
            mx[c] = fabs(dd);
        }
        CT[(size_t)3 * Fp + k] = 0.0;
    }
    // one atomic per wave and coordinate (15 000 atomics on three words took 13 us)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double v = mx[c];
        for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&dcmax[c], (unsigned long long)__double_as_longlong(v));      // (non-negative doubles order as integers)
    }
}

// RS[e][col] -> exclusive prefix over e (the running raw position at the last frame before segment e)
__global__ __launch_bounds__(256) void seg_scan_kernel(double *__restrict__ RS, int nq, int64_t ncols, int64_t stride)
{
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= ncols) return;
    double run = 0.0;
    constexpr int B = 10;                  // loads of a batch in flight together (a chain of nq dependent round trips otherwise)
    for (int e0 = 0; e0 < nq; e0 += B) {
        double v[B];
#pragma unroll
        for (int j = 0; j < B; j++) v[j] = e0 + j < nq ? RS[(size_t)(e0 + j) * stride + col] : 0.0;
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (e0 + j < nq) RS[(size_t)(e0 + j) * stride + col] = run;
            run += v[j];
        }
    }
}

// Pair products of one step.  Own entries x[0 .. EB) are the comb entries e0 + i; their partners e0 - L .. e0 + EB - 2 (index
// j = 0 .. L + EB - 2) come from LDS (j < L: rows of the step buffer, which starts with L GHOST rows of zeros for the entries
// "before" the first segment) or are own entries (j >= L).
// fused_products_fast: no predicate at all -- what a pair with a ghost row adds, x_hi^2, is taken back at the end of the
// kernel (the caller keeps  s[i] = sum over the fast steps of x[i]^2).
// fused_products_exact: every exception by predicate -- jmin: first partner that exists; jskip: the partner that is the
// time origin (entry 0 at step 0: the reference never evaluates k = 0, amof/msd.py:200) or -1; nv: own entries that exist at
// this step (the last segment may be short, the last thread's entries may lie beyond the trajectory).
template <int L>
__device__ __forceinline__ void fused_products_fast(const double (&x)[FU_EB], const double *__restrict__ xrow, double (&acc)[L + 1])
{
#pragma unroll
    for (int j = 0; j < L + FU_EB - 1; j++) {
        const double v = j >= L ? x[j >= L ? j - L : 0] : xrow[j * FU_CPW];
#pragma unroll
        for (int i = 0; i < FU_EB; i++) {
            const int w = L + i - j;                           // (compile time)
            if (w >= 1 && w <= L) {
                const double dd = x[i] - v;
                acc[w] = fma(dd, dd, acc[w]);
            }
        }
    }
}

template <int L>
__device__ __forceinline__ void fused_products_exact(const double (&x)[FU_EB], const double *__restrict__ xrow, int jmin, int jskip,
                                                  int nv, double (&acc)[L + 1])
{
#pragma unroll
    for (int j = 0; j < L + FU_EB - 1; j++) {
        if (j >= jmin && j != jskip) {
            const double v = j >= L ? x[j >= L ? j - L : 0] : xrow[j * FU_CPW];
#pragma unroll
            for (int i = 0; i < FU_EB; i++) {
                const int w = L + i - j;                           // (compile time)
                if (w >= 1 && w <= L && i < nv) {
                    const double dd = x[i] - v;
                    acc[w] = fma(dd, dd, acc[w]);
                }
            }
        }
    }
}

// GENERAL = false: every thread owns EB whole segments (nq a multiple of EB, F a multiple of d): only the predicate-free
// step exists in the kernel (the exact step's predicates cost 100 registers more per lane).  CONSTCELL: one cell for all frames.
template <int L, bool GENERAL, bool CONSTCELL, int PF>
__device__ __forceinline__ void msd_fused_body(const double *__restrict__ pos, const double *__restrict__ P, int64_t p_stride,
                                               const double *__restrict__ geom, int n_cells, const Dcom &dc, int64_t N, int F,
                                               int d, int nq, int TPC, int64_t a_begin, int64_t ncols,
                                               double *__restrict__ colpart, int64_t cp_stride, int W,
                                               int32_t *__restrict__ flag)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double *xs = reinterpret_cast<double *>(lds_raw);            // [2][L + TPC * EB][CPW]: L ghost rows of zeros, then the entries
    const int nrow = L + TPC * FU_EB;
    const int tid = threadIdx.x, cl = tid % FU_CPW, eb = tid / FU_CPW;
    // (measured and rejected: consecutive column blocks dealt to ONE XCD, so that the 128-byte lines two neighbouring blocks
    //  share -- a block's row is 192 bytes -- come from that XCD's L2 the second time: 0.476 vs 0.470 ms and the SAME
    //  FETCH_SIZE, 1.565 GB = 4/3 of the positions; 16 columns per block (whole lines, AMOF_FU_CPW=16) fetch 1.176 GB and
    //  take 0.567 ms -- 2.5 waves per workgroup: profiles/r05/msd_fused_experiments.txt)
    const int64_t lc = (int64_t)blockIdx.x * FU_CPW + cl;        // column of this call's atom range
    const bool on = eb < TPC && lc < ncols;
    const int64_t col = 3 * a_begin + (lc < ncols ? lc : 0);     // column of the frame rows: 3 a + c (idle lanes: one that exists)
    const int c = (int)(col % 3);
    const int e0 = (eb < TPC ? eb : 0) * FU_EB;                  // (idle lanes look at rows that exist)
    const size_t N3 = (size_t)N * 3;
    const double thr = dcom_thr(dc, c);
    const double *__restrict__ dcrow = dc.dcT + (size_t)c * dc.Fp;
    // a constant cell: the three numbers of this coordinate's wrap (wrap_delta_t<true>) stay in registers
    const double ginv = geom[9 + 4 * c], gper = geom[18 + c], glen = geom[4 * c];
    // entry i (segment e0 + i) exists at step r while  i d + r < rem
    const int rem = on ? max(0, min(F - e0 * d, (nq - e0) * d)) : 0;
    // steps r < r_fast need no exception in a wave whose threads own EB whole segments each: the last segment is complete
    // there (r_fast = its length); the pairs with the time origin that step 0 forms are taken back right after it
    const int r_fast = GENERAL ? F - (nq - 1) * d : d;
    bool clean = true;
    if (GENERAL) {
        const bool whole = !on || rem >= FU_EB * d || (rem > (FU_EB - 1) * d && e0 + FU_EB == nq);
        clean = __all(whole) != 0;                               // (wave-uniform)
    }
    for (int q = tid; q < 2 * L * FU_CPW; q += blockDim.x) xs[(size_t)(q / (L * FU_CPW)) * nrow * FU_CPW + q % (L * FU_CPW)] = 0.0;
    // x[i] = U_raw[k] - (c_k - c_0) at the thread's current frame of segment e0 + i (a constant per column cancels in every
    // pair; this one keeps x small): x(k) = x(k - 1) + raw_k - dc_k, started from the segment's prefix
    // (measured and rejected, profiles/r05/msd_fused_experiments.txt: positions loaded two and three steps ahead through a
    //  register ring -- 1.35 / 2.07 ms instead of 0.47: the ring's registers cost the third wave per SIMD and more)
    double x[FU_EB], prev[FU_EB], cur[PF][FU_EB], dcv[PF][FU_EB], sq[FU_EB], acc[L + 1];
    const double *pp[FU_EB], *dp[FU_EB];                          // this thread's next loads (frame rows advance by N3 per step)
#pragma unroll
    for (int w = 0; w <= L; w++) acc[w] = 0.0;
    const double c0 = dc.CT[(size_t)c * dc.Fp];
#pragma unroll
    for (int i = 0; i < FU_EB; i++) {
        const int e = e0 + i;
        x[i] = prev[i] = sq[i] = 0.0;
        const bool have = GENERAL ? i * d < rem : true;           // (!GENERAL: idle lanes read rows that exist and drop the result)
        const size_t k = (size_t)(have ? e : 0) * d;
        pp[i] = pos + k * N3 + col;
        dp[i] = dcrow + k;
        if (have) {
            x[i] = P[(size_t)e * p_stride + (lc < ncols ? lc : 0)] - (dc.CT[(size_t)c * dc.Fp + (k >= 1 ? k - 1 : 0)] - c0);
            if (k >= 1) prev[i] = *(pp[i] - N3);
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            cur[u][i] = dcv[u][i] = 0.0;
            if (have && (GENERAL ? i * d + u < rem && u < d : u < d)) {
                cur[u][i] = pp[i][(size_t)u * N3];
                dcv[u][i] = dp[i][u];
            }
        }
    }
    const int jmin = max(0, L - e0);
    bool evt = false;
    // one step: this thread's EB new comb entries (the wrapped raw difference of the frame, minus the centre-of-mass step),
    // published through LDS; then the products.  FAST: no predicate (see fused_products_fast).
    auto step = [&](int r, auto fast_tag, auto slot_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr int SL = decltype(slot_tag)::value;
        int nv = FAST ? FU_EB : 0;
#pragma unroll
        for (int i = 0; i < FU_EB; i++) {
            if (FAST || i * d + r < rem) {
                if (!FAST) nv = i + 1;
                const int k = (e0 + i) * d + r;
                double fr, per = gper, len = glen;
                if (CONSTCELL) {
                    fr = (cur[SL][i] - prev[i]) * ginv;
                } else {
                    const double *g = geom + (size_t)max(k - 1, 0) * MSD_GEOM;
                    fr = (cur[SL][i] - prev[i]) * g[9 + 4 * c];
                    per = g[18 + c];
                    len = g[4 * c];
                }
                if (per != 0.0) {
                    const double shift = 0.0 - 0.5 - 1e-7;
                    double t = fr - shift;
                    t = t - floor(t);
                    fr = t + shift;
                }
                const double raw = k >= 1 ? fr * len : 0.0;       // (frame 0 has no predecessor)
                evt |= fabs(raw) > thr;
                x[i] += raw - dcv[SL][i];
                prev[i] = cur[SL][i];
            }
        }
        // the next step's loads: in flight behind the barrier and this step's products
#pragma unroll
        for (int i = 0; i < FU_EB; i++) {
            // (GENERAL: by the entry's own length also in a fast step -- the last segment is short there, and the step
            //  r_fast - 1 would otherwise load row F, one frame behind the trajectory)
            if (!GENERAL ? r + 1 < d : (i * d + r + 1 < rem && r + 1 < d)) {
                pp[i] += N3;
                dp[i] += 1;
                cur[SL][i] = *pp[i];
                dcv[SL][i] = *dp[i];
            }
        }
        double *xb = xs + (size_t)(r & 1) * nrow * FU_CPW;
        if (eb < TPC) {
#pragma unroll
            for (int i = 0; i < FU_EB; i++) xb[(L + e0 + i) * FU_CPW + cl] = x[i];
        }
        __syncthreads();       // (one barrier per step: the other buffer is rewritten only after everybody has passed this one)
        const double *xrow = xb + (size_t)e0 * FU_CPW + cl;      // partner j = 0: entry e0 - L = row e0 behind the ghosts
        if (FAST) {
            fused_products_fast<L>(x, xrow, acc);
#pragma unroll
            for (int i = 0; i < FU_EB; i++) sq[i] = fma(x[i], x[i], sq[i]);
            if (r == 0) {
                // the reference never evaluates the time origin k = 0 (amof/msd.py:200): the pairs (entry w, entry 0) of this
                // step, w = 1 .. L, go out again (every accumulator updated unconditionally, with a selected operand: a
                // conditional update of ONE of them would turn the register array into memory)
                const double x00 = xb[L * FU_CPW + cl];
#pragma unroll
                for (int i = 0; i < FU_EB; i++) {
                    const double dd = x[i] - x00, d2 = dd * dd;
#pragma unroll
                    for (int w = 1; w <= L; w++) acc[w] -= e0 + i == w ? d2 : 0.0;
                }
            }
        } else if (nv > 0) {
            fused_products_exact<L>(x, xrow, jmin, (r == 0 && e0 <= L) ? L - e0 : -1, nv, acc);
        }
    };
    auto run = [&](int ra, int rb, auto fast_tag) {
        for (int r = ra; r < rb; r++) step(r, fast_tag, std::integral_constant<int, 0>());
    };
    if (!GENERAL) {
        run(0, d, std::true_type());
    } else {
        // steps 0 .. r_fast - 1 | r_fast .. d - 1: the first range without predicates where the whole wave allows it
        if (clean) run(0, r_fast, std::true_type());
        else run(0, r_fast, std::false_type());
        run(r_fast, d, std::false_type());
    }
    if (evt && on) atomicOr(flag, 1);
    // what the fast steps added for partners before the first segment (ghost rows: x_hi^2 at every lag w > e_hi)
#pragma unroll
    for (int w = 1; w <= L; w++) {
#pragma unroll
        for (int i = 0; i < FU_EB; i++) acc[w] -= e0 + i < w ? sq[i] : 0.0;
    }
    // per column: the threads of its segments in fixed order (eb = 0, 1, ...)
    double *red = xs;                                            // [2][TPC][CPW] (<= the step buffers)
    __syncthreads();
#pragma unroll
    for (int w = 1; w <= L; w++) {
        double *rb = red + (size_t)(w & 1) * TPC * FU_CPW;
        if (eb < TPC) rb[eb * FU_CPW + cl] = acc[w];
        __syncthreads();
        if (eb == 0 && on && w < W) {
            double tot = 0.0;
            for (int q = 0; q < TPC; q++) tot += rb[q * FU_CPW + cl];
            colpart[(size_t)w * cp_stride + lc] = tot;
        }
    }
    if (eb == 0 && on) colpart[lc] = 0.0;                        // lag 0
}

// the lean kernel at three waves per SIMD (<= 168 registers); the general one takes what it needs (its exact step is rare)
template <int L, int MAXT, bool CONSTCELL, int PF>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(3))) void msd_fused_kernel(
    const double *__restrict__ pos, const double *__restrict__ P, int64_t p_stride, const double *__restrict__ geom, int n_cells, Dcom dc,
    int64_t N, int F, int d, int nq, int TPC, int64_t a_begin, int64_t ncols, double *__restrict__ colpart, int64_t cp_stride, int W,
    int32_t *__restrict__ flag)
{
    msd_fused_body<L, false, CONSTCELL, PF>(pos, P, p_stride, geom, n_cells, dc, N, F, d, nq, TPC, a_begin, ncols, colpart, cp_stride, W, flag);
}

template <int L, int MAXT, bool CONSTCELL>
__global__ __launch_bounds__(MAXT) void msd_fused_general_kernel(
    const double *__restrict__ pos, const double *__restrict__ P, int64_t p_stride, const double *__restrict__ geom, int n_cells, Dcom dc,
    int64_t N, int F, int d, int nq, int TPC, int64_t a_begin, int64_t ncols, double *__restrict__ colpart, int64_t cp_stride, int W,
    int32_t *__restrict__ flag)
{
    msd_fused_body<L, true, CONSTCELL, 1>(pos, P, p_stride, geom, n_cells, dc, N, F, d, nq, TPC, a_begin, ncols, colpart, cp_stride, W, flag);
}

// sumsq[s][w] = sum over the atoms of species s (perm order) of their three columns: one workgroup per (s, w)
__global__ __launch_bounds__(MSD_THREADS) void msd_colreduce_kernel(const double *__restrict__ colpart, int64_t cp_stride,
                                                                    const int32_t *__restrict__ perm,
                                                                    const int32_t *__restrict__ sp_first, int64_t a_begin, int W,
                                                                    double *__restrict__ sumsq)
{
    __shared__ double red[2 * (MSD_THREADS / 64)];
    const int s = blockIdx.x / W, w = blockIdx.x % W;
    const double *__restrict__ row = colpart + (size_t)w * cp_stride;
    double acc = 0.0;
    for (int q = sp_first[s] + threadIdx.x; q < sp_first[s + 1]; q += MSD_THREADS) {
        const int64_t lc = 3 * ((int64_t)perm[q] - a_begin);
        acc += (row[lc] + row[lc + 1]) + row[lc + 2];
    }
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


// ---- host side of the fused form ----
namespace {

struct FusedDims {
    int d = 0, nq = 0, TPC = 0, L = 0;
    int64_t ncols = 0, stride = 0;      // columns of the atom range; row stride of RS / colpart (columns padded to 32)
    size_t lds = 0;
    unsigned threads = 0;
};

// does the fused form take windows m_w = w * comb_d, w = 0 .. W-1, over F frames?
bool fused_dims(int64_t F, int W, int comb_d, int64_t a0, int64_t a1, FusedDims &fd)
{
    if (comb_d < 16 || W < 2 || W > 32 || F < 64 || a1 <= a0) return false;
    fd.d = comb_d;
    fd.nq = (int)((F + comb_d - 1) / comb_d);
    fd.TPC = (fd.nq + FU_EB - 1) / FU_EB;
    if (fd.TPC > FU_MAX_TPC) return false;
    fd.L = W <= 8 ? 7 : W <= 16 ? 15 : W <= 25 ? 24 : 31;
    fd.ncols = 3 * (a1 - a0);
    fd.stride = (fd.ncols + 31) / 32 * 32;
    fd.threads = (unsigned)((fd.TPC * FU_CPW + 63) / 64 * 64);
    fd.lds = (size_t)2 * (fd.L + fd.TPC * FU_EB) * FU_CPW * sizeof(double);
    return true;
}

// pass 1 over the atoms [a0, a1): RS (SLOT_AUX7) and the tile sums, added up into csum[F][3] (device, overwritten)
int fused_pass1(amof_ctx *ctx, const amof_traj *t, const double *pos_dev, const double *d_geom, const double *d_mass, int64_t a0,
                int64_t a1, const FusedDims &fd, double *d_csum, double **d_RS_out)
{
    const int64_t N = t->n_atoms, F = t->n_frames;
    // four atoms per lane where that still gives every SIMD a wave, two down to a quarter of that, else one (measured on the
    // headline shape and its eighth, profiles/r05/msd_fused_experiments.txt: 4 x 2 0.248 / 0.092, 2 x 4 0.256 / 0.067,
    // 1 x 4 0.273 / 0.073 ms)
    const int64_t work = (a1 - a0) * (int64_t)fd.nq;
    int tpl = work >= (int64_t)4 * 64 * 1024 ? 4 : work >= (int64_t)2 * 64 * 256 ? 2 : 1, useg = 4;
    const int ntiles = (int)((a1 - a0 + 64 * tpl - 1) / (64 * tpl));
    void *d_RS, *d_cpart;
    AMOF_TRY(ensure(ctx, SLOT_AUX7, (size_t)fd.nq * (size_t)fd.stride * sizeof(double), &d_RS));
    AMOF_TRY(ensure(ctx, SLOT_AUX6, (size_t)ntiles * (size_t)F * 3 * sizeof(double), &d_cpart));
    const dim3 grid((unsigned)ntiles, (unsigned)((fd.nq + 3) / 4));
#define AMOF_SEG(T, UU)                                                                                                     \
    do {                                                                                                                    \
        if (t->n_cells == 1)                                                                                                \
            hipLaunchKernelGGL((msd_seg_kernel<T, UU, true>), grid, dim3(256), 0, ctx->stream, pos_dev, d_geom, (int)t->n_cells, N, \
                               (int)F, fd.d, fd.nq, a0, a1, d_mass, (double *)d_RS, fd.stride, (double *)d_cpart);          \
        else                                                                                                                \
            hipLaunchKernelGGL((msd_seg_kernel<T, UU, false>), grid, dim3(256), 0, ctx->stream, pos_dev, d_geom, (int)t->n_cells, N, \
                               (int)F, fd.d, fd.nq, a0, a1, d_mass, (double *)d_RS, fd.stride, (double *)d_cpart);          \
    } while (0)
    if (tpl == 4) AMOF_SEG(4, 2);
    else if (tpl == 2) AMOF_SEG(2, 4);
    else if (useg == 8) AMOF_SEG(1, 8);
    else AMOF_SEG(1, 4);
#undef AMOF_SEG
    AMOF_HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(com_tiles_kernel, dim3((unsigned)((F + COMF_FRAMES - 1) / COMF_FRAMES)), dim3(MSD_THREADS), 0, ctx->stream,
                       (const double *)d_cpart, ntiles, (int)F, d_csum);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    *d_RS_out = (double *)d_RS;
    return AMOF_OK;
}

// centre of mass from the complete csum, prefix of RS, pass 2, column sums -> d_out[S][W]; *d_flag_out: device word that is
// non-zero when a raw difference could wrap again under the centre-of-mass step (the caller reads it back and then answers
// the call with the transposed forms)
int fused_pass2(amof_ctx *ctx, const amof_traj *t, const double *pos_dev, const double *d_geom, int64_t a0, const FusedDims &fd,
                int W, const double *d_csum, double total_mass, double *d_RS, const int32_t *d_perm, const int32_t *d_spfirst,
                double *d_out, int32_t **d_flag_out)
{
    const int64_t N = t->n_atoms, F = t->n_frames;
    const int S = t->n_species;
    const int64_t Fp = (F + 31) / 32 * 32;
    void *d_dcT, *d_colpart, *d_flag;
    AMOF_TRY(ensure(ctx, SLOT_AUX4, (size_t)7 * Fp * sizeof(double) + 64, &d_dcT));
    AMOF_TRY(ensure(ctx, SLOT_AUX8, (size_t)(fd.L + 1) * (size_t)fd.stride * sizeof(double), &d_colpart));
    double *d_CT = (double *)d_dcT + (size_t)3 * Fp;
    unsigned long long *d_dcmax = (unsigned long long *)(d_CT + (size_t)4 * Fp);
    d_flag = d_dcmax + 3;
    // (dcmax and the flag word sit next to each other behind C: one fill)
    AMOF_HIP_TRY(ctx, hipMemsetAsync(d_dcmax, 0, 3 * sizeof(unsigned long long) + 4 * sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(com_steps_kernel, dim3((unsigned)((F + MSD_THREADS - 1) / MSD_THREADS)), dim3(MSD_THREADS), 0, ctx->stream,
                       d_csum, (int)F, Fp, total_mass, (double *)d_dcT, d_CT, d_dcmax);
    hipLaunchKernelGGL(seg_scan_kernel, dim3((unsigned)((fd.ncols + 255) / 256)), dim3(256), 0, ctx->stream, d_RS, fd.nq, fd.ncols,
                       fd.stride);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    Dcom dc = {(const double *)d_dcT, d_CT, d_dcmax, d_geom, {0.0, 0.0, 0.0}, Fp, (int32_t)t->n_cells, 0};
    for (int c = 0; c < 3; c++) {
        double lmin = 1e300;
        for (int64_t k = 0; k < t->n_cells; k++) lmin = std::min(lmin, fabs(t->cell[9 * k + 4 * c]));
        dc.lhalf[c] = t->pbc[c] ? lmin * (0.5 - 1e-6) : 1e300;       // (a non-periodic axis never wraps)
    }
    timing_dom_begin(ctx, "msd_fused");
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = allow_max_lds((const void *)kern);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)((fd.ncols + FU_CPW - 1) / FU_CPW)), dim3(fd.threads), fd.lds, ctx->stream, pos_dev,
                           (const double *)d_RS, fd.stride, d_geom, (int)t->n_cells, dc, N, (int)F, fd.d, fd.nq, fd.TPC, a0, fd.ncols,
                           (double *)d_colpart, fd.stride, W, (int32_t *)d_flag);
        return hipGetLastError();
    };
    // lean kernel: every thread owns EB whole segments
    const bool general = fd.nq % FU_EB != 0 || F % fd.d != 0 || getenv("AMOF_MSD_FUSED_GENERAL");
    const bool constcell = t->n_cells == 1;
    hipError_t e;
#define AMOF_FUSED_L(LL)                                                                                                        \
    (general ? (constcell ? go(msd_fused_general_kernel<LL, 512, true>) : go(msd_fused_general_kernel<LL, 512, false>))         \
             : (constcell ? go(msd_fused_kernel<LL, 512, true, 1>) : go(msd_fused_kernel<LL, 512, false, 1>)))
    e = fd.L == 7 ? AMOF_FUSED_L(7) : fd.L == 15 ? AMOF_FUSED_L(15) : fd.L == 24 ? AMOF_FUSED_L(24) : AMOF_FUSED_L(31);
#undef AMOF_FUSED_L
    AMOF_HIP_TRY(ctx, e);
    timing_dom_end(ctx, 1);
    hipLaunchKernelGGL(msd_colreduce_kernel, dim3((unsigned)(S * W)), dim3(MSD_THREADS), 0, ctx->stream, (const double *)d_colpart,
                       fd.stride, d_perm, d_spfirst, a0, W, d_out);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    *d_flag_out = (int32_t *)d_flag;
    return AMOF_OK;
}

}  // namespace

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
    std::vector<int32_t> sp_first_atom(S + 1, 0);       // species s owns perm[sp_first_atom[s] .. sp_first_atom[s + 1])
    for (int s = 0; s < S; s++) sp_first_atom[s] = sp_group_first[s] < (int32_t)groups.size() ? groups[sp_group_first[s]].start : (int32_t)perm.size();
    sp_first_atom[S] = (int32_t)perm.size();
    double total_mass = 0.0;
    if (remove_com)
        for (int64_t i = 0; i < N; i++) total_mass += t->masses[i];

    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    timing_begin(ctx);
    const double *pos_dev = nullptr;
    AMOF_TRY(stage_positions(ctx, t, &pos_dev));
    const int64_t Fp = (F + 31) / 32 * 32;
    const size_t dt_bytes = (size_t)3 * N * Fp * sizeof(double);
    const void *d_geom, *d_perm, *d_groups, *d_win, *d_mass = nullptr, *d_sgf, *d_sfa;
    void *d_com = nullptr, *d_DT = nullptr, *d_UT = nullptr, *d_part = nullptr, *d_out;
    // the call's small tables travel in ONE copy (a copy costs ~5 us of queue time however small: six of them were a tenth
    // of the pipeline)
    UploadPack pk;
    const int i_geom = pk.add(grec.data(), grec.size() * sizeof(double));
    const int i_perm = pk.add(perm.data(), perm.size() * sizeof(int32_t));
    const int i_groups = pk.add(groups.data(), groups.size() * sizeof(MsdGroup));
    const int i_win = pk.add(windows, (size_t)W * sizeof(int32_t));
    const int i_sgf = pk.add(sp_group_first.data(), sp_group_first.size() * sizeof(int32_t));
    const int i_sfa = pk.add(sp_first_atom.data(), sp_first_atom.size() * sizeof(int32_t));
    const int i_mass = remove_com ? pk.add(t->masses, (size_t)N * sizeof(double)) : -1;
    AMOF_TRY(upload_pack(ctx, SLOT_GEOM, pk));
    d_geom = pk.ptr<double>(i_geom); d_perm = pk.ptr<int32_t>(i_perm); d_groups = pk.ptr<MsdGroup>(i_groups);
    d_win = pk.ptr<int32_t>(i_win); d_sgf = pk.ptr<int32_t>(i_sgf); d_sfa = pk.ptr<int32_t>(i_sfa);
    if (remove_com) {
        d_mass = pk.ptr<double>(i_mass);
        AMOF_TRY(ensure(ctx, SLOT_AUX2, (size_t)F * 3 * sizeof(double), &d_com));
    }
    AMOF_TRY(ensure(ctx, SLOT_OUT0, (size_t)S * W * sizeof(double), &d_out));
    // windows in arithmetic progression from 0 (the only thing WindowMsd produces): comb kernels
    int comb_d = 0;
    if (W >= 2 && W <= (lds_resident ? 128 : 256) && windows[0] == 0 && windows[1] > 0 && !getenv("AMOF_MSD_NOCOMB")) {
        comb_d = windows[1];
        for (int w = 0; w < W; w++)
            if ((int64_t)windows[w] != (int64_t)w * comb_d) comb_d = 0;
    }
    // fused form (no transposed copy: see msd_seg_kernel): diagonal cells, the whole system in one call (the centre of
    // mass needs every atom; atom-sharded ranks go through amof_msd_shard_begin / _finish), centre of mass removed, no unwrap
    FusedDims fd;
    bool done = false;
    if (!unwrap && remove_com && !com_ext && hg.all_ortho && atom_begin == 0 && atom_end == N && !getenv("AMOF_MSD_NOFUSED") &&
        fused_dims(F, W, comb_d, atom_begin, atom_end, fd)) {
        double *d_RS = nullptr;
        int32_t *d_flag = nullptr, evt = 0;
        AMOF_TRY(fused_pass1(ctx, t, pos_dev, (const double *)d_geom, (const double *)d_mass, atom_begin, atom_end, fd,
                             (double *)d_com, &d_RS));
        AMOF_TRY(fused_pass2(ctx, t, pos_dev, (const double *)d_geom, atom_begin, fd, (int)W, (const double *)d_com, total_mass, d_RS,
                             (const int32_t *)d_perm, (const int32_t *)d_sfa, (double *)d_out, &d_flag));
        AMOF_TRY(fetch(ctx, &evt, d_flag, sizeof evt));
        AMOF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // (not sync_stream: the staged tables stay where they are)
        done = evt == 0;       // else: an entry could wrap again under the centre-of-mass step -- the transposed forms answer
    }
    if (!done) {
    AMOF_TRY(ensure(ctx, SLOT_AUX3, dt_bytes, &d_DT));
    AMOF_TRY(ensure(ctx, SLOT_AUX5, groups.size() * (size_t)W * sizeof(double), &d_part));

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
        AMOF_HIP_TRY(ctx, hipGetLastError());
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
    }   // (!done: the transposed forms)
    if (sumsq_dev) {
        hipLaunchKernelGGL(add_f64_kernel, dim3((unsigned)((S * W + 255) / 256)), dim3(256), 0, ctx->stream, sumsq_dev,
                           (const double *)d_out, S * (int)W);
        AMOF_HIP_TRY(ctx, hipGetLastError());
    }
    timing_end(ctx);
    if (sumsq)
        AMOF_TRY(fetch(ctx, sumsq, d_out, (size_t)S * W * sizeof(double)));
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

// ---- atom-sharded window MSD around ONE all-reduce: the fused form under atom sharding --------------------------------
namespace {

__global__ void com_from_csum_kernel(const double *__restrict__ csum, int n, double total_mass, double *__restrict__ com)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) com[i] = csum[i] / total_mass;
}

struct ShardTables {
    std::vector<double> grec;
    std::vector<int32_t> perm, sp_first;
    const double *d_geom = nullptr, *d_mass = nullptr;
    const int32_t *d_perm = nullptr, *d_sp_first = nullptr;
    double total_mass = 0.0;
    int comb_d = 0;
    bool ortho = true;
};

// what both halves of a sharded call need on the device: geometry records, the species-sorted atoms of the range, masses
int shard_tables(amof_ctx *ctx, const amof_traj *t, const int32_t *windows, int32_t W, int64_t a0, int64_t a1, ShardTables &st)
{
    const int S = t->n_species;
    const int64_t N = t->n_atoms;
    HostGeom hg;
    AMOF_TRY(build_geometry(ctx, t, hg));
    st.ortho = hg.all_ortho;
    st.grec.assign((size_t)t->n_cells * MSD_GEOM, 0.0);
    for (int64_t k = 0; k < t->n_cells; k++) {
        for (int q = 0; q < 9; q++) st.grec[(size_t)k * MSD_GEOM + q] = t->cell[9 * k + q];
        for (int q = 0; q < 9; q++) st.grec[(size_t)k * MSD_GEOM + 9 + q] = hg.invfull[(size_t)k * 9 + q];
        for (int q = 0; q < 3; q++) st.grec[(size_t)k * MSD_GEOM + 18 + q] = t->pbc[q] ? 1.0 : 0.0;
    }
    st.sp_first.assign(S + 1, 0);
    for (int s = 0; s < S; s++) {
        st.sp_first[s] = (int32_t)st.perm.size();
        for (int64_t i = a0; i < a1; i++)
            if (t->species[i] == s) st.perm.push_back((int32_t)i);
    }
    st.sp_first[S] = (int32_t)st.perm.size();
    st.total_mass = 0.0;
    for (int64_t i = 0; i < N; i++) st.total_mass += t->masses[i];
    st.comb_d = 0;
    if (W >= 2 && windows[0] == 0 && windows[1] > 0) {
        st.comb_d = windows[1];
        for (int w = 0; w < W; w++)
            if ((int64_t)windows[w] != (int64_t)w * st.comb_d) st.comb_d = 0;
    }
    UploadPack pk;
    const int i_geom = pk.add(st.grec.data(), st.grec.size() * sizeof(double));
    const int i_perm = pk.add(st.perm.data(), st.perm.size() * sizeof(int32_t));
    const int i_spf = pk.add(st.sp_first.data(), st.sp_first.size() * sizeof(int32_t));
    const int i_mass = pk.add(t->masses, (size_t)N * sizeof(double));
    AMOF_TRY(upload_pack(ctx, SLOT_GEOM, pk));
    st.d_geom = pk.ptr<double>(i_geom);
    st.d_perm = pk.ptr<int32_t>(i_perm);
    st.d_sp_first = pk.ptr<int32_t>(i_spf);
    st.d_mass = pk.ptr<double>(i_mass);
    return AMOF_OK;
}

int shard_check(amof_ctx *ctx, const amof_traj *t, const int32_t *windows, int32_t W, int64_t a0, int64_t a1)
{
    AMOF_TRY(validate_traj(ctx, t, true));
    const int64_t N = t->n_atoms, F = t->n_frames;
    if (W < 0 || (W > 0 && !windows)) return fail(ctx, AMOF_EINVAL, "NULL argument");
    if (a0 < 0 || a1 > N || a0 > a1) return fail(ctx, AMOF_EINVAL, "bad atom range");
    if (!t->pos_on_device) return fail(ctx, AMOF_EUNSUPPORTED, "the sharded form reads device-resident positions");
    if (F > 0x7fffffffLL) return fail(ctx, AMOF_EINVAL, "too many frames");
    for (int w = 0; w < W; w++)
        if (windows[w] < 0 || (F > 0 && windows[w] >= F)) return fail(ctx, AMOF_EINVAL, "window %d out of range", windows[w]);
    return AMOF_OK;
}

}  // namespace

extern "C" int amof_msd_shard_begin(amof_ctx *ctx, const amof_traj *t, const int32_t *windows, int32_t W, int64_t a0, int64_t a1,
                                    double *csum_dev)
{
    if (!ctx) return AMOF_EINVAL;
    if (!csum_dev) return fail(ctx, AMOF_EINVAL, "NULL argument");
    AMOF_TRY(shard_check(ctx, t, windows, W, a0, a1));
    const int64_t F = t->n_frames;
    ctx->shard_ticket = 0;
    ShardTables st;
    FusedDims fd;
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    AMOF_TRY(shard_tables(ctx, t, windows, W, a0, a1, st));
    if (!st.ortho || getenv("AMOF_MSD_NOFUSED") || !fused_dims(F, W, st.comb_d, a0, std::max(a1, a0 + 1), fd))
        return fail(ctx, AMOF_EUNSUPPORTED, "the fused window-MSD form does not take this call (general cell, or windows that are "
                                            "not w * d with 16 <= d, W <= 32, ceil(F / d) <= %d)", FU_MAX_TPC * FU_EB);
    timing_begin(ctx);
    timing_dom_begin(ctx, "msd_fused");
    if (a1 > a0) {
        double *d_RS = nullptr;
        AMOF_TRY(fused_dims(F, W, st.comb_d, a0, a1, fd) ? AMOF_OK : AMOF_EINVAL);
        AMOF_TRY(fused_pass1(ctx, t, t->pos, st.d_geom, st.d_mass, a0, a1, fd, csum_dev, &d_RS));
    } else {
        AMOF_HIP_TRY(ctx, hipMemsetAsync(csum_dev, 0, (size_t)F * 3 * sizeof(double), ctx->stream));
    }
    timing_dom_end(ctx, 1);
    timing_end(ctx);
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    ctx->shard_ticket = ctx->calls;       // (timing_begin counts the calls: a finish must be the NEXT call on this context)
    ctx->shard_key[0] = (int64_t)(intptr_t)t->pos; ctx->shard_key[1] = F; ctx->shard_key[2] = t->n_atoms;
    ctx->shard_key[3] = a0; ctx->shard_key[4] = a1; ctx->shard_key[5] = st.comb_d; ctx->shard_key[6] = W;
    return AMOF_OK;
}

extern "C" int amof_msd_shard_finish(amof_ctx *ctx, const amof_traj *t, const int32_t *windows, int32_t W, int64_t a0, int64_t a1,
                                     const double *csum_dev, double *sumsq_dev)
{
    if (!ctx) return AMOF_EINVAL;
    if (!csum_dev || !sumsq_dev) return fail(ctx, AMOF_EINVAL, "NULL argument");
    AMOF_TRY(shard_check(ctx, t, windows, W, a0, a1));
    const int S = t->n_species;
    const int64_t F = t->n_frames;
    ShardTables st;
    FusedDims fd;
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    AMOF_TRY(shard_tables(ctx, t, windows, W, a0, a1, st));
    const int64_t key[7] = {(int64_t)(intptr_t)t->pos, F, t->n_atoms, a0, a1, st.comb_d, W};
    bool same = ctx->shard_ticket != 0 && ctx->shard_ticket == ctx->calls;
    for (int q = 0; q < 7; q++) same = same && key[q] == ctx->shard_key[q];
    ctx->shard_ticket = 0;
    if (!same) return fail(ctx, AMOF_EINVAL, "amof_msd_shard_finish must follow amof_msd_shard_begin of the same arguments on this context");
    if (a1 == a0 || W == 0) return AMOF_OK;
    if (!fused_dims(F, W, st.comb_d, a0, a1, fd)) return fail(ctx, AMOF_EINVAL, "window list changed");
    timing_begin(ctx);
    void *d_RS, *d_out;
    AMOF_TRY(ensure(ctx, SLOT_AUX7, (size_t)fd.nq * (size_t)fd.stride * sizeof(double), &d_RS));      // (what begin left there)
    AMOF_TRY(ensure(ctx, SLOT_OUT0, (size_t)S * W * sizeof(double), &d_out));
    int32_t *d_flag = nullptr, evt = 0;
    AMOF_TRY(fused_pass2(ctx, t, t->pos, st.d_geom, a0, fd, (int)W, csum_dev, st.total_mass, (double *)d_RS, st.d_perm, st.d_sp_first,
                         (double *)d_out, &d_flag));
    AMOF_TRY(fetch(ctx, &evt, d_flag, sizeof evt));
    AMOF_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (evt) {
        // an entry could wrap again under the centre-of-mass step: the transposed forms with the completed centre of mass
        void *d_com;
        AMOF_TRY(ensure(ctx, SLOT_AUX9, (size_t)F * 3 * sizeof(double), &d_com));
        hipLaunchKernelGGL(com_from_csum_kernel, dim3((unsigned)((3 * F + 255) / 256)), dim3(256), 0, ctx->stream, csum_dev, (int)(3 * F),
                           st.total_mass, (double *)d_com);
        AMOF_HIP_TRY(ctx, hipGetLastError());
        AMOF_HIP_TRY(ctx, sync_stream(ctx));
        return msd_window_run(ctx, t, windows, W, 0, 1, a0, a1, (const double *)d_com, nullptr, sumsq_dev);
    }
    hipLaunchKernelGGL(add_f64_kernel, dim3((unsigned)((S * W + 255) / 256)), dim3(256), 0, ctx->stream, sumsq_dev, (const double *)d_out,
                       S * (int)W);
    AMOF_HIP_TRY(ctx, hipGetLastError());
    timing_end(ctx);
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
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
    AMOF_TRY(fetch(ctx, msd, d_out, (size_t)F * (S + 1) * sizeof(double)));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}
