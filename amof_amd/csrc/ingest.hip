// Trajectory ingest (host code only): XYZ / extended-XYZ positions and CP2K .cell files
// straight into the packed arrays of the hot path.
//
// Replaces, for the packed path, ase.io.read(filename, index, format='xyz') as driven by
// Trajectory.from_traj / read_lammps_traj / read_cp2k_traj (amof/trajectory.py:37-60,
// 193-228) and np.genfromtxt on the CP2K cell log (amof/trajectory.py:217).  SURVEY 8f-1.
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/amof_hip.h"

namespace {

thread_local std::string g_err;

int ingest_fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped()
    {
        if (p && n) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
    int open_file(const char *path)
    {
        fd = open(path, O_RDONLY);
        if (fd < 0) return ingest_fail(AMOF_EINVAL, std::string("cannot open ") + path + ": " + strerror(errno));
        struct stat st;
        if (fstat(fd, &st) != 0) return ingest_fail(AMOF_EINVAL, "fstat failed");
        n = (size_t)st.st_size;
        if (n == 0) return AMOF_OK;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            p = nullptr;
            return ingest_fail(AMOF_ENOMEM, "mmap failed");
        }
        p = (const char *)m;
        madvise(m, n, MADV_SEQUENTIAL);
        return AMOF_OK;
    }
};

inline const char *line_end(const char *c, const char *e)
{
    const char *q = (const char *)memchr(c, '\n', (size_t)(e - c));
    return q ? q : e;
}

inline const char *skip_ws(const char *c, const char *e)
{
    while (c < e && (*c == ' ' || *c == '\t' || *c == '\r')) c++;
    return c;
}

const double kPow10[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                         1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Decimal -> double.  Exact (correctly rounded) fast path when the significand has <= 15
// digits and |exponent| <= 22 (both the significand and the power of ten are exact doubles,
// one rounding: Clinger); anything else goes through strtod.
inline bool parse_double(const char *&c, const char *e, double &out)
{
    const char *s = skip_ws(c, e);
    const char *start = s;
    if (s >= e) return false;
    bool neg = false;
    if (*s == '-' || *s == '+') { neg = *s == '-'; s++; }
    uint64_t mant = 0;
    int digits = 0, exp10 = 0;
    bool any = false;
    while (s < e && *s >= '0' && *s <= '9') {
        if (digits < 19) { mant = mant * 10 + (uint64_t)(*s - '0'); if (mant) digits++; } else exp10++;
        s++; any = true;
    }
    if (s < e && *s == '.') {
        s++;
        while (s < e && *s >= '0' && *s <= '9') {
            if (digits < 19) { mant = mant * 10 + (uint64_t)(*s - '0'); if (mant) digits++; exp10--; }
            s++; any = true;
        }
    }
    if (!any) return false;
    if (s < e && (*s == 'e' || *s == 'E' || *s == 'd' || *s == 'D')) {
        const char *t = s + 1;
        bool eneg = false;
        if (t < e && (*t == '-' || *t == '+')) { eneg = *t == '-'; t++; }
        if (t < e && *t >= '0' && *t <= '9') {
            int ev = 0;
            while (t < e && *t >= '0' && *t <= '9') { if (ev < 10000) ev = ev * 10 + (*t - '0'); t++; }
            exp10 += eneg ? -ev : ev;
            s = t;
        }
    }
    if (digits <= 15 && exp10 >= -22 && exp10 <= 22) {
        double v = (double)mant;
        v = exp10 < 0 ? v / kPow10[-exp10] : v * kPow10[exp10];
        out = neg ? -v : v;
    } else {
        std::string tmp(start, (size_t)(s - start));
        for (char &ch : tmp) if (ch == 'd' || ch == 'D') ch = 'e';
        out = strtod(tmp.c_str(), nullptr);
    }
    c = s;
    return true;
}

struct FrameIndex {
    std::vector<size_t> off;   // byte offset of each frame's atom-count line
    int64_t n_atoms = -1;
};

int index_frames(const Mapped &m, FrameIndex &ix)
{
    const char *c = m.p, *e = m.p + m.n;
    while (c < e) {
        const char *le = line_end(c, e);
        const char *s = skip_ws(c, le);
        if (s == le) { c = le + 1; continue; }    // blank line between frames
        char *endp = nullptr;
        long long n = strtoll(std::string(s, (size_t)(le - s)).c_str(), &endp, 10);
        if (n <= 0) return ingest_fail(AMOF_EINVAL, "bad atom-count line at byte " + std::to_string((size_t)(c - m.p)));
        if (ix.n_atoms < 0) ix.n_atoms = n;
        else if (n != ix.n_atoms)
            return ingest_fail(AMOF_EINVAL, "frame " + std::to_string(ix.off.size()) + " has " + std::to_string(n) +
                                                " atoms, frame 0 has " + std::to_string(ix.n_atoms));
        ix.off.push_back((size_t)(c - m.p));
        c = le < e ? le + 1 : e;                  // comment line
        c = line_end(c, e);
        c = c < e ? c + 1 : e;
        for (long long k = 0; k < n; k++) {       // atom lines
            if (c >= e) return ingest_fail(AMOF_EINVAL, "file ends inside frame " + std::to_string(ix.off.size() - 1));
            c = line_end(c, e);
            c = c < e ? c + 1 : e;
        }
    }
    return AMOF_OK;
}

int parse_frame(const Mapped &m, size_t off, int64_t N, double *pos, char *symbols, double *lattice, int *has_lat)
{
    const char *e = m.p + m.n;
    const char *c = line_end(m.p + off, e);
    c = c < e ? c + 1 : e;
    const char *ce = line_end(c, e);              // comment line [c, ce)
    if (lattice) {
        *has_lat = 0;
        const char *key = "Lattice=\"";
        const size_t kl = strlen(key);
        for (const char *q = c; q + kl < ce; q++) {
            if (memcmp(q, key, kl) == 0) {
                const char *v = q + kl;
                int k = 0;
                while (k < 9 && parse_double(v, ce, lattice[k])) k++;
                *has_lat = k == 9;
                break;
            }
        }
    }
    if (!pos) return AMOF_OK;                     // (lattice-only pass: the cells of a stream, known before its frames)
    c = ce < e ? ce + 1 : e;
    for (int64_t k = 0; k < N; k++) {
        const char *le = line_end(c, e);
        const char *s = skip_ws(c, le);
        const char *t = s;
        while (t < le && *t != ' ' && *t != '\t') t++;
        if (symbols) {
            size_t len = (size_t)(t - s);
            memset(symbols + 4 * k, 0, 4);
            memcpy(symbols + 4 * k, s, len < 3 ? len : 3);
        }
        const char *q = t;
        if (!parse_double(q, le, pos[3 * k]) || !parse_double(q, le, pos[3 * k + 1]) || !parse_double(q, le, pos[3 * k + 2]))
            return ingest_fail(AMOF_EINVAL, "bad atom line " + std::to_string(k) + " in frame at byte " + std::to_string(off));
        c = le < e ? le + 1 : e;
    }
    return AMOF_OK;
}

}  // namespace

extern "C" const char *amof_ingest_last_error(void) { return g_err.c_str(); }

extern "C" int amof_xyz_scan(const char *path, int64_t *n_frames, int64_t *n_atoms)
{
    if (!path || !n_frames || !n_atoms) return ingest_fail(AMOF_EINVAL, "NULL argument");
    Mapped m;
    int rc = m.open_file(path);
    if (rc) return rc;
    FrameIndex ix;
    rc = index_frames(m, ix);
    if (rc) return rc;
    *n_frames = (int64_t)ix.off.size();
    *n_atoms = ix.n_atoms < 0 ? 0 : ix.n_atoms;
    return AMOF_OK;
}

// frames first, first + step, ... of an indexed mapping into pos (shared by the one-shot and the handle entry points)
static int read_frames(const Mapped &m, const FrameIndex &ix, int64_t first, int64_t count, int64_t step, int64_t n_atoms,
                       double *pos, char *symbols, double *lattice, int32_t *has_lattice, int32_t n_threads)
{
    const int64_t F = (int64_t)ix.off.size(), N = ix.n_atoms;
    // the caller sized pos / symbols for n_atoms per frame (amof_xyz_scan): a file rewritten in between must not
    // make this call write past those buffers
    if (F > 0 && N != n_atoms)
        return ingest_fail(AMOF_EINVAL, "file has " + std::to_string(N) + " atoms per frame, caller expects " +
                                            std::to_string(n_atoms) + " (file changed since amof_xyz_scan?)");
    for (int64_t k = 0; k < count; k++) {
        int64_t f = first + k * step;
        if (f < 0 || f >= F) return ingest_fail(AMOF_EINVAL, "frame " + std::to_string(f) + " out of range (file has " + std::to_string(F) + ")");
    }
    if (count == 0) return AMOF_OK;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    if ((int64_t)nt > count) nt = (int)count;
    std::vector<int> rcs((size_t)nt, AMOF_OK);
    std::vector<std::string> errs((size_t)nt);
    std::vector<int> lat_all((size_t)count, 0);
    auto worker = [&](int w) {
        for (int64_t k = w; k < count; k += nt) {
            int hl = 0;
            int r = parse_frame(m, ix.off[(size_t)(first + k * step)], N, pos ? pos + (size_t)k * N * 3 : nullptr,
                                (k == 0) ? symbols : nullptr, lattice ? lattice + 9 * k : nullptr, &hl);
            lat_all[(size_t)k] = hl;
            if (r) { rcs[(size_t)w] = r; errs[(size_t)w] = g_err; return; }
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < nt; w++) th.emplace_back(worker, w);
    worker(0);
    for (auto &t : th) t.join();
    for (int w = 0; w < nt; w++)
        if (rcs[(size_t)w]) return ingest_fail(rcs[(size_t)w], errs[(size_t)w]);
    if (has_lattice) {
        int all = 1;
        for (int64_t k = 0; k < count; k++) all &= lat_all[(size_t)k];
        *has_lattice = lattice ? all : 0;
    }
    return AMOF_OK;
}

extern "C" int amof_xyz_read(const char *path, int64_t first, int64_t count, int64_t step, int64_t n_atoms,
                             double *pos, char *symbols, double *lattice, int32_t *has_lattice, int32_t n_threads)
{
    if (!path || !pos || count < 0 || step == 0 || n_atoms < 0) return ingest_fail(AMOF_EINVAL, "bad argument");
    Mapped m;
    int rc = m.open_file(path);
    if (rc) return rc;
    FrameIndex ix;
    rc = index_frames(m, ix);
    if (rc) return rc;
    return read_frames(m, ix, first, count, step, n_atoms, pos, symbols, lattice, has_lattice, n_threads);
}

// An open trajectory file: the mapping and the frame index are built once and serve any number of batch reads (a
// streamed analysis reads a 2.6 GB file in a few dozen batches; re-indexing it for every batch would cost 0.3 s each).
struct amof_xyz_file {
    Mapped m;
    FrameIndex ix;
};

extern "C" int amof_xyz_open(const char *path, amof_xyz_file **out, int64_t *n_frames, int64_t *n_atoms)
{
    if (!path || !out) return ingest_fail(AMOF_EINVAL, "NULL argument");
    *out = nullptr;
    amof_xyz_file *f = new (std::nothrow) amof_xyz_file;
    if (!f) return ingest_fail(AMOF_ENOMEM, "out of memory");
    int rc = f->m.open_file(path);
    if (!rc) rc = index_frames(f->m, f->ix);
    if (rc) {
        delete f;
        return rc;
    }
    // batches are read at random offsets, several threads each: no sequential read-ahead hint
    if (f->m.p && f->m.n) madvise((void *)f->m.p, f->m.n, MADV_NORMAL);
    if (n_frames) *n_frames = (int64_t)f->ix.off.size();
    if (n_atoms) *n_atoms = f->ix.n_atoms < 0 ? 0 : f->ix.n_atoms;
    *out = f;
    return AMOF_OK;
}

extern "C" int amof_xyz_read_frames(amof_xyz_file *f, int64_t first, int64_t count, int64_t step, int64_t n_atoms,
                                    double *pos, char *symbols, double *lattice, int32_t *has_lattice, int32_t n_threads)
{
    if (!f || (!pos && !lattice) || count < 0 || step == 0 || n_atoms < 0) return ingest_fail(AMOF_EINVAL, "bad argument");
    return read_frames(f->m, f->ix, first, count, step, n_atoms, pos, symbols, lattice, has_lattice, n_threads);
}

extern "C" void amof_xyz_close(amof_xyz_file *f)
{
    delete f;
}

// CP2K cell log: "# Step Time Ax Ay Az Bx By Bz Cx Cy Cz Volume"; the reference keeps columns
// [2:-1] of every data row (amof/trajectory.py:217-224).
extern "C" int amof_cp2k_cell_read(const char *path, int64_t max_rows, double *cell, int64_t *n_rows)
{
    if (!path || !n_rows) return ingest_fail(AMOF_EINVAL, "NULL argument");
    Mapped m;
    int rc = m.open_file(path);
    if (rc) return rc;
    const char *c = m.p, *e = m.p + m.n;
    int64_t rows = 0;
    while (c < e) {
        const char *le = line_end(c, e);
        const char *s = skip_ws(c, le);
        if (s < le && *s != '#') {
            double v[16];
            int k = 0;
            const char *q = s;
            while (k < 16 && parse_double(q, le, v[k])) k++;
            if (k < 12) return ingest_fail(AMOF_EINVAL, "row " + std::to_string(rows) + " has " + std::to_string(k) + " columns, expected >= 12");
            if (cell) {
                if (rows >= max_rows) return ingest_fail(AMOF_ECAPACITY, "more rows than max_rows");
                for (int x = 0; x < 9; x++) cell[9 * rows + x] = v[2 + x];
            }
            rows++;
        }
        c = le < e ? le + 1 : e;
    }
    *n_rows = rows;
    return AMOF_OK;
}

// ---- packing a list of frames (host only) --------------------------------------------------------------------------------
// The reference's trajectory is a Python list of ase.Atoms (amof/trajectory.py:27-35,56-59) and every analysis walks it frame
// by frame (amof/rdf.py:88-93, amof/msd.py:218-242).  amof_pack_frames copies the frames' position arrays into the packed
// [F][N][3] array on n_threads threads (numpy releases the GIL per copy but pays ~25 us of interpreter per frame: 150 ms for
// 5000 x 9792); amof_frames_checksum fingerprints every frame's bytes (four interleaved multiply-rotate lanes over the 64-bit
// words: memory speed), so that a list that was packed before is recognised as unchanged without copying it again.
namespace {

inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

// Eight independent sums  a_j += (w ^ salt) * K_j  (the multiply is off the dependency chain: one multiply per word, the adds
// chain), salt = a counter of the word's position times an odd constant (swapped words change the sums).  One changed word
// changes its sum by (difference of two 64-bit values) * odd constant != 0 mod 2^64: any single-word edit is seen.
uint64_t frame_checksum(const double *p, int64_t n_words)
{
    const uint64_t *w = reinterpret_cast<const uint64_t *>(p);
    static const uint64_t K[8] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull,
                                  0xff51afd7ed558ccdull, 0xc4ceb9fe1a85ec53ull, 0xbf58476d1ce4e5b9ull, 0x94d049bb133111ebull};
    uint64_t a[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    uint64_t salt = 0x2545f4914f6cdd1dull;
    int64_t i = 0;
    for (; i + 8 <= n_words; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] += (w[i + j] ^ salt) * K[j];
        salt += 0x9e3779b97f4a7c15ull;
    }
    for (int j = 0; i < n_words; i++, j++) a[j] += (w[i] ^ salt) * K[j];
    uint64_t h = (uint64_t)n_words;
    for (int j = 0; j < 8; j++) h = rotl64(h ^ a[j], 29) * 0x9e3779b97f4a7c15ull;
    h ^= h >> 29;
    h *= 0xbf58476d1ce4e5b9ull;
    h ^= h >> 32;
    return h;
}

template <typename Fn> void over_frames(int64_t n_frames, int32_t n_threads, Fn fn)
{
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n_frames));
    if (nt == 1) {
        fn(0, n_frames);
        return;
    }
    std::vector<std::thread> th;
    const int64_t per = (n_frames + nt - 1) / nt;
    for (int q = 0; q < nt; q++) {
        const int64_t a = q * per, b = std::min(n_frames, a + per);
        if (a < b) th.emplace_back([=]() { fn(a, b); });
    }
    for (auto &x : th) x.join();
}

}  // namespace

extern "C" int amof_pack_frames(const double *const *frame_pos, int64_t n_frames, int64_t n_atoms, double *dst, uint64_t *checksums,
                                int32_t n_threads)
{
    if (n_frames < 0 || n_atoms < 0 || (n_frames > 0 && (!frame_pos || !dst))) return AMOF_EINVAL;
    for (int64_t k = 0; k < n_frames; k++)
        if (!frame_pos[k] && n_atoms > 0) return AMOF_EINVAL;
    const size_t words = (size_t)n_atoms * 3;
    over_frames(n_frames, n_threads, [=](int64_t a, int64_t b) {
        for (int64_t k = a; k < b; k++) {
            memcpy(dst + (size_t)k * words, frame_pos[k], words * sizeof(double));
            if (checksums) checksums[k] = frame_checksum(dst + (size_t)k * words, (int64_t)words);     // (the copy is in cache)
        }
    });
    return AMOF_OK;
}

extern "C" int amof_frames_checksum(const double *const *frame_pos, int64_t n_frames, int64_t n_atoms, uint64_t *checksums,
                                    int32_t n_threads)
{
    if (n_frames < 0 || n_atoms < 0 || (n_frames > 0 && (!frame_pos || !checksums))) return AMOF_EINVAL;
    for (int64_t k = 0; k < n_frames; k++)
        if (!frame_pos[k] && n_atoms > 0) return AMOF_EINVAL;
    const int64_t words = n_atoms * 3;
    over_frames(n_frames, n_threads, [=](int64_t a, int64_t b) {
        for (int64_t k = a; k < b; k++) checksums[k] = frame_checksum(frame_pos[k], words);
    });
    return AMOF_OK;
}
