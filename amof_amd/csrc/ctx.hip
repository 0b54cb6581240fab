// Context, scratch memory, host-side geometry and tiling for libamofhip.so.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include <chrono>
#include <thread>

#include "amof_internal.h"

namespace amof {

int fail(amof_ctx *ctx, int code, const char *fmt, ...)
{
    char msg[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = msg;
    return code;
}

int ensure(amof_ctx *ctx, Slot s, size_t bytes, void **out)
{
    DevBuf &b = ctx->buf[s];
    if (bytes == 0) bytes = 16;
    if (b.cap < bytes) {
        if (b.p) {
            // the buffer may still be in use by work queued on the stream
            AMOF_HIP_TRY(ctx, sync_stream(ctx));
            AMOF_HIP_TRY(ctx, hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        size_t want = bytes + bytes / 8;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) {
            b.p = nullptr;
            return fail(ctx, AMOF_ENOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        }
        b.cap = want;
    }
    *out = b.p;
    return AMOF_OK;
}

hipError_t sync_stream(amof_ctx *ctx)
{
    hipError_t e = hipStreamSynchronize(ctx->stream);
    ctx->pin_off = 0;      // every copy queued from the staging ring has run
    return e;
}

int upload(amof_ctx *ctx, Slot s, const void *src, size_t bytes, void **out)
{
    AMOF_TRY(ensure(ctx, s, bytes, out));
    if (!bytes) return AMOF_OK;
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (ctx->pin && need <= ctx->pin_cap - ctx->pin_off) {
        // small table: through the pinned ring (the caller's buffer is free at once, the copy is truly asynchronous)
        unsigned char *stage = ctx->pin + ctx->pin_off;
        memcpy(stage, src, bytes);
        ctx->pin_off += need;
        AMOF_HIP_TRY(ctx, hipMemcpyAsync(*out, stage, bytes, hipMemcpyHostToDevice, ctx->stream));
        return AMOF_OK;
    }
    AMOF_HIP_TRY(ctx, hipMemcpyAsync(*out, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable-host copies return once the source has been consumed
    return AMOF_OK;
}

int fetch(amof_ctx *ctx, void *dst, const void *src_dev, size_t bytes)
{
    if (!bytes) return AMOF_OK;
    if (ctx->rb && bytes <= ctx->rb_cap) {
        AMOF_HIP_TRY(ctx, hipMemcpyAsync(ctx->rb, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        AMOF_HIP_TRY(ctx, sync_stream(ctx));
        memcpy(dst, ctx->rb, bytes);
        return AMOF_OK;
    }
    AMOF_HIP_TRY(ctx, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}

int upload_pack(amof_ctx *ctx, Slot s, UploadPack &pk)
{
    AMOF_TRY(ensure(ctx, s, pk.total, &pk.base));
    if (!pk.total) return AMOF_OK;
    if (ctx->pin && pk.total <= ctx->pin_cap - ctx->pin_off) {
        unsigned char *stage = ctx->pin + ctx->pin_off;
        for (const UploadPack::Piece &pc : pk.pieces)
            if (pc.bytes) memcpy(stage + pc.off, pc.src, pc.bytes);
        ctx->pin_off += pk.total;
        AMOF_HIP_TRY(ctx, hipMemcpyAsync(pk.base, stage, pk.total, hipMemcpyHostToDevice, ctx->stream));
        return AMOF_OK;
    }
    // too big for the staging ring: piece by piece (pageable-host copies return once the source has been consumed)
    for (const UploadPack::Piece &pc : pk.pieces)
        if (pc.bytes)
            AMOF_HIP_TRY(ctx, hipMemcpyAsync(static_cast<unsigned char *>(pk.base) + pc.off, pc.src, pc.bytes, hipMemcpyHostToDevice,
                                             ctx->stream));
    return AMOF_OK;
}

int stage_positions(amof_ctx *ctx, const amof_traj *t, const double **pos_dev)
{
    if (t->pos_on_device) {
        *pos_dev = t->pos;
        return AMOF_OK;
    }
    size_t bytes = (size_t)t->n_frames * (size_t)t->n_atoms * 3 * sizeof(double);
    void *p = nullptr;
    AMOF_TRY(upload(ctx, SLOT_POS, t->pos, bytes, &p));
    *pos_dev = (const double *)p;
    return AMOF_OK;
}

int stager_begin(amof_ctx *ctx, const amof_traj *t, bool allow_lazy, Stager &st)
{
    st.ctx = ctx;
    st.t = t;
    st.lazy = false;
    st.upto = t->n_frames;
    if (t->pos_on_device || !allow_lazy || t->n_frames < 1024) {
        const double *p = nullptr;
        AMOF_TRY(stage_positions(ctx, t, &p));
        st.dev = const_cast<double *>(p);
        return AMOF_OK;
    }
    // a copy queued by a call that ended early must not race a reallocation below
    AMOF_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    void *p = nullptr;
    AMOF_TRY(ensure(ctx, SLOT_POS, (size_t)t->n_frames * (size_t)t->n_atoms * 3 * sizeof(double), &p));
    st.dev = (double *)p;
    st.lazy = true;
    st.upto = 0;
    return AMOF_OK;
}

int stager_need(Stager &st, int64_t f1)
{
    if (!st.lazy || f1 <= st.upto) return AMOF_OK;
    amof_ctx *ctx = st.ctx;
    const size_t per = (size_t)st.t->n_atoms * 3;
    AMOF_HIP_TRY(ctx, hipMemcpyAsync(st.dev + (size_t)st.upto * per, st.t->pos + (size_t)st.upto * per,
                                     (size_t)(f1 - st.upto) * per * sizeof(double), hipMemcpyHostToDevice,
                                     ctx->copy_stream));
    AMOF_HIP_TRY(ctx, hipEventRecord(ctx->ev_copy, ctx->copy_stream));
    AMOF_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copy, 0));
    st.upto = f1;
    return AMOF_OK;
}

void timing_begin(amof_ctx *ctx)
{
    ctx->ev_valid = false;
    ctx->dom_launches = 0;
    ctx->calls++;
    __atomic_store_n(&ctx->progress, 2 * ctx->calls, __ATOMIC_RELEASE);
    (void)hipEventRecord(ctx->ev_all0, ctx->stream);
}
void timing_end(amof_ctx *ctx)
{
    (void)hipEventRecord(ctx->ev_all1, ctx->stream);
    ctx->ev_valid = true;
}
void timing_dom_begin(amof_ctx *ctx, const char *path)
{
    ctx->last_path = path;
    (void)hipEventRecord(ctx->ev_dom0, ctx->stream);
}
void timing_dom_end(amof_ctx *ctx, int64_t launches)
{
    (void)hipEventRecord(ctx->ev_dom1, ctx->stream);
    ctx->dom_launches = launches;
    __atomic_store_n(&ctx->progress, 2 * ctx->calls + 1, __ATOMIC_RELEASE);
}

int validate_traj(amof_ctx *ctx, const amof_traj *t, bool need_masses)
{
    if (!ctx) return AMOF_EINVAL;
    if (!t) return fail(ctx, AMOF_EINVAL, "traj is NULL");
    if (t->n_frames < 0 || t->n_atoms < 0) return fail(ctx, AMOF_EINVAL, "negative n_frames / n_atoms");
    if (t->n_atoms > 0x7fffff00LL) return fail(ctx, AMOF_EINVAL, "n_atoms too large");
    if (t->n_frames > 0x7fffff00LL) return fail(ctx, AMOF_EINVAL, "n_frames too large");
    if (t->n_species <= 0 || t->n_species > 64) return fail(ctx, AMOF_EINVAL, "n_species must be 1..64");
    if (t->n_cells != 1 && t->n_cells != t->n_frames)
        return fail(ctx, AMOF_EINVAL, "n_cells must be 1 or n_frames");
    if ((t->n_frames > 0 && t->n_atoms > 0 && !t->pos) || !t->cell || (t->n_atoms > 0 && !t->species))
        return fail(ctx, AMOF_EINVAL, "NULL pos / cell / species");
    if (need_masses && t->n_atoms > 0 && !t->masses) return fail(ctx, AMOF_EINVAL, "masses required");
    for (int64_t i = 0; i < t->n_atoms; i++)
        if (t->species[i] < 0 || t->species[i] >= t->n_species)
            return fail(ctx, AMOF_EINVAL, "species[%lld] = %d out of range", (long long)i, t->species[i]);
    return AMOF_OK;
}

// Geometry record of one cell (arithmetic order is part of the contract; the
// oracle restates the same sequence, oracle/amof_oracle.c geom_make).
static int geom_one(const double *c, const uint8_t *pbc, double *rec, double *invf)
{
    double m00 = c[4] * c[8] - c[5] * c[7];
    double m01 = c[3] * c[8] - c[5] * c[6];
    double m02 = c[3] * c[7] - c[4] * c[6];
    double det = c[0] * m00 - c[1] * m01 + c[2] * m02;
    if (!(fabs(det) > 0.0) || !isfinite(det)) return AMOF_ESINGULAR;
    invf[0] = (c[4] * c[8] - c[5] * c[7]) / det;
    invf[1] = (c[2] * c[7] - c[1] * c[8]) / det;
    invf[2] = (c[1] * c[5] - c[2] * c[4]) / det;
    invf[3] = (c[5] * c[6] - c[3] * c[8]) / det;
    invf[4] = (c[0] * c[8] - c[2] * c[6]) / det;
    invf[5] = (c[2] * c[3] - c[0] * c[5]) / det;
    invf[6] = (c[3] * c[7] - c[4] * c[6]) / det;
    invf[7] = (c[1] * c[6] - c[0] * c[7]) / det;
    invf[8] = (c[0] * c[4] - c[1] * c[3]) / det;
    for (int k = 0; k < 9; k++) rec[k] = c[k];
    for (int k = 0; k < 3; k++) {
        double cx = invf[k], cy = invf[3 + k], cz = invf[6 + k];
        rec[18 + k] = 1.0 / sqrt(cx * cx + cy * cy + cz * cz);
        for (int i = 0; i < 3; i++) rec[9 + 3 * i + k] = pbc[k] ? invf[3 * i + k] : 0.0;
    }
    rec[21] = fabs(det);
    bool ortho = c[1] == 0.0 && c[2] == 0.0 && c[3] == 0.0 && c[5] == 0.0 && c[6] == 0.0 && c[7] == 0.0;
    rec[22] = ortho ? 1.0 : 0.0;
    rec[23] = 0.0;
    return AMOF_OK;
}

int build_geometry(amof_ctx *ctx, const amof_traj *t, HostGeom &g)
{
    int64_t nc = t->n_cells;
    g.rec.assign((size_t)nc * GEOM_STRIDE, 0.0);
    g.invfull.assign((size_t)nc * 9, 0.0);
    g.all_ortho = true;
    g.volume_sum = 0.0;
    for (int64_t k = 0; k < nc; k++) {
        int rc = geom_one(t->cell + 9 * k, t->pbc, &g.rec[(size_t)k * GEOM_STRIDE], &g.invfull[(size_t)k * 9]);
        if (rc) return fail(ctx, rc, "cell of frame %lld is singular", (long long)k);
        if (g.rec[(size_t)k * GEOM_STRIDE + 22] == 0.0) g.all_ortho = false;
    }
    // asap3 adds atoms.get_volume() once per update(): sum over frames in frame order
    for (int64_t f = 0; f < t->n_frames; f++)
        g.volume_sum += g.rec[(size_t)(nc == 1 ? 0 : f) * GEOM_STRIDE + 21];
    return AMOF_OK;
}

// Lattice vectors E != 0 that can bring a base-image vector within R:
// keep n iff max_k h_k * max(0, |n_k| - 1/2) < R * (1 - 1e-9).
static int images_one(const double *rec, const uint8_t *pbc, double R, std::vector<double> &out)
{
    out.clear();
    int M[3];
    double Rm = R * (1.0 - 1e-9);
    for (int k = 0; k < 3; k++) {
        if (!pbc[k]) { M[k] = 0; continue; }
        double q = floor(Rm / rec[18 + k] + 0.5) + 1.0;
        if (q > 64) return AMOF_ECAPACITY;
        M[k] = (int)q;
    }
    for (int a = -M[0]; a <= M[0]; a++)
        for (int b = -M[1]; b <= M[1]; b++)
            for (int c = -M[2]; c <= M[2]; c++) {
                if (a == 0 && b == 0 && c == 0) continue;
                int n[3] = {a, b, c};
                double bound = 0.0;
                for (int k = 0; k < 3; k++) {
                    double tt = fabs((double)n[k]) - 0.5;
                    if (tt > 0.0 && rec[18 + k] * tt > bound) bound = rec[18 + k] * tt;
                }
                if (!(bound < Rm)) continue;
                for (int x = 0; x < 3; x++)
                    out.push_back(fma((double)n[2], rec[6 + x], fma((double)n[1], rec[3 + x], (double)n[0] * rec[x])));
                if (out.size() / 3 > AMOF_MAX_IMAGES) return AMOF_ECAPACITY;
            }
    return AMOF_OK;
}

int build_images(amof_ctx *ctx, const amof_traj *t, const HostGeom &g, double R,
                 std::vector<double> &img, std::vector<int32_t> &nimg, int &max_img)
{
    int64_t nc = t->n_cells;
    std::vector<std::vector<double>> lists((size_t)nc);
    nimg.assign((size_t)nc, 0);
    max_img = 0;
    if (R > 0.0) {
        for (int64_t k = 0; k < nc; k++) {
            int rc = images_one(&g.rec[(size_t)k * GEOM_STRIDE], t->pbc, R, lists[(size_t)k]);
            if (rc) return fail(ctx, rc, "cutoff %g needs too many periodic images of cell %lld (max %d)", R,
                                (long long)k, AMOF_MAX_IMAGES);
            nimg[(size_t)k] = (int32_t)(lists[(size_t)k].size() / 3);
            max_img = std::max(max_img, (int)nimg[(size_t)k]);
        }
    }
    img.assign((size_t)nc * (size_t)max_img * 3, 0.0);
    for (int64_t k = 0; k < nc; k++)
        std::copy(lists[(size_t)k].begin(), lists[(size_t)k].end(), img.begin() + (size_t)k * max_img * 3);
    return AMOF_OK;
}

// granule > 0 (the fast RDF tile kernel: its 128-atom centre sub-tiles): tile sizes are multiples of the granule except
// the last tile of a species, so that only one sub-tile per species is ragged (ZIF-4 3x3x4: Zn 576 = 384 + 192 instead
// of 288 + 288, i.e. sub-tiles 128 128 128 | 128 64 instead of 128 128 32 | 128 128 32)
void build_tiles(const amof_traj *t, int tile, HostTiles &out, int granule)
{
    int S = t->n_species;
    int64_t N = t->n_atoms;
    out.nsp.assign(S, 0);
    for (int64_t i = 0; i < N; i++) out.nsp[t->species[i]]++;
    std::vector<int64_t> first(S + 1, 0);
    for (int s = 0; s < S; s++) first[s + 1] = first[s] + out.nsp[s];
    out.perm.assign((size_t)N, 0);
    std::vector<int64_t> fill(first.begin(), first.end() - 1);
    for (int64_t i = 0; i < N; i++) out.perm[(size_t)fill[t->species[i]]++] = (int32_t)i;
    out.tiles.clear();
    out.sp_first_tile.assign(S, 0);
    out.sp_ntiles.assign(S, 0);
    for (int s = 0; s < S; s++) {
        int64_t n = out.nsp[s];
        int64_t nt = (n + tile - 1) / tile;
        out.sp_first_tile[s] = (int32_t)out.tiles.size();
        out.sp_ntiles[s] = (int32_t)nt;
        int64_t base = nt ? n / nt : 0, rem = nt ? n % nt : 0, off = first[s];
        const int64_t ng = granule > 0 ? (n + granule - 1) / granule : 0;      // granules of this species
        const int64_t gbase = nt ? ng / nt : 0, grem = nt ? ng % nt : 0;
        for (int64_t k = 0; k < nt; k++) {
            int64_t cnt = base + (k < rem ? 1 : 0);
            if (granule > 0) cnt = std::min<int64_t>((gbase + (k < grem ? 1 : 0)) * granule, first[s] + n - off);
            Tile tl;
            tl.start = (int32_t)off;
            tl.count = (int32_t)cnt;
            tl.species = s;
            tl._pad = 0;
            out.tiles.push_back(tl);
            off += cnt;
        }
    }
}

}  // namespace amof

using namespace amof;

extern "C" {

int amof_abi_version(void) { return AMOF_ABI_VERSION; }

int amof_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int amof_ctx_create(int device, amof_ctx **out) { return amof_ctx_create2(device, 0, out); }

int amof_ctx_create2(int device, int flags, amof_ctx **out)
{
    if (!out) return AMOF_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return AMOF_ENODEVICE;
    if (device < 0 || device >= n) return AMOF_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return AMOF_EHIP;
    amof_ctx *ctx = new (std::nothrow) amof_ctx();
    if (!ctx) return AMOF_ENOMEM;
    ctx->device = device;
    hipError_t se;
    if (flags & (AMOF_CTX_HIGH_PRIORITY | AMOF_CTX_LOW_PRIORITY)) {
        int least = 0, greatest = 0;      // (numerically lowest = highest priority)
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
        se = hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, (flags & AMOF_CTX_HIGH_PRIORITY) ? greatest : least);
    } else {
        se = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    }
    if (se != hipSuccess) {
        delete ctx;
        return AMOF_EHIP;
    }
    ctx->stream = ctx->own_stream;
    if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_copy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming) != hipSuccess) {
        amof_ctx_destroy(ctx);
        return AMOF_EHIP;
    }
    // (without the ring uploads fall back to pageable copies: slower, still correct)
    ctx->pin_cap = (size_t)4 << 20;
    if (hipHostMalloc((void **)&ctx->pin, ctx->pin_cap, hipHostMallocDefault) != hipSuccess) {
        ctx->pin = nullptr;
        ctx->pin_cap = 0;
        (void)hipGetLastError();
    }
    ctx->rb_cap = (size_t)1 << 20;
    if (hipHostMalloc((void **)&ctx->rb, ctx->rb_cap, hipHostMallocDefault) != hipSuccess) {
        ctx->rb = nullptr;
        ctx->rb_cap = 0;
        (void)hipGetLastError();
    }
    if (hipEventCreate(&ctx->ev_all0) != hipSuccess || hipEventCreate(&ctx->ev_all1) != hipSuccess ||
        hipEventCreate(&ctx->ev_dom0) != hipSuccess || hipEventCreate(&ctx->ev_dom1) != hipSuccess) {
        amof_ctx_destroy(ctx);
        return AMOF_EHIP;
    }
    *out = ctx;
    return AMOF_OK;
}

void amof_ctx_destroy(amof_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int s = 0; s < SLOT_COUNT; s++)
        if (ctx->buf[s].p) (void)hipFree(ctx->buf[s].p);
    if (ctx->ev_all0) (void)hipEventDestroy(ctx->ev_all0);
    if (ctx->ev_all1) (void)hipEventDestroy(ctx->ev_all1);
    if (ctx->ev_dom0) (void)hipEventDestroy(ctx->ev_dom0);
    if (ctx->ev_dom1) (void)hipEventDestroy(ctx->ev_dom1);
    if (ctx->ev_copy) (void)hipEventDestroy(ctx->ev_copy);
    if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->rb) (void)hipHostFree(ctx->rb);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *amof_last_error(const amof_ctx *ctx) { return ctx ? ctx->err.c_str() : "NULL context"; }

int amof_ctx_set_stream(amof_ctx *ctx, void *hip_stream)
{
    if (!ctx) return AMOF_EINVAL;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    if (next != ctx->stream) {
        // copies out of the pinned staging ring may still be queued on the old stream (an entry point that failed
        // before its final synchronisation): drain it before the ring is reused from the new one
        AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
        AMOF_HIP_TRY(ctx, sync_stream(ctx));
        ctx->stream = next;
    }
    return AMOF_OK;
}

int amof_ctx_wait_stream(amof_ctx *ctx, void *hip_stream)
{
    if (!ctx) return AMOF_EINVAL;
    if ((hipStream_t)hip_stream == ctx->stream) return AMOF_OK;   // same queue: already ordered
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    AMOF_HIP_TRY(ctx, hipEventRecord(ctx->ev_order, (hipStream_t)hip_stream));
    AMOF_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_order, 0));
    return AMOF_OK;
}

int64_t amof_ctx_calls(const amof_ctx *ctx)
{
    return ctx ? __atomic_load_n(&ctx->progress, __ATOMIC_ACQUIRE) / 2 : 0;
}

int amof_ctx_follow(amof_ctx *ctx, amof_ctx *other, int64_t min_calls, double timeout_s)
{
    if (!ctx || !other) return AMOF_EINVAL;
    if (ctx == other) return 1;
    if (ctx->device != other->device) return fail(ctx, AMOF_EINVAL, "amof_ctx_follow: the two contexts are on different devices");
    if (min_calls > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const int64_t p = __atomic_load_n(&other->progress, __ATOMIC_ACQUIRE);
            if (p / 2 > min_calls || (p / 2 == min_calls && (p & 1))) break;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= timeout_s) return 0;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    AMOF_TRY(amof_ctx_wait_stream(ctx, (void *)other->stream));
    return 1;
}

int amof_ctx_debug_poison(amof_ctx *ctx, int byte)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    AMOF_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (int s = 0; s < SLOT_COUNT; s++)
        if (ctx->buf[s].p && ctx->buf[s].cap) AMOF_HIP_TRY(ctx, hipMemsetAsync(ctx->buf[s].p, byte, ctx->buf[s].cap, ctx->stream));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}

int amof_ctx_synchronize(amof_ctx *ctx)
{
    if (!ctx) return AMOF_EINVAL;
    AMOF_HIP_TRY(ctx, hipSetDevice(ctx->device));
    AMOF_HIP_TRY(ctx, sync_stream(ctx));
    return AMOF_OK;
}

double amof_last_kernel_seconds(const amof_ctx *ctx, int which)
{
    if (!ctx || !ctx->ev_valid) return -1.0;
    float ms = 0.f;
    hipEvent_t a = which == 1 ? ctx->ev_dom0 : ctx->ev_all0;
    hipEvent_t b = which == 1 ? ctx->ev_dom1 : ctx->ev_all1;
    if (hipEventSynchronize(b) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.0;
    return (double)ms * 1e-3;
}

int64_t amof_last_kernel_launches(const amof_ctx *ctx) { return ctx ? ctx->dom_launches : 0; }
const char *amof_last_path(const amof_ctx *ctx) { return ctx ? ctx->last_path : ""; }

}  // extern "C"
