// lpx_feeder.hip -- N1: frames from binary PCD files into pinned host memory, and a double-buffered
// prefetch -> H2D -> launch chain -> D2H pipeline over them.
//
// Counterpart of the reference's input harness: Dataloader::preload_point_clouds (src/dataloader.cpp:128-153)
// loads every data/*.pcd with pcl::io::loadPCDFile into a PointXYZI cloud, re-packs it into a PointCloud2
// (:87-126) and the processor decodes that again (src/conversions.cpp:62-85) before it calls segment().  Here the
// file payload -- float32 records exactly as stored, header fields as in data/0000000000.pcd:1-11 -- is read
// straight into hipHostMalloc memory and handed to the device with its own record layout (the *_fields entry
// points): no intermediate copy, no decode on the host.
//
// Pipeline of lpx_feeder_run (B = frame slots of the batch context, two device buffer sets):
//   chain k    : H2D of its frames' records on a copy stream into set k % 2   (after chain k-2 has computed)
//                launch chain on the context stream                            (after that H2D, and after
//                                                                               chain k-2's results left set k % 2)
//   chain k - 1: the host waits for it WHILE chain k runs, reads its 4-word counts and issues exact-size D2H
//                copies of labels / index lists / cluster labels / planes on a second copy stream.
// PCIe moves in both directions next to the compute; the device never waits for the host between chains.
// lpx_feeder_run_multi runs one such pipeline per context (chain k belongs to context k % C, one host thread each), so
// that several chains compute at once like in the device-resident case.  The lanes SHARE the copy streams (a small
// pool per direction, LPX_FEEDER_COPY_STREAMS): the link is one resource whatever the number of streams, the device
// serves about 24 hardware queues at full speed (three streams per lane made 20 lanes slower than 4), and nothing
// ever WAITS on a copy stream -- a lane issues a copy only when the host already knows that its buffer set is free
// (it has waited for that chain's compute event itself), so one lane's copies never hold up another's behind an
// event.  The 4-word frame counts are written by the chain's last kernel straight into pinned host memory.
// (Measured with GPU_MAX_HW_QUEUES=32, 2560 frames, chains of 64: 4.9 / 10.7 / 12.2 / 10.7 k frames/s on 1 / 4 / 10 /
// 20 lanes against 14.7 k device-resident on ten contexts; the link alone carries 22 k frames/s of this traffic.
// Replacing the 320 copies of a chain by two kernels that read / write the pinned memory themselves was built and
// measured: 8.0-8.7 k frames/s -- their workgroups queue for CUs behind the chains' -- so the copy engines stay.)
#include "lpx_internal.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

// device side of one pipeline: its copy streams, two buffer sets and their events
struct FeederLane
{
    hipStream_t h2d = nullptr, d2h = nullptr;  // borrowed from the feeder's pools
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_compute[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};
    void *d_in[2] = {nullptr, nullptr};
    void *d_out[2] = {nullptr, nullptr};
    size_t in_bytes = 0, out_bytes = 0;
    uint32_t *h_counts = nullptr;  // pinned, 2 sets x B x 4: written by the device (relabel_kernel), read by the host
    uint32_t *d_counts = nullptr;  // the device's address of h_counts
    uint32_t cap_b = 0, cap_pitch = 0, cap_P = 0;
    int rc = 0;
    char err[512] = {0};
};

struct lpx_feeder
{
    int device = 0;
    std::vector<lpx_pcd_info> info;
    std::vector<size_t> offset;  // byte offset of every frame in the pinned arena
    char *pinned = nullptr;      // all frames, hipHostMalloc
    size_t pinned_bytes = 0;
    uint32_t max_points = 0, max_step = 0;
    char err[512] = {0};
    std::vector<FeederLane *> lanes;  // device side of lpx_feeder_run*, one lane per context (sized on first use)
    std::vector<hipStream_t> h2d_pool, d2h_pool;  // copy streams shared by the lanes (lane c uses stream c % size)
};

static int ffail(lpx_feeder *f, int code, const char *fmt, const char *a, const char *b = "")
{
    if (f)
        snprintf(f->err, sizeof f->err, fmt, a, b);
    return code;
}

static int ffail(FeederLane *f, int code, const char *fmt, const char *a, const char *b = "")
{
    if (f)
        snprintf(f->err, sizeof f->err, fmt, a, b);
    return code;
}

// ------------------------------------------------------------------------------------------------
// pinned host memory
// ------------------------------------------------------------------------------------------------
extern "C" int lpx_host_alloc(void **out, size_t bytes)
{
    if (!out)
        return LPX_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return LPX_ERR_NO_DEVICE;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? LPX_OK : LPX_ERR_HIP;
}

extern "C" void lpx_host_free(void *p)
{
    if (p)
        hipHostFree(p);
}

// ------------------------------------------------------------------------------------------------
// binary PCD v0.7
// ------------------------------------------------------------------------------------------------
static int parse_header(FILE *fp, lpx_pcd_info *info, long *data_pos)
{
    char line[1024];
    std::vector<std::string> fields;
    std::vector<uint32_t> sizes, counts;
    std::vector<char> types;
    uint64_t width = 0, height = 1, points = 0;
    bool have_points = false, have_data = false;
    while (fgets(line, sizeof line, fp))
    {
        if (line[0] == '#')
            continue;
        char *save = nullptr;
        char *key = strtok_r(line, " \t\r\n", &save);
        if (!key)
            continue;
        std::vector<std::string> v;
        for (char *t = strtok_r(nullptr, " \t\r\n", &save); t; t = strtok_r(nullptr, " \t\r\n", &save))
            v.push_back(t);
        if (!strcmp(key, "FIELDS"))
            fields = v;
        else if (!strcmp(key, "SIZE"))
            for (auto &s : v)
                sizes.push_back((uint32_t)strtoul(s.c_str(), nullptr, 10));
        else if (!strcmp(key, "TYPE"))
            for (auto &s : v)
                types.push_back(s.empty() ? '?' : s[0]);
        else if (!strcmp(key, "COUNT"))
            for (auto &s : v)
                counts.push_back((uint32_t)strtoul(s.c_str(), nullptr, 10));
        else if (!strcmp(key, "WIDTH") && !v.empty())
            width = strtoull(v[0].c_str(), nullptr, 10);
        else if (!strcmp(key, "HEIGHT") && !v.empty())
            height = strtoull(v[0].c_str(), nullptr, 10);
        else if (!strcmp(key, "POINTS") && !v.empty())
        {
            points = strtoull(v[0].c_str(), nullptr, 10);
            have_points = true;
        }
        else if (!strcmp(key, "DATA"))
        {
            if (v.empty() || v[0] != "binary")
                return LPX_ERR_ARG;  // ascii / binary_compressed: not what the reference's data uses
            have_data = true;
            break;
        }
    }
    if (!have_data || fields.empty() || sizes.size() != fields.size() || types.size() != fields.size())
        return LPX_ERR_ARG;
    if (counts.empty())
        counts.assign(fields.size(), 1u);
    if (counts.size() != fields.size())
        return LPX_ERR_ARG;
    if (!have_points)
        points = width * height;
    if (points > 0xfffffff0ull)
        return LPX_ERR_ARG;
    uint32_t step = 0, off[3] = {0, 0, 0};
    bool found[3] = {false, false, false};
    for (size_t i = 0; i < fields.size(); ++i)
    {
        for (int a = 0; a < 3; ++a)
            if (fields[i] == (a == 0 ? "x" : a == 1 ? "y" : "z"))
            {
                if (sizes[i] != 4 || types[i] != 'F' || counts[i] != 1)
                    return LPX_ERR_ARG;  // the path computes on float32 coordinates
                off[a] = step;
                found[a] = true;
            }
        step += sizes[i] * counts[i];
    }
    if (!found[0] || !found[1] || !found[2] || step == 0)
        return LPX_ERR_ARG;
    info->n_points = (uint32_t)points;
    info->point_step = step;
    info->off_x = off[0];
    info->off_y = off[1];
    info->off_z = off[2];
    info->n_fields = (uint32_t)fields.size();
    *data_pos = ftell(fp);
    return LPX_OK;
}

extern "C" int lpx_pcd_info_read(const char *path, lpx_pcd_info *info)
{
    if (!path || !info)
        return LPX_ERR_ARG;
    FILE *fp = fopen(path, "rb");
    if (!fp)
        return LPX_ERR_ARG;
    long pos = 0;
    const int rc = parse_header(fp, info, &pos);
    fclose(fp);
    return rc;
}

// Reads exactly POINTS records into dst (the reference's files carry ~3.9 KB of trailing bytes after the payload,
// which are ignored).  dst may be pinned (lpx_host_alloc) or ordinary memory.
extern "C" int lpx_pcd_load(const char *path, void *dst, size_t dst_capacity_bytes, lpx_pcd_info *info)
{
    if (!path || !info || (!dst && dst_capacity_bytes))
        return LPX_ERR_ARG;
    FILE *fp = fopen(path, "rb");
    if (!fp)
        return LPX_ERR_ARG;
    long pos = 0;
    int rc = parse_header(fp, info, &pos);
    if (rc == LPX_OK)
    {
        const size_t need = (size_t)info->n_points * info->point_step;
        if (need > dst_capacity_bytes)
            rc = LPX_ERR_CAPACITY;
        else if (need && fread(dst, 1, need, fp) != need)
            rc = LPX_ERR_ARG;  // truncated payload
    }
    fclose(fp);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// feeder: every file preloaded into ONE pinned arena (src/dataloader.cpp:128-153 preloads all clouds too)
// ------------------------------------------------------------------------------------------------
extern "C" int lpx_feeder_create(int device, const char *const *paths, uint32_t n_files, lpx_feeder **out)
{
    if (!out || (!paths && n_files))
        return LPX_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return LPX_ERR_NO_DEVICE;
    if (device < 0 || device >= count || hipSetDevice(device) != hipSuccess)
        return LPX_ERR_ARG;
    lpx_feeder *f = new (std::nothrow) lpx_feeder();
    if (!f)
        return LPX_ERR_INTERNAL;
    f->device = device;
    size_t total = 0;
    for (uint32_t i = 0; i < n_files; ++i)
    {
        lpx_pcd_info inf;
        const int rc = lpx_pcd_info_read(paths[i], &inf);
        if (rc)
        {
            delete f;
            return rc;
        }
        f->info.push_back(inf);
        f->offset.push_back(total);
        total += ((size_t)inf.n_points * inf.point_step + 255) & ~(size_t)255;
        f->max_points = inf.n_points > f->max_points ? inf.n_points : f->max_points;
        f->max_step = inf.point_step > f->max_step ? inf.point_step : f->max_step;
    }
    if (hipHostMalloc((void **)&f->pinned, total ? total : 1, hipHostMallocDefault) != hipSuccess)
    {
        delete f;
        return LPX_ERR_HIP;
    }
    f->pinned_bytes = total;
    for (uint32_t i = 0; i < n_files; ++i)
    {
        lpx_pcd_info inf;
        const size_t cap = (i + 1 < n_files ? f->offset[i + 1] : total) - f->offset[i];
        const int rc = lpx_pcd_load(paths[i], f->pinned + f->offset[i], cap, &inf);  // file -> pinned, no staging copy
        if (rc || inf.n_points != f->info[i].n_points || inf.point_step != f->info[i].point_step)
        {
            hipHostFree(f->pinned);
            delete f;
            return rc ? rc : LPX_ERR_ARG;
        }
    }
    *out = f;
    return LPX_OK;
}

static void lane_release(FeederLane *f)
{
    for (int s = 0; s < 2; ++s)
    {
        if (f->d_in[s])
            hipFree(f->d_in[s]);
        if (f->d_out[s])
            hipFree(f->d_out[s]);
        f->d_in[s] = f->d_out[s] = nullptr;
        if (f->ev_h2d[s])
            hipEventDestroy(f->ev_h2d[s]);
        if (f->ev_compute[s])
            hipEventDestroy(f->ev_compute[s]);
        if (f->ev_d2h[s])
            hipEventDestroy(f->ev_d2h[s]);
        f->ev_h2d[s] = f->ev_compute[s] = f->ev_d2h[s] = nullptr;
    }
    if (f->h_counts)
        hipHostFree(f->h_counts);
    f->h_counts = nullptr;
    f->d_counts = nullptr;
    f->cap_b = f->cap_pitch = f->cap_P = 0;
}

static void feeder_release_device(lpx_feeder *f)
{
    for (FeederLane *l : f->lanes)
    {
        lane_release(l);
        delete l;
    }
    f->lanes.clear();
    for (hipStream_t st : f->h2d_pool)
        hipStreamDestroy(st);
    for (hipStream_t st : f->d2h_pool)
        hipStreamDestroy(st);
    f->h2d_pool.clear();
    f->d2h_pool.clear();
}

extern "C" void lpx_feeder_destroy(lpx_feeder *f)
{
    if (!f)
        return;
    hipSetDevice(f->device);
    hipDeviceSynchronize();
    feeder_release_device(f);
    if (f->pinned)
        hipHostFree(f->pinned);
    delete f;
}

extern "C" uint32_t lpx_feeder_frames(const lpx_feeder *f)
{
    return f ? (uint32_t)f->info.size() : 0u;
}

extern "C" const void *lpx_feeder_frame(const lpx_feeder *f, uint32_t i, lpx_pcd_info *info)
{
    if (!f || i >= f->info.size())
        return nullptr;
    if (info)
        *info = f->info[i];
    return f->pinned + f->offset[i];
}

extern "C" const char *lpx_feeder_last_error(const lpx_feeder *f)
{
    return f ? f->err : "no feeder";
}

// per-set device output layout: [labels | gidx | oidx | clabels] B x pitch words each, planes B x 4P, counts B x 4
struct OutSet
{
    uint32_t *labels, *gidx, *oidx;
    int32_t *clabels;
    float *planes;
    uint32_t *counts;
};

static OutSet out_set(void *base, uint32_t B, uint32_t pitch, uint32_t P)
{
    OutSet o;
    uint32_t *w = (uint32_t *)base;
    const size_t a = (size_t)B * pitch;
    o.labels = w;
    o.gidx = w + a;
    o.oidx = w + 2 * a;
    o.clabels = (int32_t *)(w + 3 * a);
    o.planes = (float *)(w + 4 * a);
    o.counts = w + 4 * a + (size_t)B * 4 * P;
    return o;
}

#define FHIP(f, call)                                                                                                \
    do                                                                                                               \
    {                                                                                                                \
        hipError_t e_ = (call);                                                                                      \
        if (e_ != hipSuccess)                                                                                        \
            return ffail((f), LPX_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));                           \
    } while (0)

static int lane_prepare(FeederLane *f, uint32_t max_step, uint32_t B, uint32_t pitch, uint32_t P)
{
    if (f->cap_b >= B && f->cap_pitch >= pitch && f->cap_P >= P && f->d_in[0])
        return LPX_OK;
    hipDeviceSynchronize();
    lane_release(f);
    f->in_bytes = (size_t)B * pitch * max_step;
    f->out_bytes = sizeof(uint32_t) * ((size_t)B * pitch * 4 + (size_t)B * 4 * P + (size_t)B * 4) + 256;
    for (int s = 0; s < 2; ++s)
    {
        FHIP(f, hipMalloc(&f->d_in[s], f->in_bytes ? f->in_bytes : 1));
        FHIP(f, hipMalloc(&f->d_out[s], f->out_bytes));
        FHIP(f, hipEventCreateWithFlags(&f->ev_h2d[s], hipEventDisableTiming));
        FHIP(f, hipEventCreateWithFlags(&f->ev_compute[s], hipEventDisableTiming));
        FHIP(f, hipEventCreateWithFlags(&f->ev_d2h[s], hipEventDisableTiming));
    }
    FHIP(f, hipHostMalloc((void **)&f->h_counts, sizeof(uint32_t) * 2 * (size_t)B * 4, hipHostMallocDefault));
    FHIP(f, hipHostGetDevicePointer((void **)&f->d_counts, f->h_counts, 0));
    memset(f->h_counts, 0, sizeof(uint32_t) * 2 * (size_t)B * 4);
    f->cap_b = B;
    f->cap_pitch = pitch;
    f->cap_P = P;
    return LPX_OK;
}

// The pipeline of one lane: chains first, first + stride, ... of the run through `ctx`.
namespace
{
struct RunArgs
{
    lpx_feeder *f;
    const uint32_t *frame_ids;
    uint32_t n_frames, B, P, step, n_chains;
    uint32_t offs[3];
    const lpx_seg_cfg *seg_cfg;
    const lpx_clu_cfg *clu_cfg;
    const lpx_stream_out *out;
};

int lane_run(FeederLane *l, lpx_ctx *ctx, const RunArgs &a, uint32_t first, uint32_t stride)
{
    lpx_feeder *f = a.f;
    const lpx_stream_out *out = a.out;
    const uint32_t B = a.B, P = a.P, step = a.step;
    const size_t up = out->frame_pitch;
    FHIP(l, hipSetDevice(f->device));

    // exact-size D2H of the results of the lane's j-th chain once it has computed (the host waits for it while the
    // lane's next chain runs)
    int flagged_rc = LPX_OK;  // first frame of this lane that came back with a device status
    char flagged_msg[96] = {0};
    auto drain = [&](uint32_t j) -> int {
        const int s = (int)(j & 1u);
        const uint32_t k = first + j * stride;
        const uint32_t lo = k * B, nb = (a.n_frames - lo < B) ? a.n_frames - lo : B;
        const OutSet o = out_set(l->d_out[s], B, l->cap_pitch, l->cap_P);
        const uint32_t *hc = l->h_counts + (size_t)s * B * 4;  // written by the chain's last kernel
        FHIP(l, hipEventSynchronize(l->ev_compute[s]));
        for (uint32_t b = 0; b < nb; ++b)
        {
            const uint32_t fid = a.frame_ids[lo + b], n = f->info[fid].n_points;
            const uint32_t ng = hc[4 * b], no = hc[4 * b + 1];
            const size_t dst = (size_t)(lo + b) * up, src = (size_t)b * l->cap_pitch;
            memcpy(out->counts + 4 * (size_t)(lo + b), hc + 4 * b, 16);
            if (n)
                FHIP(l, hipMemcpyAsync(out->labels + dst, o.labels + src, 4 * (size_t)n, hipMemcpyDeviceToHost, l->d2h));
            if (ng)
                FHIP(l, hipMemcpyAsync(out->ground_idx + dst, o.gidx + src, 4 * (size_t)ng, hipMemcpyDeviceToHost, l->d2h));
            if (no)
            {
                FHIP(l, hipMemcpyAsync(out->obstacle_idx + dst, o.oidx + src, 4 * (size_t)no, hipMemcpyDeviceToHost, l->d2h));
                FHIP(l, hipMemcpyAsync(out->cluster_labels + dst, o.clabels + src, 4 * (size_t)no, hipMemcpyDeviceToHost,
                                       l->d2h));
            }
        }
        if (out->planes)
            FHIP(l, hipMemcpyAsync(out->planes + (size_t)lo * 4 * P, o.planes, sizeof(float) * 4 * P * nb,
                                   hipMemcpyDeviceToHost, l->d2h));
        FHIP(l, hipEventRecord(l->ev_d2h[s], l->d2h));
        // a frame the device flagged (non-finite coordinates; neighbour lists that did not fit a LISTS-mode context)
        // has no valid labels: the run reports the FIRST such frame of the lane instead of LPX_OK -- after every chain
        // of the lane has been processed and drained, so that counts[] really holds every frame's status and no copy
        // is left in flight
        for (uint32_t b = 0; b < nb && flagged_rc == LPX_OK; ++b)
            if (hc[4 * b + 3] != 0u)
            {
                flagged_rc = -(int)hc[4 * b + 3];
                snprintf(flagged_msg, sizeof flagged_msg, "frame %u of the run (file %u): device status %d", lo + b,
                         a.frame_ids[lo + b], flagged_rc);
            }
        return LPX_OK;
    };

    uint32_t n_pts[LPX_MAX_BATCH];
    uint32_t j = 0;
    int rc;
    for (uint32_t k = first; k < a.n_chains; k += stride, ++j)
    {
        const int s = (int)(j & 1u);
        const uint32_t lo = k * B, nb = (a.n_frames - lo < B) ? a.n_frames - lo : B;
        // inputs of set s are free once the lane's chain j - 2 has computed: drain(j - 2) has waited for exactly that
        // on the host already, so the shared copy stream is never made to wait
        for (uint32_t b = 0; b < nb; ++b)
        {
            const uint32_t fid = a.frame_ids[lo + b];
            n_pts[b] = f->info[fid].n_points;
            if (n_pts[b])
                FHIP(l, hipMemcpyAsync((char *)l->d_in[s] + (size_t)b * l->cap_pitch * step, f->pinned + f->offset[fid],
                                       (size_t)n_pts[b] * step, hipMemcpyHostToDevice, l->h2d));
        }
        FHIP(l, hipEventRecord(l->ev_h2d[s], l->h2d));
        FHIP(l, hipStreamWaitEvent(ctx->stream, l->ev_h2d[s], 0));
        if (j >= 2)
            FHIP(l, hipStreamWaitEvent(ctx->stream, l->ev_d2h[s], 0));  // chain j - 2's results have left set s
        const OutSet o = out_set(l->d_out[s], B, l->cap_pitch, l->cap_P);
        rc = lpx_batch_impl(ctx, nb, l->d_in[s], step, a.offs, l->cap_pitch, n_pts, a.seg_cfg, a.clu_cfg, o.labels, o.gidx,
                            o.oidx, o.planes, o.clabels, l->d_counts + (size_t)s * B * 4);
        if (rc)
            return ffail(l, rc, "%s", lpx_last_error(ctx));
        FHIP(l, hipEventRecord(l->ev_compute[s], ctx->stream));
        if (j >= 1 && (rc = drain(j - 1)))
            return rc;
    }
    if (j && (rc = drain(j - 1)))
        return rc;
    for (int s = 0; s < 2 && (uint32_t)s < j; ++s)  // the lane's own last copies (the stream is shared)
        FHIP(l, hipEventSynchronize(l->ev_d2h[s]));
    return flagged_rc ? ffail(l, flagged_rc, "%s", flagged_msg) : LPX_OK;
}
}  // namespace

// Frames frame_ids[0 .. n_frames) of the feeder through n_ctx batch contexts of equal slot count B (the chain
// length): chain k -- frames [k B, (k+1) B) -- runs on context k % n_ctx, every context with its own
// buffer sets and host thread (the copy streams are shared).  Results to host arrays pitched by out->frame_pitch elements per frame
// (planes: 4 P floats, counts: 4 words {n_ground, n_obstacle, n_clusters, status}).  Synchronous at return.
extern "C" int lpx_feeder_run_multi(lpx_feeder *f, lpx_ctx *const *ctxs, uint32_t n_ctx, const uint32_t *frame_ids,
                                    uint32_t n_frames, const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg,
                                    const lpx_stream_out *out)
{
    if (!f || !ctxs || n_ctx == 0 || !seg_cfg || !clu_cfg || !out || (!frame_ids && n_frames))
        return LPX_ERR_ARG;
    if (!out->labels || !out->ground_idx || !out->obstacle_idx || !out->cluster_labels || !out->counts)
        return ffail(f, LPX_ERR_ARG, "%s", "every output array except planes is required");
    if (n_frames == 0)
        return LPX_OK;
    for (uint32_t c = 0; c < n_ctx; ++c)
    {
        if (!ctxs[c])
            return LPX_ERR_ARG;
        if (ctxs[c]->device != f->device)
            return ffail(f, LPX_ERR_ARG, "%s", "feeder and context live on different devices");
        if (ctxs[c]->batch != ctxs[0]->batch)
            return ffail(f, LPX_ERR_ARG, "%s", "the contexts of one run must have the same number of frame slots");
        for (uint32_t e = 0; e < c; ++e)
            if (ctxs[e] == ctxs[c])
                return ffail(f, LPX_ERR_ARG, "%s", "the same context twice");
    }
    RunArgs a;
    a.f = f;
    a.frame_ids = frame_ids;
    a.n_frames = n_frames;
    a.B = ctxs[0]->batch;
    a.P = seg_cfg->number_of_planar_partitions;
    a.seg_cfg = seg_cfg;
    a.clu_cfg = clu_cfg;
    a.out = out;
    uint32_t pitch = 0;
    a.step = 0;
    for (uint32_t j = 0; j < n_frames; ++j)
    {
        if (frame_ids[j] >= f->info.size())
            return ffail(f, LPX_ERR_ARG, "%s", "frame id out of range");
        const lpx_pcd_info &inf = f->info[frame_ids[j]];
        pitch = inf.n_points > pitch ? inf.n_points : pitch;
        if (j == 0)
            a.step = inf.point_step;
        const lpx_pcd_info &i0 = f->info[frame_ids[0]];
        if (inf.point_step != a.step || inf.off_x != i0.off_x || inf.off_y != i0.off_y || inf.off_z != i0.off_z)
            return ffail(f, LPX_ERR_ARG, "%s", "the frames of one run must share a record layout");
    }
    if (pitch > out->frame_pitch)
        return ffail(f, LPX_ERR_ARG, "%s", "a frame holds more points than out->frame_pitch");
    const lpx_pcd_info lay = f->info[frame_ids[0]];
    a.offs[0] = lay.off_x;
    a.offs[1] = lay.off_y;
    a.offs[2] = lay.off_z;
    a.n_chains = (n_frames + a.B - 1) / a.B;
    if (hipSetDevice(f->device) != hipSuccess)
        return ffail(f, LPX_ERR_HIP, "%s", "hipSetDevice failed");
    const uint32_t lanes = n_ctx < a.n_chains ? n_ctx : a.n_chains;
    while (f->lanes.size() < lanes)
        f->lanes.push_back(new FeederLane());
    static const int pool_env = LPX_KNOB("LPX_FEEDER_COPY_STREAMS") ? atoi(LPX_KNOB("LPX_FEEDER_COPY_STREAMS")) : 2;
    const size_t pool = (size_t)(pool_env < 1 ? 1 : (pool_env > 16 ? 16 : pool_env));
    while (f->h2d_pool.size() < pool)
    {
        hipStream_t up = nullptr, down = nullptr;
        if (hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&down, hipStreamNonBlocking) != hipSuccess)
            return ffail(f, LPX_ERR_HIP, "%s", "hipStreamCreateWithFlags failed");
        f->h2d_pool.push_back(up);
        f->d2h_pool.push_back(down);
    }
    int rc;
    for (uint32_t c = 0; c < lanes; ++c)
    {
        f->lanes[c]->h2d = f->h2d_pool[c % pool];
        f->lanes[c]->d2h = f->d2h_pool[c % pool];
        if ((rc = lane_prepare(f->lanes[c], f->max_step, a.B, pitch, a.P)))
            return ffail(f, rc, "%s", f->lanes[c]->err);
        if ((rc = lpx_reserve(ctxs[c], pitch, 0)))
            return ffail(f, rc, "%s", lpx_last_error(ctxs[c]));
    }
    if (lanes == 1)
        rc = f->lanes[0]->rc = lane_run(f->lanes[0], ctxs[0], a, 0, 1);
    else
    {
        std::vector<std::thread> th;
        for (uint32_t c = 0; c < lanes; ++c)
            th.emplace_back([&, c] { f->lanes[c]->rc = lane_run(f->lanes[c], ctxs[c], a, c, lanes); });
        for (std::thread &t : th)
            t.join();
    }
    for (uint32_t c = 0; c < lanes; ++c)
        if (f->lanes[c]->rc)
        {
            // a lane that stopped early may have left work in flight on the others' streams: settle before returning
            hipDeviceSynchronize();
            return ffail(f, f->lanes[c]->rc, "%s", f->lanes[c]->err);
        }
    return LPX_OK;
}

extern "C" int lpx_feeder_run(lpx_feeder *f, lpx_ctx *ctx, const uint32_t *frame_ids, uint32_t n_frames,
                              const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg, const lpx_stream_out *out)
{
    return lpx_feeder_run_multi(f, &ctx, ctx ? 1u : 0u, frame_ids, n_frames, seg_cfg, clu_cfg, out);
}
