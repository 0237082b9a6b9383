/*
 * lidar_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY:
 * the parity checker for the MI355X path and the timed "port" CPU baseline in bench.py.
 * Never linked into the product library.  See lidar_oracle.h for the pinning status.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off: the reference is built for baseline x86-64 (CMakeLists.txt:1-2, build.sh:13),
 * i.e. no FMA; every float expression below must round after each operation.
 */
#include "lidar_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                     */
/* ------------------------------------------------------------------------------------------ */

static inline const float *pt_at(const void *pts, size_t stride, uint32_t i)
{
    return (const float *)((const char *)pts + (size_t)i * stride);
}

typedef struct
{
    float key;
    uint32_t idx;
} key_idx;

/* total order (key, idx): what a stable sort of iota by key gives.  Canonical tie rule (H2). */
static int cmp_key_idx(const void *pa, const void *pb);
static void sort_key_idx(void *base, size_t n);
static int cmp_key_idx(const void *pa, const void *pb)
{
    const key_idx *a = (const key_idx *)pa;
    const key_idx *b = (const key_idx *)pb;
    if (a->key < b->key)
        return -1;
    if (b->key < a->key)
        return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* ------------------------------------------------------------------------------------------ */
/* 3x3 Jacobi SVD: Eigen 3.4 JacobiSVD<Matrix3f>::compute as used at src/segmentation.cpp:87-94 */
/* (Eigen/src/SVD/JacobiSVD.h, Eigen/src/Jacobi/Jacobi.h; square real case, V only).            */
/* ------------------------------------------------------------------------------------------ */

/* The reference's only parallelism is these two index sorts: std::sort(std::execution::par) maps onto TBB
 * (src/segmentation.cpp:119-122, :165-168).  For the "reference-like" CPU baseline of bench.py the restatement
 * can run them on several threads too: chunks are qsort'ed concurrently and merged pairwise (the order is total,
 * so the result does not depend on the thread count).  Default 1 thread. */
#include <pthread.h>
static int g_sort_threads = 1;

void orc_set_sort_threads(int n)
{
    g_sort_threads = n < 1 ? 1 : n;
}

typedef struct
{
    key_idx *a, *tmp;
    size_t lo, mid, hi;
} sort_job;

static void *sort_chunk_job(void *arg)
{
    sort_job *j = (sort_job *)arg;
    qsort(j->a + j->lo, j->hi - j->lo, sizeof(key_idx), cmp_key_idx);
    return NULL;
}

static void *merge_job(void *arg)
{
    sort_job *j = (sort_job *)arg;
    size_t i = j->lo, k = j->mid, o = j->lo;
    while (i < j->mid && k < j->hi)
        j->tmp[o++] = cmp_key_idx(&j->a[k], &j->a[i]) < 0 ? j->a[k++] : j->a[i++];
    while (i < j->mid)
        j->tmp[o++] = j->a[i++];
    while (k < j->hi)
        j->tmp[o++] = j->a[k++];
    memcpy(j->a + j->lo, j->tmp + j->lo, sizeof(key_idx) * (j->hi - j->lo));
    return NULL;
}

static void sort_key_idx(void *base, size_t n)
{
    size_t t = (size_t)g_sort_threads;
    if (t > n / 8192)
        t = n / 8192; /* grain: threads only pay off on chunks of thousands of keys */
    if (t < 2)
    {
        qsort(base, n, sizeof(key_idx), cmp_key_idx);
        return;
    }
    key_idx *a = (key_idx *)base, *tmp = (key_idx *)malloc(sizeof(key_idx) * n);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * t);
    sort_job *jobs = (sort_job *)malloc(sizeof(sort_job) * t);
    size_t *cut = (size_t *)malloc(sizeof(size_t) * (t + 1));
    for (size_t c = 0; c <= t; ++c)
        cut[c] = n * c / t;
    for (size_t c = 0; c < t; ++c)
    {
        jobs[c] = (sort_job){a, tmp, cut[c], 0, cut[c + 1]};
        pthread_create(&th[c], NULL, sort_chunk_job, &jobs[c]);
    }
    for (size_t c = 0; c < t; ++c)
        pthread_join(th[c], NULL);
    for (size_t width = 1; width < t; width *= 2) /* pairwise merges, each round in parallel */
    {
        size_t nj = 0;
        for (size_t c = 0; c + width < t; c += 2 * width)
        {
            const size_t hi = c + 2 * width < t ? c + 2 * width : t;
            jobs[nj] = (sort_job){a, tmp, cut[c], cut[c + width], cut[hi]};
            pthread_create(&th[nj], NULL, merge_job, &jobs[nj]);
            ++nj;
        }
        for (size_t c = 0; c < nj; ++c)
            pthread_join(th[c], NULL);
    }
    free(cut);
    free(jobs);
    free(th);
    free(tmp);
}

typedef struct
{
    float c, s;
} jrot;

/* JacobiRotation::makeJacobi(x, y, z) */
static jrot make_jacobi(float x, float y, float z)
{
    jrot j;
    const float deno = 2.0f * fabsf(y);
    if (deno < FLT_MIN)
    {
        j.c = 1.0f;
        j.s = 0.0f;
        return j;
    }
    const float tau = (x - z) / deno;
    const float w = sqrtf(tau * tau + 1.0f);
    float t;
    if (tau > 0.0f)
        t = 1.0f / (tau + w);
    else
        t = 1.0f / (tau - w);
    const float sign_t = t > 0.0f ? 1.0f : -1.0f;
    const float n = 1.0f / sqrtf(t * t + 1.0f);
    j.s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
    j.c = n;
    return j;
}

/* apply_rotation_in_the_plane on rows p,q of a 3x3 (applyOnTheLeft) */
static void rot_left(float *w, int p, int q, jrot j)
{
    if (j.c == 1.0f && j.s == 0.0f)
        return;
    for (int i = 0; i < 3; ++i)
    {
        const float xi = w[p * 3 + i], yi = w[q * 3 + i];
        w[p * 3 + i] = j.c * xi + j.s * yi;
        w[q * 3 + i] = -j.s * xi + j.c * yi;
    }
}

/* applyOnTheRight(p,q,j) == rotation in the plane of columns p,q with j.transpose() = (c,-s) */
static void rot_right(float *w, int p, int q, jrot j)
{
    const float c = j.c, s = -j.s;
    if (c == 1.0f && s == 0.0f)
        return;
    for (int i = 0; i < 3; ++i)
    {
        const float xi = w[i * 3 + p], yi = w[i * 3 + q];
        w[i * 3 + p] = c * xi + s * yi;
        w[i * 3 + q] = -s * xi + c * yi;
    }
}

/* Test instrumentation (tests/golden/make_jacobi_real.py): every plane fit of orc_segment / orc_plane_from_points can
 * be recorded -- the float32 covariance handed to the 3x3 solve, the plane it produced, the sweeps the Jacobi loop
 * took and the point count -- so that the restated solve is checked against float64 LAPACK on the covariances REAL
 * frames produce, not only on synthetic matrices.  Off unless orc_trace_fits() hands over a buffer; single-threaded
 * (the fits of orc_segment run on the calling thread). */
#define ORC_FIT_RECORD 16 /* floats: cov[9], plane[4], sweeps, n, failed */
static float *g_fit_trace = NULL;
static uint32_t g_fit_cap = 0, g_fit_count = 0;
static int g_last_sweeps = 0;

void orc_trace_fits(float *buf, uint32_t cap_records)
{
    g_fit_trace = buf;
    g_fit_cap = buf ? cap_records : 0;
    g_fit_count = 0;
}

uint32_t orc_trace_count(void)
{
    return g_fit_count;
}

/* returns 0 on success, 1 if the input is not finite (Eigen: InvalidInput) */
static int jacobi_svd3(const float *a, float *v, float *sigma)
{
    g_last_sweeps = 0;
    float w[9];
    float scale = 0.0f;
    for (int i = 0; i < 9; ++i)
    {
        const float m = fabsf(a[i]);
        if (!(m <= scale)) /* PropagateNaN max */
            scale = m;
    }
    if (!isfinite(scale))
        return 1;
    if (scale == 0.0f)
        scale = 1.0f;
    for (int i = 0; i < 9; ++i)
        w[i] = a[i] / scale;
    for (int i = 0; i < 9; ++i)
        v[i] = (i % 4 == 0) ? 1.0f : 0.0f;

    const float precision = 2.0f * FLT_EPSILON;
    const float consider_as_zero = FLT_MIN;
    float max_diag = fmaxf(fabsf(w[0]), fmaxf(fabsf(w[4]), fabsf(w[8])));

    int finished = 0;
    while (!finished)
    {
        finished = 1;
        ++g_last_sweeps;
        for (int p = 1; p < 3; ++p)
        {
            for (int q = 0; q < p; ++q)
            {
                const float threshold = fmaxf(consider_as_zero, precision * max_diag);
                if (fabsf(w[p * 3 + q]) > threshold || fabsf(w[q * 3 + p]) > threshold)
                {
                    finished = 0;
                    /* real_2x2_jacobi_svd */
                    float m00 = w[p * 3 + p], m01 = w[p * 3 + q], m10 = w[q * 3 + p], m11 = w[q * 3 + q];
                    jrot rot1;
                    const float t = m00 + m11;
                    const float d = m10 - m01;
                    if (fabsf(d) < FLT_MIN)
                    {
                        rot1.s = 0.0f;
                        rot1.c = 1.0f;
                    }
                    else
                    {
                        const float u = t / d;
                        const float tmp = sqrtf(1.0f + u * u);
                        rot1.s = 1.0f / tmp;
                        rot1.c = u / tmp;
                    }
                    /* m.applyOnTheLeft(0,1,rot1) */
                    if (!(rot1.c == 1.0f && rot1.s == 0.0f))
                    {
                        const float a0 = m00, a1 = m01, b0 = m10, b1 = m11;
                        m00 = rot1.c * a0 + rot1.s * b0;
                        m01 = rot1.c * a1 + rot1.s * b1;
                        m10 = -rot1.s * a0 + rot1.c * b0;
                        m11 = -rot1.s * a1 + rot1.c * b1;
                    }
                    const jrot j_right = make_jacobi(m00, m01, m11);
                    /* j_left = rot1 * j_right.transpose() */
                    jrot j_left;
                    {
                        const float c2 = j_right.c, s2 = -j_right.s;
                        j_left.c = rot1.c * c2 - rot1.s * s2;
                        j_left.s = rot1.c * s2 + rot1.s * c2;
                    }
                    rot_left(w, p, q, j_left);
                    rot_right(w, p, q, j_right);
                    rot_right(v, p, q, j_right);
                    max_diag = fmaxf(max_diag, fmaxf(fabsf(w[p * 3 + p]), fabsf(w[q * 3 + q])));
                }
            }
        }
    }
    float sv[3];
    for (int i = 0; i < 3; ++i)
        sv[i] = fabsf(w[i * 3 + i]) * scale;
    /* selection sort, descending, swapping columns of V */
    for (int i = 0; i < 3; ++i)
    {
        int pos = i;
        float best = sv[i];
        for (int k = i + 1; k < 3; ++k)
            if (sv[k] > best)
            {
                best = sv[k];
                pos = k;
            }
        if (best == 0.0f)
            break;
        if (pos != i)
        {
            const float ts = sv[i];
            sv[i] = sv[pos];
            sv[pos] = ts;
            for (int r = 0; r < 3; ++r)
            {
                const float tv = v[r * 3 + i];
                v[r * 3 + i] = v[r * 3 + pos];
                v[r * 3 + pos] = tv;
            }
        }
    }
    if (sigma)
        memcpy(sigma, sv, sizeof sv);
    return 0;
}

void orc_jacobi_svd3(const float *a, float *v, float *sigma)
{
    jacobi_svd3(a, v, sigma);
}

/* ------------------------------------------------------------------------------------------ */
/* plane estimate: src/segmentation.cpp:62-102, canonical arithmetic                          */
/*   centroid / covariance from EXACT integer moments of coordinates rounded to 2^-16 m        */
/*   (order independent, so a parallel accumulation reproduces it bit for bit), then the       */
/*   float 3x3 Jacobi SVD above; normal = V.col(2), d = normal . centroid.                     */
/* ------------------------------------------------------------------------------------------ */

#define FIX_SCALE 65536.0f    /* 2^16 */
#define FIX_CLAMP 16777216.0f /* 2^24 m: |q| <= 2^40, products <= 2^80, sums over < 2^23 points fit 128 bits */

typedef __int128 i128;

typedef struct
{
    uint64_t n;
    i128 sx, sy, sz, sxx, sxy, sxz, syy, syz, szz;
} moments;

/* Any finite coordinate is accepted (the reference processes any finite float, src/segmentation.cpp:311-345).
 * Beyond +-2^24 m (further than any Earth-fixed frame reaches) the MOMENTS use the clamped value so that the
 * sums stay exact integers; the inlier test always uses the float coordinate itself. */
static inline int to_fix(float v, int64_t *q)
{
    if (!isfinite(v))
        return 1;
    if (v > FIX_CLAMP)
        v = FIX_CLAMP;
    if (v < -FIX_CLAMP)
        v = -FIX_CLAMP;
    *q = (int64_t)llrintf(v * FIX_SCALE); /* round-half-even; the product by 2^16 is exact */
    return 0;
}

static inline int mom_add(moments *m, const float *p)
{
    int64_t x, y, z;
    if (to_fix(p[0], &x) | to_fix(p[1], &y) | to_fix(p[2], &z))
        return 1;
    m->n += 1;
    m->sx += x;
    m->sy += y;
    m->sz += z;
    m->sxx += (i128)x * x;
    m->sxy += (i128)x * y;
    m->sxz += (i128)x * z;
    m->syy += (i128)y * y;
    m->syz += (i128)y * z;
    m->szz += (i128)z * z;
    return 0;
}

static double i128_to_double(i128 v)
{
    const int neg = v < 0;
    const unsigned __int128 mag = neg ? (unsigned __int128)0 - (unsigned __int128)v : (unsigned __int128)v;
    const uint64_t hi = (uint64_t)(mag >> 64), lo = (uint64_t)mag;
    const double d = (double)hi * 18446744073709551616.0 + (double)lo;
    return neg ? -d : d;
}

/* returns 0 ok, 1 failure (n<3 or non-finite) */
static int plane_from_moments(const moments *m, float *plane)
{
    if (m->n < 3)
        return 1;
    const double n = (double)m->n;
    const double den = n * (double)(m->n - 1);
    const double inv20 = 1.0 / 65536.0, inv40 = inv20 * inv20; /* 2^-16, 2^-32 */
    const float cx = (float)((i128_to_double(m->sx) / n) * inv20);
    const float cy = (float)((i128_to_double(m->sy) / n) * inv20);
    const float cz = (float)((i128_to_double(m->sz) / n) * inv20);
    const i128 N = (i128)m->n;
    const float cxx = (float)((i128_to_double(N * m->sxx - m->sx * m->sx) / den) * inv40);
    const float cxy = (float)((i128_to_double(N * m->sxy - m->sx * m->sy) / den) * inv40);
    const float cxz = (float)((i128_to_double(N * m->sxz - m->sx * m->sz) / den) * inv40);
    const float cyy = (float)((i128_to_double(N * m->syy - m->sy * m->sy) / den) * inv40);
    const float cyz = (float)((i128_to_double(N * m->syz - m->sy * m->sz) / den) * inv40);
    const float czz = (float)((i128_to_double(N * m->szz - m->sz * m->sz) / den) * inv40);
    const float cov[9] = {cxx, cxy, cxz, cxy, cyy, cyz, cxz, cyz, czz};
    float v[9];
    const int failed = jacobi_svd3(cov, v, NULL);
    const float a = v[2], b = v[5], c = v[8];
    if (!failed)
    {
        plane[0] = a;
        plane[1] = b;
        plane[2] = c;
        plane[3] = (a * cx + b * cy) + c * cz;
    }
    if (g_fit_trace && g_fit_count < g_fit_cap)
    {
        float *r = g_fit_trace + (size_t)ORC_FIT_RECORD * g_fit_count++;
        memcpy(r, cov, sizeof cov);
        r[9] = failed ? 0.0f : plane[0];
        r[10] = failed ? 0.0f : plane[1];
        r[11] = failed ? 0.0f : plane[2];
        r[12] = failed ? 0.0f : plane[3];
        r[13] = (float)g_last_sweeps;
        r[14] = (float)m->n;
        r[15] = (float)failed;
    }
    return failed;
}

int orc_plane_from_points(const float *xyz, uint32_t n, float *plane)
{
    moments m;
    memset(&m, 0, sizeof m);
    for (uint32_t i = 0; i < n; ++i)
        if (mom_add(&m, xyz + 3 * (size_t)i))
            return ORC_ERR_RANGE;
    return plane_from_moments(&m, plane) ? 1 : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Segmenter::segment                                                                          */
/* ------------------------------------------------------------------------------------------ */

int orc_segment(const void *pts, size_t stride, uint32_t n, const orc_seg_cfg *cfg, uint32_t *labels,
                uint32_t *ground_idx, uint32_t *n_ground, uint32_t *obstacle_idx, uint32_t *n_obstacle,
                float *planes, uint32_t *seg_status)
{
    const uint32_t P = cfg->number_of_planar_partitions;
    uint32_t ng = 0, no = 0;
    *n_ground = 0;
    *n_obstacle = 0;
    /* src/segmentation.cpp:315 resizes with UNKNOWN; we always write UNKNOWN (Q3) */
    for (uint32_t i = 0; i < n; ++i)
        labels[i] = ORC_LABEL_UNKNOWN;
    if (planes)
        memset(planes, 0, sizeof(float) * 4 * (size_t)P);
    if (seg_status)
        for (uint32_t s = 0; s < P; ++s)
            seg_status[s] = ORC_SEG_TOO_FEW_POINTS;
    if (n == 0 || P == 0)
        return ORC_OK;
    /* non-finite coordinates are undefined behaviour upstream (comparators on NaN): rejected */
    for (uint32_t i = 0; i < n; ++i)
    {
        const float *p = pt_at(pts, stride, i);
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2]))
            return ORC_ERR_RANGE;
    }

    /* form_planar_partitions, :104-149: argsort by x (canonical ties: by index), P equal slabs */
    key_idx *sx = (key_idx *)malloc(sizeof(key_idx) * n);
    for (uint32_t i = 0; i < n; ++i)
    {
        sx[i].key = pt_at(pts, stride, i)[0];
        sx[i].idx = i;
    }
    sort_key_idx(sx, n); /* :119-122 std::sort(std::execution::par, ...) */

    const uint32_t n_per = n / P; /* :124 */
    key_idx *sz = (key_idx *)malloc(sizeof(key_idx) * (n_per ? n_per : 1));
    uint8_t *is_ground = (uint8_t *)malloc(n_per ? n_per : 1);
    uint32_t *zord_ground = (uint32_t *)malloc(sizeof(uint32_t) * (n_per ? n_per : 1));
    int rc = ORC_OK;

    for (uint32_t s = 0; s < P && rc == ORC_OK; ++s)
    {
        const key_idx *seg = sx + (size_t)s * n_per; /* in-segment position k -> seg[k].idx */
        const uint32_t ns = n_per;
        if (ns < 3) /* :224-229 */
            continue;

        /* extract_initial_seeds, :151-217 */
        for (uint32_t k = 0; k < ns; ++k)
        {
            sz[k].key = pt_at(pts, stride, seg[k].idx)[2];
            sz[k].idx = k;
        }
        sort_key_idx(sz, ns); /* :165-168 std::sort(std::execution::par, ...) */
        const float z_floor = -1.5f * cfg->sensor_height_m; /* :171 */
        uint32_t cut_lo = 0;
        for (uint32_t i = 0; i < ns; ++i)
            if (sz[i].key > z_floor)
            {
                cut_lo = i;
                break;
            }
        const key_idx *rem = sz + cut_lo; /* :182 erase prefix */
        const uint32_t nrem = ns - cut_lo;
        uint32_t n_rep = nrem < cfg->number_of_lower_point_representatives
                             ? nrem
                             : cfg->number_of_lower_point_representatives;
        float z_mean = 0.0f;
        for (uint32_t i = 0; i < n_rep; ++i) /* :193-197 sequential float sum, ascending z */
            z_mean += rem[i].key;
        z_mean /= (float)n_rep;
        const float z_max = z_mean + cfg->initial_seed_threshold;
        uint32_t n_seed = 0;
        for (uint32_t i = 0; i < nrem; ++i) /* :202-210: no point above -> cut stays 0 (Q4) */
            if (rem[i].key > z_max)
            {
                n_seed = i;
                break;
            }

        /* fit_ground_plane, :219-309 */
        memset(is_ground, 0, ns);
        for (uint32_t i = 0; i < n_seed; ++i)
        {
            is_ground[rem[i].idx] = 1;
            zord_ground[i] = rem[i].idx;
        }
        uint32_t n_g = n_seed;
        int all_obstacle = 0;
        int tested = 0; /* has the inlier test run at least once? */
        float plane[4] = {0, 0, 0, 0};
        for (uint32_t it = 0; it < cfg->number_of_iterations; ++it)
        {
            if (n_g < 3) /* :251-259 */
            {
                all_obstacle = 1;
                break;
            }
            moments m;
            memset(&m, 0, sizeof m);
            for (uint32_t k = 0; k < ns; ++k)
                if (is_ground[k] && mom_add(&m, pt_at(pts, stride, seg[k].idx)))
                {
                    rc = ORC_ERR_RANGE;
                    break;
                }
            if (rc != ORC_OK)
                break;
            if (plane_from_moments(&m, plane)) /* :275-283 */
            {
                all_obstacle = 1;
                break;
            }
            const float a = plane[0], b = plane[1], c = plane[2], d = plane[3];
            const float thr = cfg->orthogonal_distance_threshold * sqrtf((a * a + b * b) + c * c); /* :293 */
            n_g = 0;
            for (uint32_t k = 0; k < ns; ++k)
            {
                const float *p = pt_at(pts, stride, seg[k].idx);
                const float dist = ((p[0] * a + p[1] * b) + p[2] * c) - d; /* :287-291 */
                is_ground[k] = dist < thr;                                  /* :299 signed (Q1) */
                n_g += is_ground[k];
            }
            tested = 1;
        }
        if (rc != ORC_OK)
            break;
        if (planes)
            memcpy(planes + 4 * (size_t)s, plane, sizeof plane);
        if (all_obstacle)
        {
            if (seg_status)
                seg_status[s] = ORC_SEG_ALL_OBSTACLE;
            for (uint32_t k = 0; k < ns; ++k)
            {
                labels[seg[k].idx] = ORC_LABEL_OBSTACLE;
                obstacle_idx[no++] = seg[k].idx;
            }
            continue;
        }
        if (seg_status)
            seg_status[s] = ORC_SEG_OK;
        if (!tested)
        {
            /* number_of_iterations == 0: ground = seeds in z order, no obstacles (:243-247) */
            for (uint32_t i = 0; i < n_seed; ++i)
            {
                labels[seg[zord_ground[i]].idx] = ORC_LABEL_GROUND;
                ground_idx[ng++] = seg[zord_ground[i]].idx;
            }
            continue;
        }
        /* :331-343: ground then obstacle, ascending in-segment order (Q7) */
        for (uint32_t k = 0; k < ns; ++k)
            if (is_ground[k])
            {
                labels[seg[k].idx] = ORC_LABEL_GROUND;
                ground_idx[ng++] = seg[k].idx;
            }
        for (uint32_t k = 0; k < ns; ++k)
            if (!is_ground[k])
            {
                labels[seg[k].idx] = ORC_LABEL_OBSTACLE;
                obstacle_idx[no++] = seg[k].idx;
            }
    }
    free(zord_ground);
    free(is_ground);
    free(sz);
    free(sx);
    *n_ground = ng;
    *n_obstacle = no;
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* std::nth_element of libstdc++ 11 restated (bits/stl_algo.h:79-97,1642-1651,1819-1838,       */
/* 1878-1907,1964-1986,4809-4812; bits/stl_heap.h:131-147,220-262,340-360): third-party        */
/* algorithm the reference relies on at src/kdtree.hpp:205-206 (GCC 11.4 libstdc++, the        */
/* toolchain of this image).  Elements are kd nodes compared on one axis.                      */
/* ------------------------------------------------------------------------------------------ */

typedef struct
{
    float p[3];
    uint32_t idx;
} kdnode;

#define LESS(a, b) ((a).p[axis] < (b).p[axis])
#define SWAP(a, b)                                                                                                   \
    do                                                                                                               \
    {                                                                                                                \
        kdnode t_ = (a);                                                                                             \
        (a) = (b);                                                                                                   \
        (b) = t_;                                                                                                    \
    } while (0)

static void push_heap_(kdnode *f, long hole, long top, kdnode value, int axis)
{
    long parent = (hole - 1) / 2;
    while (hole > top && LESS(f[parent], value))
    {
        f[hole] = f[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    f[hole] = value;
}

static void adjust_heap_(kdnode *f, long hole, long len, kdnode value, int axis)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2)
    {
        child = 2 * (child + 1);
        if (LESS(f[child], f[child - 1]))
            child--;
        f[hole] = f[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2)
    {
        child = 2 * (child + 1);
        f[hole] = f[child - 1];
        hole = child - 1;
    }
    push_heap_(f, hole, top, value, axis);
}

static void heap_select_(kdnode *a, long first, long middle, long last, int axis)
{
    kdnode *f = a + first;
    const long len = middle - first;
    if (len >= 2) /* __make_heap */
    {
        long parent = (len - 2) / 2;
        for (;;)
        {
            kdnode value = f[parent];
            adjust_heap_(f, parent, len, value, axis);
            if (parent == 0)
                break;
            parent--;
        }
    }
    for (long i = middle; i < last; ++i)
        if (LESS(a[i], a[first]))
        {
            /* __pop_heap(first, middle, i) */
            kdnode value = a[i];
            a[i] = a[first];
            adjust_heap_(f, 0, len, value, axis);
        }
}

static void insertion_sort_(kdnode *a, long first, long last, int axis)
{
    if (first == last)
        return;
    for (long i = first + 1; i != last; ++i)
    {
        if (LESS(a[i], a[first]))
        {
            kdnode val = a[i];
            memmove(a + first + 1, a + first, sizeof(kdnode) * (size_t)(i - first));
            a[first] = val;
        }
        else
        {
            kdnode val = a[i];
            long l = i, nx = i - 1;
            while (LESS(val, a[nx]))
            {
                a[l] = a[nx];
                l = nx;
                --nx;
            }
            a[l] = val;
        }
    }
}

static long unguarded_partition_pivot_(kdnode *a, long first, long last, int axis)
{
    const long mid = first + (last - first) / 2;
    /* __move_median_to_first(first, first+1, mid, last-1) */
    const long A = first + 1, B = mid, C = last - 1;
    if (LESS(a[A], a[B]))
    {
        if (LESS(a[B], a[C]))
            SWAP(a[first], a[B]);
        else if (LESS(a[A], a[C]))
            SWAP(a[first], a[C]);
        else
            SWAP(a[first], a[A]);
    }
    else if (LESS(a[A], a[C]))
        SWAP(a[first], a[A]);
    else if (LESS(a[B], a[C]))
        SWAP(a[first], a[C]);
    else
        SWAP(a[first], a[B]);
    /* __unguarded_partition(first+1, last, pivot=first) */
    long f = first + 1, l = last;
    for (;;)
    {
        while (LESS(a[f], a[first]))
            ++f;
        --l;
        while (LESS(a[first], a[l]))
            --l;
        if (!(f < l))
            return f;
        SWAP(a[f], a[l]);
        ++f;
    }
}

static void nth_element_(kdnode *a, long first, long nth, long last, int axis)
{
    if (first == last || nth == last)
        return;
    long n = last - first;
    int lg = 0;
    while (n > 1)
    {
        n >>= 1;
        ++lg;
    }
    long depth_limit = 2L * lg;
    while (last - first > 3)
    {
        if (depth_limit == 0)
        {
            heap_select_(a, first, nth + 1, last, axis);
            SWAP(a[first], a[nth]);
            return;
        }
        --depth_limit;
        const long cut = unguarded_partition_pivot_(a, first, last, axis);
        if (cut <= nth)
            first = cut;
        else
            last = cut;
    }
    insertion_sort_(a, first, last, axis);
}

void orc_nth_element_u32(float *keys, uint32_t *payload, uint32_t first, uint32_t nth, uint32_t last)
{
    const uint32_t n = last - first;
    if (n == 0)
        return;
    kdnode *a = (kdnode *)malloc(sizeof(kdnode) * n);
    for (uint32_t i = 0; i < n; ++i)
    {
        a[i].p[0] = keys[first + i];
        a[i].p[1] = a[i].p[2] = 0.0f;
        a[i].idx = payload[first + i];
    }
    nth_element_(a, 0, (long)(nth - first), (long)n, 0);
    for (uint32_t i = 0; i < n; ++i)
    {
        keys[first + i] = a[i].p[0];
        payload[first + i] = a[i].idx;
    }
    free(a);
}

/* ------------------------------------------------------------------------------------------ */
/* KDTree<float,3>: rebuild (src/kdtree.hpp:174-225), radius_search (:292-341), dist_sqr       */
/* (:145-163).  The pointer tree is implicit: a node range [b,e) has its node at               */
/* mid = b + (e-b)/2, left child [b,mid), right child [mid+1,e) (:203,:210-218).               */
/* ------------------------------------------------------------------------------------------ */

typedef struct
{
    uint32_t b, e, depth;
} kdrange;

static kdnode *kd_build(const float *xyz, uint32_t m)
{
    kdnode *nodes = (kdnode *)malloc(sizeof(kdnode) * (m ? m : 1));
    for (uint32_t i = 0; i < m; ++i)
    {
        nodes[i].p[0] = xyz[3 * (size_t)i];
        nodes[i].p[1] = xyz[3 * (size_t)i + 1];
        nodes[i].p[2] = xyz[3 * (size_t)i + 2];
        nodes[i].idx = i;
    }
    kdrange *stack = (kdrange *)malloc(sizeof(kdrange) * 128);
    uint32_t sp = 0;
    stack[sp++] = (kdrange){0, m, 0};
    while (sp)
    {
        const kdrange r = stack[--sp];
        if (r.b >= r.e)
            continue;
        const int axis = (int)(r.depth % 3);
        const uint32_t mid = r.b + (r.e - r.b) / 2;
        nth_element_(nodes, r.b, mid, r.e, axis);
        if (mid > r.b)
            stack[sp++] = (kdrange){r.b, mid, r.depth + 1};
        if (mid + 1 < r.e)
            stack[sp++] = (kdrange){mid + 1, r.e, r.depth + 1};
    }
    free(stack);
    return nodes;
}

/* src/kdtree.hpp:145-157: (a0-b0)^2 + ((a1-b1)^2 + ((a2-b2)^2 + 0)) */
static inline float dist_sqr(const float *a, const float *b)
{
    const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    return d0 * d0 + (d1 * d1 + (d2 * d2 + 0.0f));
}

typedef struct
{
    uint32_t idx;
    float dist;
} neigh_t;

typedef struct
{
    uint32_t b, e, axis;
} rs_item;

/* out must hold m entries; returns count */
static uint32_t kd_radius_search(const kdnode *nodes, uint32_t m, const float *target, float r2, neigh_t *out,
                                 rs_item *stack)
{
    uint32_t cnt = 0, sp = 0;
    stack[sp++] = (rs_item){0, m, 0};
    while (sp)
    {
        const rs_item it = stack[--sp];
        if (it.b >= it.e) /* nullptr child */
            continue;
        const uint32_t mid = it.b + (it.e - it.b) / 2;
        const kdnode *node = nodes + mid;
        const float dist = dist_sqr(target, node->p);
        if (dist <= r2) /* :315 inclusive */
        {
            out[cnt].idx = node->idx;
            out[cnt].dist = dist;
            ++cnt;
        }
        const uint32_t next = (it.axis + 1) % 3;
        const float delta = node->p[it.axis] - target[it.axis];
        const float abs_delta_sqr = delta * delta;
        if (abs_delta_sqr <= r2)
        {
            stack[sp++] = (rs_item){mid + 1, it.e, next}; /* right pushed first ... */
            stack[sp++] = (rs_item){it.b, mid, next};     /* ... so left is visited first */
        }
        else if (delta > 0)
            stack[sp++] = (rs_item){it.b, mid, next};
        else
            stack[sp++] = (rs_item){mid + 1, it.e, next};
    }
    return cnt;
}

int orc_kd_layout(const float *xyz, uint32_t m, uint32_t *layout_idx)
{
    kdnode *nodes = kd_build(xyz, m);
    for (uint32_t i = 0; i < m; ++i)
        layout_idx[i] = nodes[i].idx;
    free(nodes);
    return ORC_OK;
}

int orc_kd_preorder(const float *xyz, uint32_t m, uint32_t *preorder_idx)
{
    kdnode *nodes = kd_build(xyz, m);
    neigh_t *out = (neigh_t *)malloc(sizeof(neigh_t) * (m ? m : 1));
    rs_item *stack = (rs_item *)malloc(sizeof(rs_item) * 256);
    const float target[3] = {0, 0, 0};
    const uint32_t cnt = m ? kd_radius_search(nodes, m, target, INFINITY, out, stack) : 0;
    for (uint32_t i = 0; i < cnt; ++i)
        preorder_idx[i] = out[i].idx;
    free(stack);
    free(out);
    free(nodes);
    return cnt == m ? ORC_OK : ORC_ERR_ARG;
}

int orc_radius_search(const float *xyz, uint32_t m, const float *target, float r2, uint32_t *out_idx,
                      float *out_dist, uint32_t *count)
{
    kdnode *nodes = kd_build(xyz, m);
    neigh_t *out = (neigh_t *)malloc(sizeof(neigh_t) * (m ? m : 1));
    rs_item *stack = (rs_item *)malloc(sizeof(rs_item) * 256);
    const uint32_t cnt = m ? kd_radius_search(nodes, m, target, r2, out, stack) : 0;
    for (uint32_t i = 0; i < cnt; ++i)
    {
        out_idx[i] = out[i].idx;
        out_dist[i] = out[i].dist;
    }
    *count = cnt;
    free(stack);
    free(out);
    free(nodes);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Clusterer::cluster, src/clustering.cpp:47-125                                               */
/* ------------------------------------------------------------------------------------------ */

static uint8_t *g_trace_expanded = NULL; /* analysis only: which points had radius_search called on them */

int orc_cluster_stats(const void *pts, size_t stride, uint32_t m, const orc_clu_cfg *cfg, int32_t *labels,
                      uint32_t *n_clusters, uint64_t *n_expansions, uint64_t *n_visits)
{
    uint64_t expansions = 0, visits = 0;
    if (n_clusters)
        *n_clusters = 0;
    if (m == 0) /* :51-54 */
        return ORC_OK;
    for (uint32_t i = 0; i < m; ++i)
        labels[i] = ORC_CLUSTER_UNDEFINED; /* :50 */

    float *xyz = (float *)malloc(sizeof(float) * 3 * (size_t)m);
    for (uint32_t i = 0; i < m; ++i)
    {
        const float *p = pt_at(pts, stride, i);
        xyz[3 * (size_t)i] = p[0];
        xyz[3 * (size_t)i + 1] = p[1];
        xyz[3 * (size_t)i + 2] = p[2];
    }
    kdnode *nodes = kd_build(xyz, m); /* :63 */
    uint8_t *removed = (uint8_t *)calloc(m, 1);
    neigh_t *neigh = (neigh_t *)malloc(sizeof(neigh_t) * m);
    rs_item *stack = (rs_item *)malloc(sizeof(rs_item) * 256);
    size_t qcap = 1024, icap = 1024;
    uint32_t *queue = (uint32_t *)malloc(sizeof(uint32_t) * qcap);
    uint32_t *indices = (uint32_t *)malloc(sizeof(uint32_t) * icap);

    /* :66-67 double: pow(1.0 - quality, 2) * distance_squared */
    const double thr = pow(1.0 - (double)cfg->cluster_quality, 2) * (double)cfg->distance_squared;

    int32_t label = 0;
    for (uint32_t i = 0; i < m; ++i) /* :70 */
    {
        if (removed[i])
            continue;
        size_t qh = 0, qt = 0, ni = 0;
        queue[qt++] = i;
        while (qh < qt) /* :80 FIFO */
        {
            const uint32_t j = queue[qh++];
            if (removed[j])
                continue;
            const uint32_t cnt = kd_radius_search(nodes, m, xyz + 3 * (size_t)j, cfg->distance_squared, neigh, stack);
            ++expansions;
            if (g_trace_expanded)
                g_trace_expanded[j] = 1;
            for (uint32_t t = 0; t < cnt; ++t)
            {
                const uint32_t k = neigh[t].idx;
                if (removed[k])
                    continue;
                ++visits;
                labels[k] = label;
                if (ni == icap)
                    indices = (uint32_t *)realloc(indices, sizeof(uint32_t) * (icap *= 2));
                indices[ni++] = k; /* :100 duplicates kept (Q8) */
                if ((double)neigh[t].dist <= thr)
                    removed[k] = 1; /* :102-105 */
                else
                {
                    if (qt == qcap)
                        queue = (uint32_t *)realloc(queue, sizeof(uint32_t) * (qcap *= 2));
                    queue[qt++] = k;
                }
            }
        }
        if (ni < cfg->min_cluster_size || ni > cfg->max_cluster_size) /* :113 */
        {
            for (size_t t = 0; t < ni; ++t)
                labels[indices[t]] = ORC_CLUSTER_INVALID;
        }
        else
            ++label;
    }
    if (n_clusters)
        *n_clusters = (uint32_t)label;
    if (n_expansions)
        *n_expansions = expansions;
    if (n_visits)
        *n_visits = visits;
    free(indices);
    free(queue);
    free(stack);
    free(neigh);
    free(removed);
    free(nodes);
    free(xyz);
    return ORC_OK;
}

/* analysis helper (tools/): expanded[i] = 1 iff the loop called radius_search on point i */
int orc_cluster_trace(const void *pts, size_t stride, uint32_t m, const orc_clu_cfg *cfg, int32_t *labels,
                      uint32_t *n_clusters, uint8_t *expanded)
{
    memset(expanded, 0, m);
    g_trace_expanded = expanded;
    const int rc = orc_cluster_stats(pts, stride, m, cfg, labels, n_clusters, NULL, NULL);
    g_trace_expanded = NULL;
    return rc;
}

int orc_cluster(const void *pts, size_t stride, uint32_t m, const orc_clu_cfg *cfg, int32_t *labels,
                uint32_t *n_clusters)
{
    return orc_cluster_stats(pts, stride, m, cfg, labels, n_clusters, NULL, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* N3: 2-D convex hull of a cluster (the convex branch of findOrderedConcaveOutlines,           */
/* src/polygon_simplification.cpp:96-115, and findOrderedConvexOutlines :32-80).                */
/*                                                                                              */
/* PARITY UNPINNED: the reference calls geom::constructConvexHull(points, ANDREW_MONOTONE_CHAIN, */
/* COUNTERCLOCKWISE) from the git submodule Convex-Hull (github.com/YevgeniyEngineer/Convex-Hull, */
/* .gitmodules:1-3), which is NOT vendored in the reference checkout (empty directory, no pinned */
/* commit available here).  What follows is the published algorithm (A. M. Andrew, "Another      */
/* efficient algorithm for convex hulls in two dimensions", 1979) with these conventions:        */
/*   - points (x, y) = first two coordinates as float32 (:104-107), sorted by (x, y, index);     */
/*   - exact duplicates of the previous sorted point are skipped;                                */
/*   - lower hull then upper hull, a point is popped while cross(h[-2], h[-1], p) <= 0, so       */
/*     collinear boundary points are NOT hull vertices; cross is evaluated in float32,           */
/*     (bx-ax)*(cy-ay) - (by-ay)*(cx-ax), no contraction;                                        */
/*   - result counter-clockwise starting at the lowest (x, y) point, no repeated closing vertex; */
/*     one distinct point -> 1 vertex, two -> 2 vertices.                                        */
/* out_idx receives indices into the input array.                                               */
/* ------------------------------------------------------------------------------------------ */

typedef struct
{
    float x, y;
    uint32_t idx;
} hpt;

static int cmp_hpt(const void *a, const void *b)
{
    const hpt *p = (const hpt *)a, *q = (const hpt *)b;
    if (p->x != q->x)
        return p->x < q->x ? -1 : 1;
    if (p->y != q->y)
        return p->y < q->y ? -1 : 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx);
}

static inline float hcross(const hpt *a, const hpt *b, const hpt *c)
{
    const float l = (b->x - a->x) * (c->y - a->y);
    const float r = (b->y - a->y) * (c->x - a->x);
    return l - r;
}

int orc_convex_hull(const float *xy, uint32_t n, uint32_t *out_idx, uint32_t *count)
{
    *count = 0;
    if (n == 0)
        return ORC_OK;
    hpt *p = (hpt *)malloc(sizeof(hpt) * n);
    for (uint32_t i = 0; i < n; ++i)
    {
        p[i].x = xy[2 * (size_t)i] + 0.0f; /* -0 -> +0 so that the order is the value order */
        p[i].y = xy[2 * (size_t)i + 1] + 0.0f;
        p[i].idx = i;
    }
    qsort(p, n, sizeof(hpt), cmp_hpt);
    uint32_t u = 0; /* distinct points */
    for (uint32_t i = 0; i < n; ++i)
        if (u == 0 || p[i].x != p[u - 1].x || p[i].y != p[u - 1].y)
            p[u++] = p[i];
    if (u <= 2)
    {
        for (uint32_t i = 0; i < u; ++i)
            out_idx[i] = p[i].idx;
        *count = u;
        free(p);
        return ORC_OK;
    }
    hpt *h = (hpt *)malloc(sizeof(hpt) * 2 * (size_t)u);
    uint32_t k = 0;
    for (uint32_t i = 0; i < u; ++i) /* lower hull */
    {
        while (k >= 2 && hcross(&h[k - 2], &h[k - 1], &p[i]) <= 0.0f)
            --k;
        h[k++] = p[i];
    }
    const uint32_t lower = k + 1;
    for (uint32_t i = u - 1; i-- > 0;) /* upper hull */
    {
        while (k >= lower && hcross(&h[k - 2], &h[k - 1], &p[i]) <= 0.0f)
            --k;
        h[k++] = p[i];
    }
    --k; /* the last point repeats the first */
    for (uint32_t i = 0; i < k; ++i)
        out_idx[i] = h[i].idx;
    *count = k;
    free(h);
    free(p);
    return ORC_OK;
}

/* Hulls of every valid cluster with fewer than max_points points (20 in the reference, :97), clusters in label
 * order: hull_offsets[n_clusters + 1], hull_indices = indices into the clustered cloud.  Larger clusters get an
 * empty hull here (the reference sends them to the concave-hull submodule, absent -- out of scope). */
int orc_cluster_hulls(const void *pts, size_t stride, uint32_t m, const int32_t *labels, uint32_t n_clusters,
                      uint32_t max_points, uint32_t *hull_offsets, uint32_t *hull_indices)
{
    uint32_t *cnt = (uint32_t *)calloc((size_t)n_clusters + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < m; ++i)
        if (labels[i] >= 0 && (uint32_t)labels[i] < n_clusters)
            ++cnt[labels[i] + 1];
    for (uint32_t c = 0; c < n_clusters; ++c)
        cnt[c + 1] += cnt[c];
    uint32_t *members = (uint32_t *)malloc(sizeof(uint32_t) * (m ? m : 1));
    uint32_t *fill = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n_clusters + 1));
    memcpy(fill, cnt, sizeof(uint32_t) * ((size_t)n_clusters + 1));
    for (uint32_t i = 0; i < m; ++i) /* src/processor.cpp:183-197: members in index order */
        if (labels[i] >= 0 && (uint32_t)labels[i] < n_clusters)
            members[fill[labels[i]]++] = i;
    uint32_t total = 0;
    float *xy = (float *)malloc(sizeof(float) * 2 * (m ? m : 1));
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (m ? m : 1));
    for (uint32_t c = 0; c < n_clusters; ++c)
    {
        hull_offsets[c] = total;
        const uint32_t b = cnt[c], n = cnt[c + 1] - cnt[c];
        if (n == 0 || n >= max_points)
            continue;
        for (uint32_t t = 0; t < n; ++t)
        {
            const float *p = pt_at(pts, stride, members[b + t]);
            xy[2 * t] = p[0];
            xy[2 * t + 1] = p[1];
        }
        uint32_t k = 0;
        orc_convex_hull(xy, n, tmp, &k);
        for (uint32_t t = 0; t < k; ++t)
            hull_indices[total + t] = members[b + tmp[t]];
        total += k;
    }
    hull_offsets[n_clusters] = total;
    free(tmp);
    free(xy);
    free(fill);
    free(members);
    free(cnt);
    return ORC_OK;
}
