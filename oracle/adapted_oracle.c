/*
 * adapted_oracle.c -- CPU ORACLE for the `adapted detect` hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this.  The product (adapted_amd/) never does.
 *
 * A plain-C restatement of the reference's per-minibatch segmentation path
 * (KleistLab/ADAPTed v0.2.4; citations are file:line under /root/reference):
 *
 *   N1 normalize_signal / med_mad / clip_signal     adapted/detect/normalize.py:15-63
 *   D1 downscale_signal                             adapted/detect/downscale.py:4-41
 *   G1/G2 c_llr_trace, c_llr_trace_gains, _gains    adapted/detect/_c_llr.pyx:23-37,67-88,176-236
 *   T1 LLRTrace._trace_start_end                    adapted/detect/llr.py:135-142
 *   P1 find_peaks_in_trace                          adapted/detect/llr.py:204-224
 *   P2 correct_for_plateau                          adapted/detect/llr.py:145-177
 *   P3 correct_for_split_peak                       adapted/detect/llr.py:180-201
 *   P4 detect_full_polya_trace_peak_with_spike      adapted/detect/llr.py:406-479
 *   V1 validate_boundaries                          adapted/detect/combined.py:358-631
 *   V2 find_open_pores                              adapted/detect/anomalies.py:15-35
 *   V3 real_range_check                             adapted/detect/real_range.py:33-63
 *   V4 mean_var_shift_polyA_check                   adapted/detect/mvs.py:45-158
 *   S1 calc_partition_stats                         adapted/partition/signal_partitions.py:81-96
 *   K1 detect_rna_start_peak                        adapted/detect/start_peak.py:7-119
 *   drivers combined_detect_llr2 / _start_peak      adapted/detect/combined.py:122-227,312-355
 *
 * Third-party algorithms the reference calls and that are NOT in /root/reference are
 * restated from their published behaviour and pinned by tests against the installed
 * libraries and the golden vectors (tests/golden/, made by oracle/gen_golden.py running
 * the real reference in the build container):
 *   numpy (unpinned in reference setup.py:35-42; environment.yml numpy=1.24.4):
 *       float32/float64 add.reduce order (8192-element chunks, 8-accumulator pairwise
 *       blocks of <=128), mean/var/std/median/nanmedian/percentile(linear) dtype rules
 *   scipy.signal.find_peaks (+ _local_maxima_1d, _select_by_peak_distance,
 *       _peak_prominences, _peak_widths), scipy.stats.linregress r-value (scipy unpinned)
 *   bottleneck.move_mean / move_var float32 streaming recurrences (bottleneck unpinned)
 *   libm log(): the reference's Cython module calls glibc log; so does this file.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC  (no -march flags: the
 * reference's own extension is built for generic x86-64, i.e. without FMA contraction).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ config / rows */

typedef struct {
    /* core */
    int32_t min_obs_adapter, max_obs_adapter, min_obs_polya, downscale_factor, max_obs_trace;
    double sig_norm_outlier_thresh;
    /* llr_boundaries */
    double adapter_peak_prominence, adapter_peak_rel_height;
    int32_t adapter_peak_width, _pad0;
    /* mvs_polya */
    int32_t mvs_detect_check, mvs_detect_overwrite, search_window, pA_mean_window;
    int32_t pA_var_window, median_shift_window, polyA_window, _pad1;
    double pA_mean_range[2], pA_var_range[2], median_shift_range[2];
    double polyA_med_range[2], polyA_local_range[2], pA_mean_adapter_med_scale_range[2];
    /* real_range */
    int32_t detect_open_pores, real_signal_check, mean_window, max_obs_local_range;
    double mean_start_range[2], mean_end_range[2], local_range[2], adapter_mad_range[2];
    /* med_shift */
    int32_t detect_med_shift, med_shift_window;
    double med_shift_range[2];
    /* rna_start_peak */
    int32_t sp_downscale_factor, start_peak_max_idx, sp_offset1, sp_offset2;
    double open_pore_pa;
    /* cnn_boundaries */
    int32_t polya_cand_k, fallback_to_llr_short_reads;
    /* which primary method's columns receive the primary boundaries: 0 llr, 1 cnn, 2 start_peak */
    int32_t primary_method, _pad2;
} orc_cfg;

enum { /* numeric columns, DetectResults field order (adapted/container_types.py:23-92) */
    C_SIGNAL_LEN, C_PRELOADED,
    C_ADAPTER_START, C_ADAPTER_END, C_ADAPTER_LEN, C_ADAPTER_MEAN, C_ADAPTER_STD, C_ADAPTER_MED, C_ADAPTER_MAD,
    C_POLYA_START, C_POLYA_END, C_POLYA_LEN, C_POLYA_MEAN, C_POLYA_STD, C_POLYA_MED, C_POLYA_MAD,
    C_RNA_START, C_RNA_LEN, C_RNA_MEAN, C_RNA_STD, C_RNA_MED, C_RNA_MAD,
    C_SP_IDX, C_SP_PA, C_SP_NEXT_IDX, C_SP_NEXT_PA, C_SP_OPEN_PORE_IDX,
    C_MED_SHIFT, C_PRIMARY_ADAPTER_END, C_PRIMARY_POLYA_END,
    C_MVS_MEAN, C_MVS_VAR, C_MVS_POLYA_MED, C_MVS_LOCAL_RANGE, C_MVS_MED_SHIFT,
    C_REAL_MEAN_START, C_REAL_MEAN_END, C_REAL_LOCAL_RANGE,
    C_MVS_ADAPTER_END,
    ORC_NCOL
};

enum { /* fail codes <-> the reference's fail_reason strings (combined.py:396-580) */
    F_NONE = 0,
    F_NO_ADAPTER = 1,        /* "No adapter detected (primary)" */
    F_ADAPTER_MAD = 2,       /* "adapter MAD check failed" */
    F_OPEN_PORE = 3,         /* "Open pore too close to boundary" */
    F_REAL_RANGE = 4,        /* "Real signal check failed" */
    F_NO_POLYA = 5,          /* "No polya detected (primary)" */
    F_MVS_NOT_ENOUGH = 6,    /* "MVS polya check failed: not enough signal" */
    F_MVS_CHECKS = 7,        /* "MVS polya check failed: <mean var med range shift>" */
    F_MED_SHIFT = 8,         /* "Median shift check failed" */
    F_EXC_TOPK_NONE = 9,     /* TypeError: 'NoneType' object is not iterable (combined.py:464) */
    F_EXC_SLICE = 10,        /* TypeError: slice indices must be integers ... (start-peak None rows) */
    F_EXC_MOVE_WINDOW = 11,  /* ValueError from bottleneck: window > n */
    F_EXC_PA_RANGE = 12,     /* ValueError("pA_mean_range is not specified") */
    /* 13, 14: CNN fallback exceptions (see orc_cnn_fallback); 9..14 are the exception rows */
    F_NO_ADAPTER_MVS = 15    /* "No adapter detected in range (mvs_detect)" (combined.py:540, mvs_detect_overwrite) */
};
#define MVS_FLAG_TO_EARLY_STOP 256 /* mvs_fail_mask bit 8: mvs_llr_polya_end_to_early_stop (combined.py:559-561) */

#define ORC_MAX_CAND 16
#define ORC_MAX_OPEN_PORES 4096 /* (test infrastructure: rows may be large; the reference's list is unbounded) */

typedef struct {
    double col[ORC_NCOL];
    uint64_t present;      /* bit c set <=> col[c] is not None */
    int32_t success;
    int32_t fail_code;
    int32_t mvs_fail_mask; /* bit0 mean, bit1 var, bit2 med, bit3 range, bit4 shift FAILED; bit8: polya_end set to the early-stop position */
    int32_t start_peak_type; /* 0 None, 1 "open pore in adapter", 2 "potential concatemer adapter-only read" */
    int32_t n_cand;        /* -1 <=> polya_candidates is None */
    int32_t n_open_pores;  /* -1 <=> open_pores is None; may exceed ORC_MAX_OPEN_PORES (list truncated) */
    int64_t cand[ORC_MAX_CAND];
    int32_t open_pores[ORC_MAX_OPEN_PORES];
} orc_row;

int orc_sizeof_cfg(void) { return (int)sizeof(orc_cfg); }
int orc_sizeof_row(void) { return (int)sizeof(orc_row); }
int orc_ncol(void) { return ORC_NCOL; }

static void row_set(orc_row *r, int c, double v) { r->col[c] = v; r->present |= (1ull << c); }

/* ------------------------------------------------------------------ numpy sums */

#define PW_BLOCK 128
#define NP_BUFSIZE 8192

static float pw_f32(const float *a, long n)
{
    if (n < 8) {
        float res = 0.f;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= PW_BLOCK) {
        float r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pw_f32(a, n2) + pw_f32(a + n2, n - n2);
    }
}

static double pw_f64(const double *a, long n)
{
    if (n < 8) {
        double res = 0.;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= PW_BLOCK) {
        double r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pw_f64(a, n2) + pw_f64(a + n2, n - n2);
    }
}

/* np.add.reduce over a contiguous 1-D array: the inner loop sees <= 8192 elements at a time */
float orc_np_sum_f32(const float *a, long n)
{
    float t = 0.f;
    for (long s = 0; s < n; s += NP_BUFSIZE) {
        long c = n - s < NP_BUFSIZE ? n - s : NP_BUFSIZE;
        t += pw_f32(a + s, c);
    }
    return t;
}

double orc_np_sum_f64(const double *a, long n)
{
    double t = 0.;
    for (long s = 0; s < n; s += NP_BUFSIZE) {
        long c = n - s < NP_BUFSIZE ? n - s : NP_BUFSIZE;
        t += pw_f64(a + s, c);
    }
    return t;
}

float orc_np_mean_f32(const float *a, long n) { return orc_np_sum_f32(a, n) / (float)n; }

/* np.var(float32): mean in f32, (x-mean)^2 in f32, sum / n  (numpy/_core/_methods.py _var) */
float orc_np_var_f32(const float *a, long n)
{
    float mu = orc_np_mean_f32(a, n);
    float *t = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
    for (long i = 0; i < n; i++) { float d = a[i] - mu; t[i] = d * d; }
    float v = orc_np_sum_f32(t, n) / (float)n;
    free(t);
    return v;
}
float orc_np_std_f32(const float *a, long n) { return sqrtf(orc_np_var_f32(a, n)); }

/* np.nanstd(float64 1-D) (numpy/lib/_nanfunctions_impl.py _nanvar): NaNs -> 0, mean over
 * the non-NaN count, squared deviations with the NaN slots zeroed again */
double orc_np_nanstd_f64(const double *a, long n)
{
    if (n == 0) return NAN;
    double *t = (double *)malloc(sizeof(double) * n);
    long cnt = 0;
    for (long i = 0; i < n; i++) { if (a[i] == a[i]) { t[i] = a[i]; cnt++; } else t[i] = 0.; }
    double avg = orc_np_sum_f64(t, n) / (double)cnt;
    for (long i = 0; i < n; i++) { if (a[i] == a[i]) { double d = a[i] - avg; t[i] = d * d; } else t[i] = 0.; }
    double var = orc_np_sum_f64(t, n) / (double)cnt;
    free(t);
    if (cnt == 0) return NAN;
    return sqrt(var);
}

/* ------------------------------------------------------------------ selection */

static void swapf(float *a, float *b) { float t = *a; *a = *b; *b = t; }

/* k-th smallest (0-based) in place; afterwards a[k] is in sorted position, a[<k] <= a[k] <= a[>k] */
static void nth_element_f32(float *a, long n, long k)
{
    long lo = 0, hi = n - 1;
    while (lo < hi) {
        long mid = lo + (hi - lo) / 2;
        if (a[mid] < a[lo]) swapf(&a[mid], &a[lo]);
        if (a[hi] < a[lo]) swapf(&a[hi], &a[lo]);
        if (a[hi] < a[mid]) swapf(&a[hi], &a[mid]);
        float p = a[mid];
        long i = lo, j = hi;
        while (i <= j) {
            while (a[i] < p) i++;
            while (p < a[j]) j--;
            if (i <= j) { swapf(&a[i], &a[j]); i++; j--; }
        }
        if (k <= j) hi = j;
        else if (k >= i) lo = i;
        else return;
    }
}

/* np.median of a NaN-free float32 buffer (DESTROYS buf): odd -> middle; even -> f32 (a+b)/2 */
static float median_inplace_f32(float *buf, long n)
{
    if (n == 0) return NAN;
    long k = n / 2;
    nth_element_f32(buf, n, k);
    float hi = buf[k];
    if (n & 1) return hi;
    float lo = buf[0];
    for (long i = 1; i < k; i++) if (buf[i] > lo) lo = buf[i];
    return (lo + hi) / 2.0f;
}

/* np.median(x[0:n]) for float32 (NaN anywhere -> NaN, numpy _median_nancheck) */
float orc_np_median_f32(const float *x, long n)
{
    if (n <= 0) return NAN;
    float *b = (float *)malloc(sizeof(float) * n);
    for (long i = 0; i < n; i++) { if (x[i] != x[i]) { free(b); return NAN; } b[i] = x[i]; }
    float m = median_inplace_f32(b, n);
    free(b);
    return m;
}

/* np.median(|x - c|) in float32 */
float orc_np_mad_f32(const float *x, long n, float c)
{
    if (n <= 0) return NAN;
    float *b = (float *)malloc(sizeof(float) * n);
    for (long i = 0; i < n; i++) { float d = fabsf(x[i] - c); if (d != d) { free(b); return NAN; } b[i] = d; }
    float m = median_inplace_f32(b, n);
    free(b);
    return m;
}

/* np.nanmedian over a flat float32 buffer; *n_valid = number of non-NaN */
float orc_np_nanmedian_f32(const float *x, long n, long *n_valid)
{
    float *b = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
    long c = 0;
    for (long i = 0; i < n; i++) if (x[i] == x[i]) b[c++] = x[i];
    if (n_valid) *n_valid = c;
    float m = median_inplace_f32(b, c);
    free(b);
    return m;
}

static float kth_copy_f32(const float *x, long n, long k, float *scratch)
{
    memcpy(scratch, x, sizeof(float) * n);
    nth_element_f32(scratch, n, k);
    return scratch[k];
}

/* np.percentile(x, (qa, qb)) with the default linear method, then np.subtract(*res).
 * numpy/lib/_function_base_impl.py: virtual index (n-1)*q in f64, gamma f64,
 * _lerp: diff = b - a in the ARRAY dtype (f32), a + diff*g (f64), and for g >= 0.5
 * b - diff*(1-g). */
static double np_percentile_f32(const float *x, long n, double q100, float *scratch)
{
    double q = q100 / 100.0;
    double vi = (double)(n - 1) * q;
    long lo = (long)floor(vi);
    if (lo < 0) lo = 0;
    if (lo > n - 1) lo = n - 1;
    long hi = lo + 1 > n - 1 ? n - 1 : lo + 1;
    double g = vi - (double)lo;
    float a = kth_copy_f32(x, n, lo, scratch);
    float b = kth_copy_f32(x, n, hi, scratch);
    float diff = b - a;
    double r = (double)a + (double)diff * g;
    if (g >= 0.5) r = (double)b - (double)diff * (1.0 - g);
    return r;
}

double orc_np_percentile_diff_f32(const float *x, long n, double qa, double qb)
{
    if (n <= 0) return NAN;
    for (long i = 0; i < n; i++) if (x[i] != x[i]) return NAN;
    float *s = (float *)malloc(sizeof(float) * n);
    double a = np_percentile_f32(x, n, qa, s);
    double b = np_percentile_f32(x, n, qb, s);
    free(s);
    return a - b;
}

/* ------------------------------------------------------------------ N1 / D1 */

/* N1: normalisation parameters of one minibatch (batch[:, :T] of a row-major [N, m] matrix).
 * out[0]=med, out[1]=mad (float32 values as double), out[2]=lo32, out[3]=hi32.
 * returns 0, or -1 if mad == 0 (the reference raises ValueError, normalize.py:56-59). */
int orc_norm_params(const float *batch, long N, long m, long T, double thresh, double *out)
{
    if (T > m) T = m;
    long tot = N * T;
    float *buf = (float *)malloc(sizeof(float) * (tot > 0 ? tot : 1));
    long c = 0;
    for (long r = 0; r < N; r++) {
        const float *row = batch + r * m;
        for (long i = 0; i < T; i++) if (row[i] == row[i]) buf[c++] = row[i];
    }
    float med = median_inplace_f32(buf, c);
    c = 0;
    for (long r = 0; r < N; r++) {
        const float *row = batch + r * m;
        for (long i = 0; i < T; i++) { float d = fabsf(row[i] - med); if (d == d) buf[c++] = d; }
    }
    float mad = median_inplace_f32(buf, c);
    free(buf);
    double dmed = (double)med, dmad = (double)mad;
    out[0] = dmed; out[1] = dmad;
    out[2] = (double)(float)(dmed - dmad * thresh);  /* np.clip bounds: python floats cast to f32 */
    out[3] = (double)(float)(dmed + dmad * thresh);
    if (mad == 0.0f) return -1;
    return 0;
}

static inline float norm1(float x, float med, float mad, float lo, float hi)
{
    /* np.clip = minimum(maximum(x, lo), hi); NaN propagates */
    float c = x;
    if (c == c) { if (c < lo) c = lo; if (c > hi) c = hi; }
    return (c - med) / mad;
}

/* D1 on one row: pooled[j] = mean_f32(norm(row[off + j*ds .. +ds)))  (zero padded tail).
 * normalise != 0: apply N1 first (values beyond T are not part of the array at all).
 * Lseg = number of input samples (e.g. T - min_obs); returns number of pooled outputs. */
long orc_pool_row(const float *row, long Lseg, int ds, int normalise,
                  float med, float mad, float lo, float hi, float *out)
{
    long L = (Lseg + ds - 1) / ds;
    float tmp[256];
    for (long j = 0; j < L; j++) {
        for (int k = 0; k < ds; k++) {
            long i = j * ds + k;
            float v = 0.0f;
            if (i < Lseg) v = normalise ? norm1(row[i], med, mad, lo, hi) : row[i];
            tmp[k] = v;
        }
        out[j] = pw_f32(tmp, ds) / (float)ds;
    }
    return L;
}

/* ------------------------------------------------------------------ G1 / G2 */

static inline double var_c(long start, long end, const double *c, const double *c2)
{
    if (start == end) return 0;
    if (start == 0) {
        double m = c[end - 1] / (double)end;
        return c2[end - 1] / (double)end - m * m;
    }
    double d = (double)(end - start);
    double m = (c[end - 1] - c[start - 1]) / d;
    return (c2[end - 1] - c2[start - 1]) / d - m * m;
}

void orc_cumsum(const float *s, long n, double *c, double *c2)
{
    double a = 0, b = 0;
    for (long i = 0; i < n; i++) {
        double v = (double)s[i];
        if (i == 0) { a = v; b = v * v; } else { a += v; b += v * v; }
        c[i] = a; c2[i] = b;
    }
}

/* _gains(start, end, c, c2, offset_head, offset_tail, stride=1) -> gains (n = len(c)) */
void orc_gains(long start, long end, const double *c, const double *c2, long n,
               long offset_head, long offset_tail, double *g)
{
    for (long i = 0; i < n; i++) g[i] = 0.0;
    if (n == 0) return;
    double vs = (double)(end - start) * log(var_c(start, end, c, c2));
    for (long i = start + offset_head; i < end - offset_tail; i++) {
        double h = (double)(i - start) * log(var_c(start, i, c, c2));
        double t = (double)(end - i) * log(var_c(i, end, c, c2));
        g[i] = vs - (h + t);
    }
}

/* ------------------------------------------------------------------ the _c_llr module as an API (SURVEY 8(b) "Native", 8(f) rank 4)
 * c_llr_trace / c_llr_trace_gains / _gains with stride and both early-stopping forms, reference adapted/detect/_c_llr.pyx:67-236.
 * np.cumsum of float64 (first element taken as is), np.multiply(raw, raw) rounded before it is added. */
void orc_cumsum_f64(const double *raw, long n, double *c, double *c2)
{
    double a = 0, b = 0;
    for (long i = 0; i < n; i++) {
        double v = raw[i], q = v * v;
        if (i == 0) { a = v; b = q; } else { a += v; b += q; }
        c[i] = a; c2[i] = b;
    }
}

/* np.diff(g[lo:i:stride]).mean(), Python slice rules for a negative lo (_c_llr.pyx:115, :160, :165); NaN for < 2 elements */
static double diff_mean(const double *g, long n, long lo, long i, long stride, double *tmp)
{
    if (lo < 0) { lo += n; if (lo < 0) lo = 0; }
    long cnt = 0;
    if (i > lo) cnt = (i - lo + stride - 1) / stride;
    if (cnt < 2) return NAN; /* mean of an empty array */
    for (long k = 0; k + 1 < cnt; k++) tmp[k] = g[lo + (k + 1) * stride] - g[lo + k * stride];
    return orc_np_sum_f64(tmp, cnt - 1) / (double)(cnt - 1);
}

/* -> 0, or -1 when the reference's assert (early-stop stride % stride == 0) fails.  tmp: >= n doubles of scratch. */
int orc_c_llr_trace_gains(const double *c, const double *c2, long n, long start, long end, long min_obs, long border_trim,
                          long stride, long a_es, long a_w, long a_s, long p_es, long p_w, long p_s, double *g, double *tmp)
{
    for (long i = 0; i < n; i++) g[i] = 0.0;
    if (p_es > 0) { if (a_s % stride || p_s % stride) return -1; }
    else if (a_es > 0) { if (a_s % stride) return -1; }
    const long s0 = start + min_obs;
    const double vs = (double)(end - start) * log(var_c(start, end, c, c2));
    int adapter_found = 0;
    for (long i = s0; i < end - border_trim; i += stride) {
        if (p_es > 0) {
            if (!adapter_found && i >= s0 + a_w && (i - s0) % a_s == 0) {
                if (diff_mean(g, n, i - a_w, i, stride, tmp) < 0) adapter_found = 1;
            }
            if (adapter_found) {
                if (diff_mean(g, n, i - p_w, i, stride, tmp) > 0) break;
            }
        } else if (a_es > 0) {
            if (i >= s0 + a_w && (i - s0) % a_s == 0) {
                if (diff_mean(g, n, i - a_w, i, stride, tmp) < 0) break;
            }
        }
        double h = (double)(i - start) * log(var_c(start, i, c, c2));
        double t = (double)(end - i) * log(var_c(i, end, c, c2));
        g[i] = vs - (h + t);
    }
    return 0;
}

/* ------------------------------------------------------------------ scipy find_peaks */

typedef struct {
    int use_distance; double distance;
    int use_prominence; double pmin;
    int use_width; double wmin;
    double rel_height;
} fp_opts;

static long local_maxima(const double *x, long n, long *peaks)
{
    long m = 0, i = 1, imax = n - 1;
    while (i < imax) {
        if (x[i - 1] < x[i]) {
            long ia = i + 1;
            while (ia < imax && x[ia] == x[i]) ia++;
            if (x[ia] < x[i]) {
                peaks[m++] = (i + ia - 1) / 2;
                i = ia;
            }
        }
        i++;
    }
    return m;
}

typedef struct { double pr; long idx; } prio_t;
static int cmp_prio(const void *a, const void *b)
{
    const prio_t *x = (const prio_t *)a, *y = (const prio_t *)b;
    if (x->pr < y->pr) return -1;
    if (x->pr > y->pr) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

static void peak_prominence(const double *x, long n, long peak, double *prom, long *lb, long *rb)
{
    long i = peak; *lb = peak;
    double left_min = x[peak];
    while (0 <= i && x[i] <= x[peak]) {
        if (x[i] < left_min) { left_min = x[i]; *lb = i; }
        i--;
    }
    i = peak; *rb = peak;
    double right_min = x[peak];
    while (i <= n - 1 && x[i] <= x[peak]) {
        if (x[i] < right_min) { right_min = x[i]; *rb = i; }
        i++;
    }
    *prom = x[peak] - (left_min > right_min ? left_min : right_min);
}

static double peak_width(const double *x, long peak, double prom, long lb, long rb, double rel_height)
{
    double height = x[peak] - prom * rel_height;
    long i = peak;
    while (lb < i && height < x[i]) i--;
    double left_ip = (double)i;
    if (x[i] < height) left_ip += (height - x[i]) / (x[i + 1] - x[i]);
    i = peak;
    while (i < rb && height < x[i]) i++;
    double right_ip = (double)i;
    if (x[i] < height) right_ip -= (height - x[i]) / (x[i - 1] - x[i]);
    return right_ip - left_ip;
}

/* returns number of peaks written to out (capacity cap; counting continues past cap) */
long orc_find_peaks(const double *x, long n, int use_distance, double distance,
                    int use_prominence, double pmin, int use_width, double wmin,
                    double rel_height, long *out, long cap)
{
    if (n < 3) return 0;
    long *peaks = (long *)malloc(sizeof(long) * (n / 2 + 2));
    long np_ = local_maxima(x, n, peaks);
    if (use_distance && np_ > 0) {
        long dist = (long)ceil(distance);
        unsigned char *keep = (unsigned char *)malloc(np_);
        prio_t *pr = (prio_t *)malloc(sizeof(prio_t) * np_);
        for (long i = 0; i < np_; i++) { keep[i] = 1; pr[i].pr = x[peaks[i]]; pr[i].idx = i; }
        qsort(pr, np_, sizeof(prio_t), cmp_prio);
        for (long i = np_ - 1; i >= 0; i--) {
            long j = pr[i].idx;
            if (!keep[j]) continue;
            long k = j - 1;
            while (0 <= k && peaks[j] - peaks[k] < dist) { keep[k] = 0; k--; }
            k = j + 1;
            while (k < np_ && peaks[k] - peaks[j] < dist) { keep[k] = 0; k++; }
        }
        long w = 0;
        for (long i = 0; i < np_; i++) if (keep[i]) peaks[w++] = peaks[i];
        np_ = w;
        free(keep); free(pr);
    }
    long m = 0;
    for (long p = 0; p < np_; p++) {
        if (use_prominence || use_width) {
            double prom; long lb, rb;
            peak_prominence(x, n, peaks[p], &prom, &lb, &rb);
            if (use_prominence && !(pmin <= prom)) continue;
            if (use_width) {
                double w = peak_width(x, peaks[p], prom, lb, rb, rel_height);
                if (!(wmin <= w)) continue;
            }
        }
        if (m < cap) out[m] = peaks[p];
        m++;
    }
    free(peaks);
    return m;
}

/* ------------------------------------------------------------------ T1 / P1..P4 */

/* LLRTrace._trace_start_end (llr.py:135-142) */
void orc_trace_start_end(const double *g, long n, long *start, long *end)
{
    long s = 0, e = 0;
    for (long i = 0; i < n; i++) if (!(g[i] <= 0)) { s = i; break; }
    for (long i = 0; i < n; i++) if (!(g[n - 1 - i] <= 0)) { e = i; break; }
    *start = s; *end = n - e - 1;
}

/* correct_for_plateau (llr.py:145-177), s=10, t=0.9, window=500 */
long orc_correct_plateau(const double *g, long n, long peak)
{
    const long s = 10, window = 500;
    const double t = 0.9;
    long wn = (peak + window < n ? peak + window : n) - peak;
    if (wn <= 0) return peak;
    const double *w = g + peak;
    long nch = wn - 1;
    long plateau_end = -1;
    for (long i = nch - s; i >= 0; i--) {
        int ok = 1;
        for (long j = i; j < i + (s - 1); j++) {
            if (j >= nch) break; /* python slice truncation (cannot happen for i <= nch - s) */
            if (!(w[j + 1] - w[j] >= 0)) { ok = 0; break; }
        }
        if (ok && w[i + (s - 1)] > t * w[0]) { plateau_end = i + (s - 1); break; }
    }
    if (plateau_end > 0) peak += plateau_end;
    return peak;
}

/* correct_for_split_peak (llr.py:180-201): find_peaks(window, width=10, prominence=1.0) */
long orc_correct_split(const double *g, long n, long peak)
{
    const long window = 500;
    long wn = (peak + window < n ? peak + window : n) - peak;
    if (wn <= 0) return peak;
    long first;
    long k = orc_find_peaks(g + peak, wn, 0, 0, 1, 1.0, 1, 10.0, 0.5, &first, 1);
    if (k > 0 && g[first + peak] >= 0.9 * g[peak]) return first + peak;
    return peak;
}

/* P1+P2+P3+A1: first adapter-end candidate in pooled units, or -1 when there is none.
 * raw_first (optional) receives the uncorrected first peak. */
long orc_adapter_candidate(const double *g, long n, double prominence, double rel_height, long width,
                           long *raw_first, long *n_peaks)
{
    long start, end;
    if (n_peaks) *n_peaks = 0;
    if (raw_first) *raw_first = -1;
    if (n == 0) return -1;
    orc_trace_start_end(g, n, &start, &end);
    long cn = end - start;
    if (cn <= 0) return -1;
    const double *clip = g + start;
    double sd = orc_np_nanstd_f64(clip, cn);
    long first;
    long k = orc_find_peaks(clip, cn, 0, 0, 1, prominence * sd, 1, (double)width, rel_height, &first, 1);
    if (n_peaks) *n_peaks = k;
    if (k <= 0) return -1;
    long peak = first + start;
    if (raw_first) *raw_first = peak;
    peak = orc_correct_plateau(g, n, peak);
    peak = orc_correct_split(g, n, peak);
    return peak;
}

/* P4 detect_full_polya_trace_peak_with_spike (llr.py:406-479) */
long orc_polya_peak(const double *g, long n)
{
    if (n < 3) return 0;
    double *x = (double *)malloc(sizeof(double) * n);
    for (long i = 0; i < n; i++) {
        double v = g[i];
        if (v != v) v = 0.0;                 /* np.nan_to_num(nan=0) */
        else if (isinf(v)) v = v > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
        x[i] = v;
    }
    long pk[2];
    long k = orc_find_peaks(x, n, 1, 10.0, 1, 1.0, 1, 10.0, 0.5, pk, 2);
    free(x);
    if (k == 0) return 0;
    if (k == 1) return pk[0];
    double h0 = g[pk[0]], h1 = g[pk[1]];
    if (h1 > h0) return pk[1];
    if (h1 < h0 * 0.5) return pk[0];
    /* idx_min = argmin(llr_trace[p0:p1]) (first minimum; NaN wins like np.argmin) */
    long idx_min = pk[0];
    {
        double mv = g[pk[0]];
        int nan_found = (mv != mv);
        for (long i = pk[0] + 1; i < pk[1] && !nan_found; i++) {
            if (g[i] != g[i]) { idx_min = i; nan_found = 1; break; }
            if (g[i] < mv) { mv = g[i]; idx_min = i; }
        }
    }
    long cnt = pk[1] - idx_min;
    if (cnt <= 0) return 0;
    /* scipy.stats.linregress r-value: cov(x, y, bias=1) */
    double *xs = (double *)malloc(sizeof(double) * cnt), *ys = (double *)malloc(sizeof(double) * cnt);
    for (long i = 0; i < cnt; i++) { xs[i] = (double)(idx_min + i); ys[i] = g[idx_min + i]; }
    double xm = orc_np_sum_f64(xs, cnt) / (double)cnt, ym = orc_np_sum_f64(ys, cnt) / (double)cnt;
    double sxx = 0, sxy = 0, syy = 0;
    for (long i = 0; i < cnt; i++) {
        double dx = xs[i] - xm, dy = ys[i] - ym;
        sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
    }
    free(xs); free(ys);
    double inv = 1.0 / (double)cnt; /* np.cov: c *= true_divide(1, fact) */
    double ssxm = sxx * inv, ssxym = sxy * inv, ssym = syy * inv;
    double r;
    if (ssxm == 0.0 || ssym == 0.0) r = 0.0;
    else {
        r = ssxym / sqrt(ssxm * ssym);
        if (r > 1.0) r = 1.0; else if (r < -1.0) r = -1.0;
    }
    if (r * r >= 0.99) return pk[1];
    return 0;
}

/* ------------------------------------------------------------------ bottleneck f32 */

/* move_mean / move_var (float32, NaN-free input), outputs for i >= window-1 only:
 * out[k] = value at index window-1+k, k in [0, n-window] */
/* NaN samples are counted out of the window; a window with fewer than w valid samples yields NaN
 * (min_count = window) -- bottleneck/src/move_template.c move_mean / move_var, every intermediate in float32 */
static void bn_move_mean_f32(const float *a, long n, long w, float *out)
{
    float asum = 0.f;
    long count = 0;
    for (long i = 0; i < w; i++) { float ai = a[i]; if (ai == ai) { asum += ai; count++; } }
    out[0] = count >= w ? asum / (float)count : NAN;
    float inv = (float)(1.0 / (double)count);
    for (long i = w; i < n; i++) {
        float ai = a[i], aold = a[i - w];
        if (ai == ai) {
            if (aold == aold) asum += ai - aold;
            else { asum += ai; count++; inv = (float)(1.0 / (double)count); }
        } else if (aold == aold) { asum -= aold; count--; inv = (float)(1.0 / (double)count); }
        out[i - w + 1] = count >= w ? asum * inv : NAN;
    }
}

static void bn_move_var_f32(const float *a, long n, long w, float *out)
{
    float amean = 0.f, assqdm = 0.f;
    long count = 0;
    for (long i = 0; i < w; i++) {
        float ai = a[i];
        if (ai == ai) {
            count++;
            float delta = ai - amean;
            amean += delta / (float)count;
            assqdm += delta * (ai - amean);
        }
    }
    if (count >= w) { if (assqdm < 0) assqdm = 0; out[0] = assqdm / (float)count; }
    else out[0] = NAN;
    float ddof_inv = (float)(1.0 / (double)count), count_inv = (float)(1.0 / (double)count);
    for (long i = w; i < n; i++) {
        float ai = a[i], aold = a[i - w];
        if (ai == ai) {
            if (aold == aold) {
                float delta = ai - aold;
                aold -= amean;
                amean += delta * count_inv;
                ai -= amean;
                assqdm += (ai + aold) * delta;
            } else {
                count++;
                count_inv = ddof_inv = (float)(1.0 / (double)count);
                float delta = ai - amean;
                amean += delta * count_inv;
                assqdm += delta * (ai - amean);
            }
        } else if (aold == aold) {
            count--;
            count_inv = ddof_inv = (float)(1.0 / (double)count);
            if (count > 0) {
                float delta = aold - amean;
                amean -= delta * count_inv;
                assqdm -= delta * (aold - amean);
            } else { amean = 0.f; assqdm = 0.f; }
        }
        if (count >= w) { if (assqdm < 0) assqdm = 0; out[i - w + 1] = assqdm * ddof_inv; }
        else out[i - w + 1] = NAN;
    }
}

void orc_bn_move_mean_f32(const float *a, long n, long w, float *out) { bn_move_mean_f32(a, n, w, out); }
void orc_bn_move_var_f32(const float *a, long n, long w, float *out) { bn_move_var_f32(a, n, w, out); }

/* ------------------------------------------------------------------ V2 / V3 / V4 / S1 */

static int in_range(double v, double lo, double hi) { return lo <= v && v <= hi; }

static int range_is_empty(const double *r) { return isinf(r[0]) && r[0] < 0 && isinf(r[1]) && r[1] > 0; }

/* find_open_pores on x[0:n] (anomalies.py:15-35); writes up to cap positions, returns count;
 * *last = last reported position (undefined when count == 0) */
static long find_open_pores(const float *x, long n, int32_t *out, long cap, long *last)
{
    long cnt = 0, npos = 0, prev = -1, lastpos = -1, nvalid = 0;
    for (long i = 0; i < n; i++) {
        if (200.0f <= x[i]) { /* in_range(signal, 200.0, None) */
            if (npos >= 1 && i - prev >= 10) {
                if (nvalid < cap) out[nvalid] = (int32_t)i;
                nvalid++; *last = i;
            }
            prev = i; lastpos = i; npos++;
        }
    }
    if (npos == 0) return 0;
    if (npos == 1) { out[0] = (int32_t)lastpos; *last = lastpos; return 1; }
    if (nvalid == 0) { out[0] = (int32_t)lastpos; *last = lastpos; return 1; }
    cnt = nvalid;
    return cnt;
}

static void partition_stats(const float *sig, long S, long has_start, long start, long has_end, long end,
                            orc_row *r, int c_start, int c_len)
{
    /* calc_partition_stats: (start, len, mean, std, med, mad); mean..mad follow c_len */
    if (has_start) row_set(r, c_start, (double)start);
    if (!has_start || !has_end || end <= start) return;
    long length = end - start;
    row_set(r, c_len, (double)length);
    long a = start < S ? start : S, b = end < S ? end : S;
    if (a < 0) a = 0; /* negative starts do not occur on this path */
    long n = b - a;
    const float *x = sig + a;
    float mean, sd, med, mad;
    if (n <= 0) { mean = sd = med = mad = NAN; }
    else {
        mean = orc_np_mean_f32(x, n);
        sd = orc_np_std_f32(x, n);
        med = orc_np_median_f32(x, n);
        mad = orc_np_mad_f32(x, n, med);
    }
    row_set(r, c_len + 1, (double)mean);
    row_set(r, c_len + 2, (double)sd);
    row_set(r, c_len + 3, (double)med);
    row_set(r, c_len + 4, (double)mad);
}

typedef struct { int ok; int vec_fail; double mean, var, med, lrange, shift; int exc; } mvs_out;

static mvs_out mvs_check(const float *sig, long S, long a_e, long p_e, const orc_cfg *cfg, const double *pA_mean_range)
{
    mvs_out o; memset(&o, 0, sizeof(o));
    o.vec_fail = 31;
    if (p_e == 0 || a_e == 0 || p_e < a_e || p_e - a_e <= 2) return o;
    if (S < a_e + cfg->median_shift_window) return o;
    long a = a_e < S ? a_e : S, b = p_e < S ? p_e : S;
    long n = b - a;
    const float *x = sig + a;
    float fvar, fmean;
    if (p_e - a_e <= cfg->pA_var_window + 2) fvar = orc_np_var_f32(x, n);
    else {
        long w = cfg->pA_var_window;
        if (w > n || w < 1) { o.exc = F_EXC_MOVE_WINDOW; return o; }
        float *t = (float *)malloc(sizeof(float) * (n - w + 1));
        bn_move_var_f32(x, n, w, t);
        fvar = orc_np_nanmedian_f32(t, n - w + 1, NULL);
        free(t);
    }
    if (p_e - a_e <= cfg->pA_mean_window + 2) fmean = orc_np_mean_f32(x, n);
    else {
        long w = cfg->pA_mean_window;
        if (w > n || w < 1) { o.exc = F_EXC_MOVE_WINDOW; return o; }
        float *t = (float *)malloc(sizeof(float) * (n - w + 1));
        bn_move_mean_f32(x, n, w, t);
        fmean = orc_np_nanmedian_f32(t, n - w + 1, NULL);
        free(t);
    }
    float fmed = orc_np_median_f32(x, n);
    double lrange = orc_np_percentile_diff_f32(x, n, 85.0, 15.0);
    long r1 = a_e + cfg->median_shift_window; if (r1 > S) r1 = S;
    long l0 = a_e - cfg->median_shift_window; if (l0 < 0) l0 = 0;
    float shift = orc_np_median_f32(sig + a, r1 - a) - orc_np_median_f32(sig + l0, a - l0);
    o.mean = (double)fmean; o.var = (double)fvar; o.med = (double)fmed; o.lrange = lrange; o.shift = (double)shift;
    int f = 0;
    if (!in_range(o.mean, pA_mean_range[0], pA_mean_range[1])) f |= 1;
    if (!in_range(o.var, cfg->pA_var_range[0], cfg->pA_var_range[1])) f |= 2;
    if (!in_range(o.med, cfg->polyA_med_range[0], cfg->polyA_med_range[1])) f |= 4;
    if (!in_range(o.lrange, cfg->polyA_local_range[0], cfg->polyA_local_range[1])) f |= 8;
    if (!in_range(o.shift, cfg->median_shift_range[0], cfg->median_shift_range[1])) f |= 16;
    o.vec_fail = f; o.ok = (f == 0);
    return o;
}

/* mean_var_shift_polyA_detect_at_loc(signal, loc, params, return_values=True, less_signal_ok=False)
 * (adapted/detect/mvs.py:181-338): first position in [loc - offset, loc + search_window) whose moving mean and
 * moving variance are both in range, then the poly(A) median / local range / median shift at max(loc, position).
 * The array comparisons of in_range (utils.py:26) are float32 array against scalar bounds: numpy 1.x casts the
 * bounds to float32 (value-based casting; the goldens were made with numpy 1.26), the scalar checks at the end
 * compare Python floats (float64). */
typedef struct { int ok, exc; long idx; double mean, var, med, lrange, shift; } mvs_loc_out;

static int in_range_f32(float v, double lo, double hi) { return (float)lo <= v && v <= (float)hi; }

static mvs_loc_out mvs_detect_at_loc(const float *sig, long S, long loc, const orc_cfg *cfg, const double *pA_mean_range)
{
    mvs_loc_out o; memset(&o, 0, sizeof(o));
    const long wm = cfg->pA_mean_window, wv = cfg->pA_var_window;
    const long offset = wm > wv ? wm : wv;
    const long tailw = cfg->median_shift_window > cfg->polyA_window ? cfg->median_shift_window : cfg->polyA_window;
    if (S < loc + cfg->search_window + tailw) return o; /* not enough signal after loc (:216-231) */
    if (loc < offset) return o;                         /* not enough signal before loc (:234-247) */
    const long n = offset + cfg->search_window;
    const float *x = sig + (loc - offset);
    if (wm < 1 || wv < 1 || wm > n || wv > n) { o.exc = F_EXC_MOVE_WINDOW; return o; }
    float *mm = (float *)malloc(sizeof(float) * n), *mv = (float *)malloc(sizeof(float) * n);
    /* bottleneck keeps the first window-1 outputs as NaN (min_count = window): full-length series here */
    for (long i = 0; i < n; i++) mm[i] = mv[i] = NAN;
    bn_move_mean_f32(x, n, wm, mm + wm - 1);
    bn_move_var_f32(x, n, wv, mv + wv - 1);
    long idx = 0;
    for (long i = 0; i < n; i++)
        if (in_range_f32(mm[i], pA_mean_range[0], pA_mean_range[1]) && in_range_f32(mv[i], cfg->pA_var_range[0], cfg->pA_var_range[1])) { idx = i; break; }
    float mean, var;
    if (idx > 0) { mean = mm[idx]; var = mv[idx]; idx += loc - offset; }
    else { mean = mm[2 * offset]; var = mv[2 * offset]; } /* (2*offset < n is required of the config) */
    free(mm); free(mv);
    o.idx = idx; o.mean = (double)mean; o.var = (double)var;
    const long loc_ = loc > idx ? loc : idx;
    long e1 = loc_ + cfg->polyA_window; if (e1 > S) e1 = S;
    long e2 = loc_ + cfg->median_shift_window; if (e2 > S) e2 = S;
    o.med = (double)orc_np_median_f32(sig + loc_, e1 - loc_);
    o.lrange = orc_np_percentile_diff_f32(sig + loc_, e1 - loc_, 85.0, 15.0);
    o.shift = (double)(orc_np_median_f32(sig + loc_, e2 - loc_) - orc_np_median_f32(sig, loc_));
    o.ok = idx > 0 && in_range(o.med, cfg->polyA_med_range[0], cfg->polyA_med_range[1]) &&
           in_range(o.lrange, cfg->polyA_local_range[0], cfg->polyA_local_range[1]) &&
           in_range(o.shift, cfg->median_shift_range[0], cfg->median_shift_range[1]);
    return o;
}

/* V1 validate_boundaries.  sig = row[:full_len] => S = min(full_len, m) samples.
 * cand == NULL <=> polya_end_topk is None. */
void orc_validate(const float *sig, long m, long full_len, long adapter_end_in, long polya_end_in,
                  const int64_t *cand, int n_cand, const orc_cfg *cfg, orc_row *r)
{
    memset(r, 0, sizeof(*r));
    long S = full_len < m ? full_len : m;
    long a_s = 0, a_e = adapter_end_in, p_best = polya_end_in;
    int p_none = 0; /* polya_end_best became None (mvs_detect_overwrite, combined.py:559-561) */
    int success = 1, fail = F_NONE;
    float adapter_med = 0, adapter_mad = 0; int have_med = 0;
    r->n_cand = -1; r->n_open_pores = -1;

    if (a_e == 0) { success = 0; fail = F_NO_ADAPTER; }
    else {
        long b = a_e < S ? a_e : S;
        adapter_med = orc_np_median_f32(sig, b);
        adapter_mad = orc_np_mad_f32(sig, b, adapter_med);
        have_med = 1;
    }
    if (success && have_med && adapter_mad != 0.0f &&
        !in_range((double)adapter_mad, cfg->adapter_mad_range[0], cfg->adapter_mad_range[1])) {
        success = 0; fail = F_ADAPTER_MAD;
    }
    if (success && cfg->detect_open_pores) {
        long b = a_e < S ? a_e : S, last = 0;
        long k = find_open_pores(sig, b, r->open_pores, ORC_MAX_OPEN_PORES, &last);
        r->n_open_pores = (int32_t)k;
        if (k > 0) {
            a_s = last;
            if (a_e - a_s < cfg->min_obs_adapter) { success = 0; fail = F_OPEN_PORE; }
        }
    }
    if (success && cfg->real_signal_check) {
        long a = a_s < S ? a_s : S, b = a_e < S ? a_e : S;
        long n = b - a; if (n < 0) n = 0;
        const float *x = sig + a;
        int ok = 0;
        if (n >= 2 * cfg->mean_window) {
            float ms = orc_np_mean_f32(x, cfg->mean_window);
            float me = orc_np_mean_f32(x + n - cfg->mean_window, cfg->mean_window);
            row_set(r, C_REAL_MEAN_START, (double)ms);
            row_set(r, C_REAL_MEAN_END, (double)me);
            if (in_range((double)ms, cfg->mean_start_range[0], cfg->mean_start_range[1]) &&
                in_range((double)me, cfg->mean_end_range[0], cfg->mean_end_range[1])) {
                long k = n < cfg->max_obs_local_range ? n : cfg->max_obs_local_range;
                double lr = orc_np_percentile_diff_f32(x + n - k, k, 85.0, 15.0);
                row_set(r, C_REAL_LOCAL_RANGE, lr);
                ok = in_range(lr, cfg->local_range[0], cfg->local_range[1]);
            }
        }
        if (!ok) { success = 0; fail = F_REAL_RANGE; }
    }
    if (success && cfg->mvs_detect_check) {
        if (p_best == 0) { success = 0; fail = F_NO_POLYA; }
        else {
            double pr[2] = { cfg->pA_mean_range[0], cfg->pA_mean_range[1] };
            int exc = 0;
            if (range_is_empty(pr) && !range_is_empty(cfg->pA_mean_adapter_med_scale_range)) {
                pr[0] = cfg->pA_mean_adapter_med_scale_range[0] * (double)adapter_med;
                pr[1] = cfg->pA_mean_adapter_med_scale_range[1] * (double)adapter_med;
            } else if (range_is_empty(pr)) { exc = F_EXC_PA_RANGE; }
            if (!exc && cand == NULL) exc = F_EXC_TOPK_NONE;
            if (exc) { memset(r, 0, sizeof(*r)); r->n_cand = -1; r->n_open_pores = -1; r->success = 0; r->fail_code = exc; return; }
            for (int c = 0; c < n_cand; c++) {
                long p_e = (long)cand[c];
                if (p_e == 0) break;
                if (cfg->mvs_detect_overwrite) {
                    /* combined.py:517-562: look for the adapter end in [loc, loc + search_window) by the MVS method */
                    mvs_loc_out o = mvs_detect_at_loc(sig, S, a_e, cfg, pr);
                    if (o.exc) { memset(r, 0, sizeof(*r)); r->n_cand = -1; r->n_open_pores = -1; r->success = 0; r->fail_code = o.exc; return; }
                    row_set(r, C_MVS_ADAPTER_END, (double)o.idx);
                    row_set(r, C_MVS_MEAN, o.mean); row_set(r, C_MVS_VAR, o.var);
                    row_set(r, C_MVS_POLYA_MED, o.med); row_set(r, C_MVS_LOCAL_RANGE, o.lrange);
                    row_set(r, C_MVS_MED_SHIFT, o.shift);
                    int p_e_none = 0;
                    if (!o.ok) { success = 0; fail = F_NO_ADAPTER_MVS; }
                    else if (o.idx - a_e > 0) {
                        a_e = o.idx;
                        /* Boundaries.polya_end_adjust, .polya_truncated and .trace_early_stop_pos are None on every
                         * v0.2.4 call path (combined.py:146-152, :324-328, cnn :280-300): the new poly(A) end is None */
                        if (a_e > p_e) { p_e_none = 1; r->mvs_fail_mask |= MVS_FLAG_TO_EARLY_STOP; }
                    }
                    if (success) { p_best = p_e; p_none = p_e_none; break; }
                    continue;
                }
                mvs_out o = mvs_check(sig, S, a_e, p_e, cfg, pr);
                if (o.exc) { memset(r, 0, sizeof(*r)); r->n_cand = -1; r->n_open_pores = -1; r->success = 0; r->fail_code = o.exc; return; }
                row_set(r, C_MVS_MEAN, o.mean); row_set(r, C_MVS_VAR, o.var);
                row_set(r, C_MVS_POLYA_MED, o.med); row_set(r, C_MVS_LOCAL_RANGE, o.lrange);
                row_set(r, C_MVS_MED_SHIFT, o.shift);
                if (!o.ok) {
                    success = 0;
                    if (o.mean == 0) { fail = F_MVS_NOT_ENOUGH; r->mvs_fail_mask = 0; }
                    else { fail = F_MVS_CHECKS; r->mvs_fail_mask = o.vec_fail; }
                }
                if (success) { p_best = p_e; break; }
            }
        }
    }
    if (success && cfg->detect_med_shift) {
        long w = cfg->med_shift_window;
        long r1 = a_e + w; if (r1 > full_len) r1 = full_len; if (r1 > S) r1 = S;
        long a = a_e < S ? a_e : S;
        long l0 = a_e - w; if (l0 < 0) l0 = 0; if (l0 > S) l0 = S;
        float sh = orc_np_median_f32(sig + a, r1 - a) - orc_np_median_f32(sig + l0, a - l0);
        row_set(r, C_MED_SHIFT, (double)sh);
        if (!in_range((double)sh, cfg->med_shift_range[0], cfg->med_shift_range[1])) { success = 0; fail = F_MED_SHIFT; }
    }
    partition_stats(sig, S, 1, a_s, 1, a_e, r, C_ADAPTER_START, C_ADAPTER_LEN);
    partition_stats(sig, S, 1, a_e, !p_none, p_best, r, C_POLYA_START, C_POLYA_LEN);
    partition_stats(sig, S, !p_none, p_best, 1, S, r, C_RNA_START, C_RNA_LEN);
    /* DetectResults(adapter_end=..., polya_end=...) are set explicitly (combined.py:603-604) */
    row_set(r, C_ADAPTER_END, (double)a_e);
    if (!p_none) row_set(r, C_POLYA_END, (double)p_best);
    row_set(r, C_SIGNAL_LEN, (double)full_len);
    row_set(r, C_PRELOADED, (double)S);
    row_set(r, C_PRIMARY_ADAPTER_END, (double)adapter_end_in);
    row_set(r, C_PRIMARY_POLYA_END, (double)polya_end_in);
    if (cand) {
        r->n_cand = n_cand;
        for (int c = 0; c < n_cand && c < ORC_MAX_CAND; c++) r->cand[c] = cand[c];
    }
    r->success = success; r->fail_code = fail;
}

/* ------------------------------------------------------------------ K1 start peak */

typedef struct {
    int32_t valid;          /* 0 <=> the (None, ..., None) row of start_peak.py:83-84 */
    int32_t flagged_type;   /* 0 None, 1 open pore in adapter, 2 potential concatemer */
    int32_t has_open_pore;  /* open_pore_idx column not None */
    int32_t _pad;
    int64_t start_peak_idx, next_greater_idx, open_pore_idx; /* already multiplied by ds */
    float start_peak_pa, next_greater_pa;
} orc_sp;

int orc_sizeof_sp(void) { return (int)sizeof(orc_sp); }

void orc_start_peak_row(const float *row, long m, long full_len, const orc_cfg *cfg, orc_sp *o)
{
    memset(o, 0, sizeof(*o));
    int ds = cfg->sp_downscale_factor;
    long off1 = cfg->sp_offset1, spmax = cfg->start_peak_max_idx, off2 = cfg->sp_offset2;
    long end_idx = (full_len < m ? full_len : m) / ds;
    long L = (m + ds - 1) / ds;
    float *p = (float *)malloc(sizeof(float) * (L > 0 ? L : 1));
    orc_pool_row(row, m, ds, 0, 0, 1, 0, 0, p);
    /* open pore: argmax(raw[:end_idx] > open_pore_pa) // ds ; recorded only if > 0 */
    long op = 0;
    for (long i = 0; i < end_idx && i < m; i++) if (row[i] > (float)cfg->open_pore_pa) { op = i; break; }
    op /= ds;
    int has_op = op > 0;
    /* max over pooled[off1:spmax] (python slice clipping); NaN propagates like np.max */
    long a = off1 < L ? off1 : L, b = spmax < L ? spmax : L;
    if (b - a <= 0) { free(p); return; }           /* .max() of an empty slice raises */
    float mx = p[a]; int isnan_ = (mx != mx);
    for (long i = a + 1; i < b; i++) { if (p[i] != p[i]) isnan_ = 1; else if (p[i] > mx) mx = p[i]; }
    if (isnan_) mx = NAN;
    long max_idx = 0;  /* argmax(pooled[a:b] == max_) : first True, 0 if none (NaN) */
    for (long i = a; i < b; i++) if (p[i] == mx) { max_idx = i - a; break; }
    max_idx += off1;
    long s0 = spmax + off2;
    long e0 = end_idx < L ? end_idx : L;
    long a2 = s0 < L ? s0 : L;
    if (e0 - a2 <= 0) { free(p); return; }         /* argmax of an empty slice raises */
    long nxt = 0;
    for (long i = a2; i < e0; i++) if (p[i] > mx) { nxt = i - a2; break; }
    nxt += s0;
    if (nxt >= L) { free(p); return; }             /* IndexError */
    float nxt_v = p[nxt];
    o->valid = 1;
    o->start_peak_idx = max_idx * ds; o->start_peak_pa = mx;
    o->next_greater_idx = nxt * ds; o->next_greater_pa = nxt_v;
    if (has_op) {
        /* np.isclose(next, open, atol=2, rtol=0.01): |a-b| <= atol + rtol*|b| */
        if (fabs((double)nxt - (double)op) <= 2.0 + 0.01 * fabs((double)op)) { o->flagged_type = 1; }
        else if (max_idx < op && op < nxt) { o->flagged_type = 2; }
        if (o->flagged_type) { o->has_open_pore = 1; o->open_pore_idx = op * ds; }
    }
    free(p);
}

/* ------------------------------------------------------------------ drivers */

/* LLR primary detection for one read given the minibatch normalisation parameters.
 * Returns 0; -2 if the read has no valid pooled block (the reference crashes the
 * minibatch there: np.argmin of an empty trace, llr.py:136).
 * Optional stage outputs (may be NULL): down[L], g1[L], g2[L]. */
int orc_llr_primary(const float *row, long m, const orc_cfg *cfg, const double *np4,
                    long *adapter_end, long *polya_end, long *n_valid_out,
                    float *down_out, double *g1_out, double *g2_out, long *stage_idx)
{
    long T = cfg->max_obs_trace < m ? cfg->max_obs_trace : m;
    long off = cfg->min_obs_adapter;
    int ds = cfg->downscale_factor;
    long Lseg = T - off; if (Lseg < 0) Lseg = 0;
    long L = (Lseg + ds - 1) / ds;
    float *down = (float *)malloc(sizeof(float) * (L > 0 ? L : 1));
    orc_pool_row(row + off, Lseg, ds, 1, (float)np4[0], (float)np4[1], (float)np4[2], (float)np4[3], down);
    long n_nan = 0;
    for (long j = 0; j < L; j++) if (down[j] != down[j]) n_nan++;
    long n = L - n_nan;
    if (n_valid_out) *n_valid_out = n;
    *adapter_end = 0; *polya_end = 0;
    if (stage_idx) { stage_idx[0] = stage_idx[1] = stage_idx[2] = stage_idx[3] = -1; }
    if (down_out) memcpy(down_out, down, sizeof(float) * L);
    if (n == 0) { free(down); return -2; }
    double *c = (double *)malloc(sizeof(double) * n), *c2 = (double *)malloc(sizeof(double) * n);
    double *g = (double *)malloc(sizeof(double) * n);
    orc_cumsum(down, n, c, c2);
    orc_gains(0, n - 1, c, c2, n, 5, 5, g);
    if (g1_out) memcpy(g1_out, g, sizeof(double) * n);
    long raw_first, npk;
    long cand = orc_adapter_candidate(g, n, cfg->adapter_peak_prominence, cfg->adapter_peak_rel_height,
                                      cfg->adapter_peak_width / ds, &raw_first, &npk);
    if (stage_idx) { stage_idx[0] = raw_first; stage_idx[1] = cand; }
    if (cand >= 0) {
        if (cand > 0) *adapter_end = cand * ds + off;
        orc_gains(cand, n - 1, c, c2, n, 1, 1, g);
        if (g2_out) memcpy(g2_out, g, sizeof(double) * n);
        long pe = orc_polya_peak(g, n);
        if (stage_idx) stage_idx[2] = pe;
        if (pe > 0) *polya_end = pe * ds + off;
    }
    free(down); free(c); free(c2); free(g);
    return 0;
}

/* combined_detect_llr (adapted/detect/combined.py:39-119; API only, no call site in v0.2.4): ONE read, normalised by its own
 * median / MAD over signal[:T]; the pooled signal starts at sample 0 (no min_obs_adapter slice) with offset_head =
 * 5 + min_obs_adapter // ds, yet min_obs_adapter is still added to the positions; an adapter candidate at index 0 ends the
 * search (no poly(A) trace); validate_boundaries sees the whole signal.  `row` holds the read's first min(len, m) samples,
 * NaN behind them (the reference is handed the unpadded array: its pooling pads the read's end with zeros).
 * returns 0, -1 (MAD == 0), -2 (no pooled block): the reference raises in both cases. */
int orc_detect_llr_single(const float *row, long m, long full_len, const orc_cfg *cfg, orc_row *out)
{
    double np4[4];
    int rc = orc_norm_params(row, 1, m, cfg->max_obs_trace, cfg->sig_norm_outlier_thresh, np4);
    if (rc) return rc;
    long T = cfg->max_obs_trace < m ? cfg->max_obs_trace : m;
    long have = full_len < m ? full_len : m;
    long Lseg = T < have ? T : have; if (Lseg < 0) Lseg = 0;
    int ds = cfg->downscale_factor;
    long L = (Lseg + ds - 1) / ds;
    float *down = (float *)malloc(sizeof(float) * (L > 0 ? L : 1));
    orc_pool_row(row, Lseg, ds, 1, (float)np4[0], (float)np4[1], (float)np4[2], (float)np4[3], down);
    long n_nan = 0;
    for (long j = 0; j < L; j++) if (down[j] != down[j]) n_nan++;
    long n = L - n_nan;
    if (n == 0) { free(down); return -2; }
    double *c = (double *)malloc(sizeof(double) * n), *c2 = (double *)malloc(sizeof(double) * n);
    double *g = (double *)malloc(sizeof(double) * n);
    orc_cumsum(down, n, c, c2);
    orc_gains(0, n - 1, c, c2, n, 5 + cfg->min_obs_adapter / ds, 5, g);
    long raw_first, npk, ae = 0, pe = 0;
    long cand = orc_adapter_candidate(g, n, cfg->adapter_peak_prominence, cfg->adapter_peak_rel_height,
                                      cfg->adapter_peak_width / ds, &raw_first, &npk);
    if (cand > 0) {
        ae = cand * ds + cfg->min_obs_adapter;
        orc_gains(cand, n - 1, c, c2, n, 1, 1, g);
        long p = orc_polya_peak(g, n);
        if (p > 0) pe = p * ds + cfg->min_obs_adapter;
    }
    free(down); free(c); free(c2); free(g);
    int64_t cd = pe;
    orc_validate(row, m, full_len, ae, pe, pe > 0 ? &cd : NULL, 1, cfg, out);
    return 0;
}

/* combined_detect_llr2 over one minibatch [N, m].  with_start_peak != 0 additionally fills
 * the start_peak_* columns from K1 (a build extension; default off).
 * returns 0, -1 (MAD == 0: the reference raises and drops the minibatch), -2 (empty trace). */
int orc_detect_llr_minibatch(const float *batch, const int32_t *full_len, long N, long m,
                             const orc_cfg *cfg, orc_row *rows, int with_start_peak, double *np4_out)
{
    double np4[4];
    int rc = orc_norm_params(batch, N, m, cfg->max_obs_trace, cfg->sig_norm_outlier_thresh, np4);
    if (np4_out) memcpy(np4_out, np4, sizeof(np4));
    if (rc) return rc;
    for (long r = 0; r < N; r++) {
        long ae, pe;
        rc = orc_llr_primary(batch + r * m, m, cfg, np4, &ae, &pe, NULL, NULL, NULL, NULL, NULL);
        if (rc) return rc;
        int64_t cand = pe;
        /* polya_end_topk is only assigned when a poly(A) end was found (combined.py:206-210) */
        orc_validate(batch + r * m, m, full_len[r], ae, pe, pe > 0 ? &cand : NULL, 1, cfg, &rows[r]);
        if (with_start_peak) {
            orc_sp sp;
            orc_start_peak_row(batch + r * m, m, full_len[r], cfg, &sp);
            if (sp.valid) {
                row_set(&rows[r], C_SP_IDX, (double)sp.start_peak_idx);
                row_set(&rows[r], C_SP_PA, (double)sp.start_peak_pa);
                row_set(&rows[r], C_SP_NEXT_IDX, (double)sp.next_greater_idx);
                row_set(&rows[r], C_SP_NEXT_PA, (double)sp.next_greater_pa);
                if (sp.has_open_pore) row_set(&rows[r], C_SP_OPEN_PORE_IDX, (double)sp.open_pore_idx);
                rows[r].start_peak_type = sp.flagged_type;
            }
        }
    }
    return 0;
}

/* combined_detect_start_peak (combined.py:312-355) over one minibatch.
 * Mirrors the pandas quirk: one all-None row turns the index columns into float64 and
 * every read of the minibatch then fails on the slice TypeError. */
int orc_detect_start_peak_minibatch(const float *batch, const int32_t *full_len, long N, long m,
                                    const orc_cfg *cfg, orc_row *rows)
{
    orc_sp *sp = (orc_sp *)malloc(sizeof(orc_sp) * (N > 0 ? N : 1));
    int any_none = 0;
    for (long r = 0; r < N; r++) {
        orc_start_peak_row(batch + r * m, m, full_len[r], cfg, &sp[r]);
        if (!sp[r].valid) any_none = 1;
    }
    for (long r = 0; r < N; r++) {
        orc_row *o = &rows[r];
        if (any_none) {
            memset(o, 0, sizeof(*o)); o->n_cand = -1; o->n_open_pores = -1;
            o->success = 0; o->fail_code = F_EXC_SLICE;
            continue;
        }
        long ng = (long)sp[r].next_greater_idx;
        orc_validate(batch + r * m, m, full_len[r], ng, ng, NULL, 0, cfg, o);
        if (o->fail_code == F_EXC_TOPK_NONE) continue; /* bare DetectResults(success=False, ...) */
        row_set(o, C_SP_IDX, (double)sp[r].start_peak_idx);
        row_set(o, C_SP_PA, (double)sp[r].start_peak_pa);
        row_set(o, C_SP_NEXT_IDX, (double)sp[r].next_greater_idx);
        row_set(o, C_SP_NEXT_PA, (double)sp[r].next_greater_pa);
        if (sp[r].has_open_pore) row_set(o, C_SP_OPEN_PORE_IDX, (double)sp[r].open_pore_idx);
        o->start_peak_type = sp[r].flagged_type;
        if (sp[r].flagged_type) o->success = 0; /* fail_reason gets "+<type>" appended only if already failed */
    }
    free(sp);
    return 0;
}

/* ------------------------------------------------------------------ CNN path (C1, C4) */

/* C1 prepare_data for one read (adapted/detect/cnn.py:70-82): pooled raw signal from min_obs_adapter,
 * per-read nanmedian / MAD, (x - med)/mad, torch.nan_to_num(-5.0).  out[Lc], Lc = ceil((m-off)/ds). */
long orc_cnn_prepare_row(const float *row, long m, const orc_cfg *cfg, float *out)
{
    long off = cfg->min_obs_adapter; int ds = cfg->downscale_factor;
    long Lc = orc_pool_row(row + off, m - off, ds, 0, 0, 1, 0, 0, out);
    float med = orc_np_nanmedian_f32(out, Lc, NULL);
    float *t = (float *)malloc(sizeof(float) * (Lc > 0 ? Lc : 1));
    for (long j = 0; j < Lc; j++) t[j] = fabsf(out[j] - med);
    float mad = orc_np_nanmedian_f32(t, Lc, NULL);
    free(t);
    for (long j = 0; j < Lc; j++) {
        float v = (out[j] - med) / mad;
        if (v != v) v = -5.0f;
        else if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
        out[j] = v;
    }
    return Lc;
}

/* C4 "hail mary" (adapted/detect/combined.py:259-298): per-read normalisation of signal[:min(T, full_len)],
 * pooling of [adapter_end:polya_end], LLR trace with offsets 5/5, P4.  Returns the new polya_end in samples
 * (0 if none); *status = 0, 13 (empty trace: np.argmin raises) or 14 (MAD == 0). */
long orc_cnn_fallback(const float *row, long m, long full_len, long a_e, long p_e, const orc_cfg *cfg, int *status)
{
    long S = full_len < m ? full_len : m;
    long T = cfg->max_obs_trace < S ? cfg->max_obs_trace : S;
    double np4[4];
    *status = 0;
    if (orc_norm_params(row, 1, m, T, cfg->sig_norm_outlier_thresh, np4)) { *status = 14; return 0; }
    long b = p_e < T ? p_e : T;
    long Lseg = b > a_e ? b - a_e : 0;
    int ds = cfg->downscale_factor;
    long L = (Lseg + ds - 1) / ds;
    float *down = (float *)malloc(sizeof(float) * (L > 0 ? L : 1));
    orc_pool_row(row + a_e, Lseg, ds, 1, (float)np4[0], (float)np4[1], (float)np4[2], (float)np4[3], down);
    long n_nan = 0;
    for (long j = 0; j < L; j++) if (down[j] != down[j]) n_nan++;
    long n = L - n_nan;
    if (n <= 0) { free(down); *status = 13; return 0; }
    double *c = (double *)malloc(sizeof(double) * n), *c2 = (double *)malloc(sizeof(double) * n), *g = (double *)malloc(sizeof(double) * n);
    orc_cumsum(down, n, c, c2);
    orc_gains(0, n - 1, c, c2, n, 5, 5, g);
    long pe = orc_polya_peak(g, n);
    free(down); free(c); free(c2); free(g);
    return pe > 0 ? pe * ds + a_e : 0;
}

/* combined_detect_cnn given the CNN's predictions preds[N][1+k] (adapted/detect/combined.py:243-305) */
void orc_detect_cnn_from_preds(const float *batch, const int32_t *full_len, long N, long m, const int64_t *preds, int k,
                               const orc_cfg *cfg, orc_row *rows)
{
    for (long r = 0; r < N; r++) {
        const float *row = batch + r * m;
        const int64_t *p = preds + r * (1 + k);
        orc_validate(row, m, full_len[r], (long)p[0], (long)p[1], p + 1, k, cfg, &rows[r]);
        if (!rows[r].success && !(rows[r].fail_code >= 9 && rows[r].fail_code <= 14) && p[0] > 0 && p[1] > 0 && p[1] - p[0] > 1000 &&
            full_len[r] < 2 * (long)cfg->max_obs_adapter && cfg->fallback_to_llr_short_reads) {
            int st;
            long pe = orc_cnn_fallback(row, m, full_len[r], (long)p[0], (long)p[1], cfg, &st);
            if (st) {
                memset(&rows[r], 0, sizeof(orc_row)); rows[r].n_cand = -1; rows[r].n_open_pores = -1; rows[r].fail_code = st;
            } else if (pe > 0) {
                int64_t c1 = pe;
                orc_validate(row, m, full_len[r], (long)p[0], pe, &c1, 1, cfg, &rows[r]);
            }
        }
    }
}
