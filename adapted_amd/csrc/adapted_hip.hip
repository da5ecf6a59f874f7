// adapted_hip.hip -- libadapted_hip.so: C ABI (include/adapted_hip.h) over the gfx950 kernels.
// One translation unit; kernels live in the headers next to this file.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "adapted_hip.h"
#include "common.h"
__device__ int g_ablate = 0;
__device__ unsigned long long g_dbg[ADP_NDBG] = {0};
// k_partition_stats' tallies (slots 0-4 of what = 8), spread over ADP_NTALLY cache lines by workgroup: every workgroup adding to ONE
// address serialises device-wide -- 2 x 96 000 same-address atomics were 7.7 ms of the kernel's 13 at the default window
__device__ unsigned long long g_bs_tally[ADP_NTALLY][8] = {{0}};
#include "llr_stream.h"
#include "n1_select.h"
#include "n1_fused.h"
#include "peaks.h"
#include "synth.h"
#include "validate.h"
#include "cand_stats2.h"
#include "series_pipe.h"
#include "cnn_topk.h"
#include "cnn_conv.h"
#include "cnn_conv_split.h"
#include "trace_api.h"
#include "wave_stats.h"

static thread_local std::string g_err;

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
            return ADP_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) { if (hipFree(p) != hipSuccess) return -1; p = nullptr; cap = 0; }
        if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return -1; }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

struct ProfEntry { const char *name; hipEvent_t a, b; };
#define ADP_MAX_LANES 4

struct adp_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream3 = nullptr;          // copy stream (adp_memcpy_h2d_async)
    hipEvent_t ev_copy[16] = {};            // adp_copy_mark / adp_copy_wait
    hipStream_t stream2 = nullptr;          // side stream: the start-peak scan (HBM-bound) beside the float64 gains (ALU-bound)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_conv[3] = {};             // conv stack: layer 2 done (main -> side), layer 3 of an even / odd chunk done (side -> main)
    adp_cfg cfg;
    int max_reads = 0, m = 0;
    // geometry of the LLR path
    int T = 0, off = 0, ds = 1, L = 0, Lp = 0, nck = 0, nsum = 0;
    DevBuf mbs, ghist, gbelow, gcnt, cbuf, fz, fcnt, n1heavy, ct_pk, ct_pv, ct_out, gstat, down, nvalid, ck, tail, trace, bmax, bmin, t1, adapter_idx, polya_idx;
    DevBuf bounds, topk_none, rows, preq, series, have_series, vscratch, pk, pkv, npk, mk, st, sp, any_none, sig_stage, len_stage, bounds_stage;
    int vslots = 0, vstride = 0, pslots = 0;
    bool profiling = false;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    int last_n = 0, last_nmb = 0;
    int layout = 0;   // ADP_LAYOUT_*: 1 = the single-read API (pooled from sample 0)
    int oh1 = 5;      // head offset of the first gains pass
    int pos_off = 0;  // added to pooled indices * ds for sample positions
    DevBuf rng0;      // per-read [0, T) ranges of the single-read layout
    // CNN head (cnn_conv.h): weights of the four layers, two activation buffers [chunk][64][Lpad]
    DevBuf cnn_w, cnn_act[2], cnn_x, cnn_sc, ct_st, ct_lnz, ct_ap, cstat, op_arena, op_used, series_plan;
    DevBuf tr_buf, tr_meta; // adp_c_llr_trace: staging of host arrays
    unsigned int op_last_used = 0;
    bool cnn_have_w = false;
    int cnn_Lpad = 0, cnn_L1 = 0, cnn_chunk = 0, n_cu = 256;
    // conv stack: 1 = split float16 MFMA (cnn_conv_split.h, the default), 0 = exact float32 MFMA (cnn_conv.h; ADP_CNN_CONV=f32).
    // cnn_redo_f32: the split kernels met an activation outside the float16 range in this call -- it is being repeated in float32
    int cnn_mode = 1;
    DevBuf cnn_actf[2];                                  // the exact-float32 stack's activations [chunk][64][Lpad] (cnn_act: the split rows)
    int cnn_f_Lpad = 0, cnn_f_L1 = 0, cnn_f_chunk = 0;
    bool cnn_redo_f32 = false;
    DevBuf cnn_wsp;          // split B fragments of layers 1 and 2
    float cnn_sw[4] = {1.f, 1.f, 1.f, 1.f}; // scales of the split weights: layers 1, 2, 3 (folded into layer 2's kernel) and 0 (into layer 1's)
    // grouped execution of the LLR path (llr_grouped): child handles ("lanes") with their own streams and a workspace for ONE
    // group of minibatches; consecutive groups go to alternating lanes so that the phases of neighbouring groups overlap
    adp_handle *lane[ADP_MAX_LANES] = {};
    adp_handle *owner = nullptr;            // a lane's parent: owns the open-pore arena and collects the profile
    DevBuf sphead;                          // K1 state between k_sp_head, the pooling pass and k_sp_tail
    DevBuf mbstat, mbparams;                // parent: minibatch status / N1 parameters of a grouped call
    std::vector<hipEvent_t> ev_sync;        // parent: phase-done events of the groups (no timing)
    hipEvent_t ev_start = nullptr;          // parent: inputs staged, arena counter reset
    bool last_grouped = false;
    // launch attributes already requested through this handle (hipFuncSetAttribute per kernel instantiation and device; kept per
    // handle -- a handle is used by one thread at a time -- instead of in process-wide statics)
    unsigned attr_done = 0;
    size_t lds_series_set = 0;
};

static int env_int(const char *name, int dflt);

static int geom(adp_handle *h)
{
    const adp_cfg &c = h->cfg;
    if (c.downscale_factor < 1 || c.downscale_factor > 32 || c.sp_downscale_factor < 1 || c.sp_downscale_factor > 64) {
        g_err = "downscale_factor must be in [1, 32]"; return ADP_ERR_UNSUPPORTED;
    }
    if (c.mvs_detect_overwrite && c.mvs_detect_check) {
        // mvs.py:264-266 reads the series at 2 * offset when nothing is found: an IndexError in the reference otherwise
        const int off = c.pA_mean_window > c.pA_var_window ? c.pA_mean_window : c.pA_var_window;
        if (c.search_window <= off || c.search_window < 1) { g_err = "mvs_detect_overwrite needs search_window > max(pA_mean_window, pA_var_window)"; return ADP_ERR_UNSUPPORTED; }
    }
    if (c.polya_cand_k > ADP_MAX_CAND - 1) { g_err = "polya_cand_k too large"; return ADP_ERR_UNSUPPORTED; }
    if (h->m > BS_MAXCHUNK * 8192) { g_err = "preload longer than 1 Mi samples: k_partition_stats keeps one chunk sum per 8192 samples in LDS"; return ADP_ERR_UNSUPPORTED; }
    if ((c.max_obs_trace - c.min_obs_adapter) / c.downscale_factor > 400000) { g_err = "max_obs_trace too large for the LDS state of k_polya_peak"; return ADP_ERR_UNSUPPORTED; }
    h->T = c.max_obs_trace < h->m ? c.max_obs_trace : h->m;
    h->off = c.min_obs_adapter;
    h->ds = c.downscale_factor;
    h->oh1 = 5; h->pos_off = h->off;
    if (h->layout == ADP_LAYOUT_SINGLE_READ) { // combined_detect_llr: no min_obs_adapter slice, a longer head offset instead,
        h->off = 0;                             // and min_obs_adapter still added to the positions (combined.py:66-68, 93-95)
        h->oh1 = 5 + c.min_obs_adapter / c.downscale_factor;
        h->pos_off = c.min_obs_adapter;
    }
    int Lseg = h->T - h->off;
    h->L = Lseg > 0 ? (Lseg + h->ds - 1) / h->ds : 0;
    h->Lp = ((h->L + 63) / 64) * 64;
    if (h->Lp == 0) h->Lp = 64;
    h->nck = h->Lp / CK;
    h->nsum = h->Lp / SUMBLK;
    return 0;
}

// Workspace for a call over `reads` reads (grow-only; allocated by the call that needs it, not by adp_create: a grouped LLR
// call keeps its big buffers in the lanes, sized for one group).  llr: the pooled signal / trace / peak-list buffers too.
static int alloc_all(adp_handle *h, int reads, bool llr)
{
    const size_t R = (size_t)(reads < 1 ? 1 : reads), Lp = (size_t)h->Lp;
    int bad = 0;
    if (llr) {
    bad |= h->down.ensure(R * Lp * 4);
    bad |= h->gstat.ensure(R * 24);
    bad |= h->ck.ensure(R * h->nck * sizeof(double2));
    bad |= h->tail.ensure(R * sizeof(double2));
    bad |= h->trace.ensure(R * Lp * 8);
    bad |= h->bmax.ensure(R * h->nsum * 8);
    bad |= h->bmin.ensure(R * h->nsum * 8);
    bad |= h->t1.ensure(R * sizeof(int2));
    }
    bad |= h->nvalid.ensure(R * 4);
    bad |= h->adapter_idx.ensure(R * 4);
    bad |= h->polya_idx.ensure(R * 4);
    bad |= h->bounds.ensure(R * (1 + ADP_MAX_CAND) * 8);
    bad |= h->topk_none.ensure(R);
    bad |= h->rows.ensure(R * sizeof(adp_row));
    bad |= h->preq.ensure(R * sizeof(PartReq));
    bad |= h->series.ensure(R * 2 * MVS_CAP * 4);
    bad |= h->have_series.ensure(R);
    bad |= h->sp.ensure(R * sizeof(SpOut));
    bad |= h->any_none.ensure(64);
    h->vstride = ((h->m + 63) / 64) * 64;
    // k_validate: 6 waves per SIMD = 6144 resident waves; each slot owns two series of up to m floats (moving mean and
    // variance of slices k_mvs_series did not prepare): the slot count is bounded so that this scratch stays below 10 GB
    {
        size_t cap = (size_t)10000000000ull / ((size_t)8 * h->vstride);
        if (cap < 256) cap = 256;
        size_t want = R < 1024 * VAL_WPE ? R : 1024 * VAL_WPE;
        h->vslots = (int)(want < cap ? want : cap);
    }
    bad |= h->vscratch.ensure((size_t)h->vslots * 2 * h->vstride * 4);
    h->pslots = (int)(R < 8192 ? R : 8192);
    if (llr) {
    bad |= h->pk.ensure(R * (Lp / 2 + 1) * 4);   // per-read peak lists (k_gains -> k_polya_peak)
    if (env_int("ADP_PK_VALUES", 1)) bad |= h->pkv.ensure(R * (Lp / 2 + 1) * 8);  // ... and the maxima's heights (twice the list's size: only when they are used)
    bad |= h->npk.ensure(R * 4);
    bad |= h->mk.ensure((size_t)h->pslots * (Lp / 2 + 1) * 4);
    }
    if (bad) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    return 0;
}

// grid = (ceil(m / 1024), n); block = 256.  Read r's samples lie at packed[offs[r] .. offs[r] + min(full_len, m)).
template <class SRC>
__global__ void __launch_bounds__(256) k_expand_ragged(const SRC *__restrict__ packed, const int64_t *__restrict__ offs,
                                                        const int32_t *__restrict__ full_len, const float *__restrict__ scale,
                                                        const float *__restrict__ offset, int m, float *__restrict__ out)
{
    const int r = blockIdx.y;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= m) return;
    const int len = full_len[r] < m ? (full_len[r] > 0 ? full_len[r] : 0) : m;
    const SRC *src = packed + offs[r] + i0;
    float *dst = out + (size_t)r * m + i0;
    const float nanv = __builtin_nanf("");
    float sc = 1.f, of = 0.f;
    if (scale) { sc = scale[r]; of = offset[r]; }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (i0 + j >= m) break;
        float v = nanv;
        if (i0 + j < len) v = scale ? sc * ((float)src[j] + of) : (float)src[j];
        dst[j] = v;
    }
}

// the same for raw samples that STAY raw (adp_detect_llr_i16 reads them): int16 [n, m], zeros behind each read
__global__ void __launch_bounds__(256) k_expand_ragged_raw(const int16_t *__restrict__ packed, const int64_t *__restrict__ offs,
                                                           const int32_t *__restrict__ full_len, int m, int16_t *__restrict__ out)
{
    const int r = blockIdx.y;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= m) return;
    const int len = full_len[r] < m ? (full_len[r] > 0 ? full_len[r] : 0) : m;
    const int16_t *src = packed + offs[r] + i0;
    int16_t *dst = out + (size_t)r * m + i0;
#pragma unroll
    for (int j = 0; j < 4; j++) if (i0 + j < m) dst[j] = (i0 + j < len) ? src[j] : (int16_t)0;
}

extern "C" {

int adp_abi_version(void) { return ADP_ABI_VERSION; }
int adp_sizeof_cfg(void) { return (int)sizeof(adp_cfg); }
int adp_sizeof_row(void) { return (int)sizeof(adp_row); }
const char *adp_last_error(void) { return g_err.c_str(); }

int adp_device_count(void)
{
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    return n;
}

int adp_create(int device, const adp_cfg *cfg, int max_reads, int m, adp_handle **out)
{
    if (!cfg || !out || max_reads < 1 || m < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    adp_handle *h = new adp_handle();
    h->device = device;
    h->cfg = *cfg;
    h->max_reads = max_reads;
    h->m = m;
    int rc = geom(h);
    if (rc) { delete h; return rc; }
    if (hipStreamCreate(&h->stream) != hipSuccess || hipStreamCreate(&h->stream2) != hipSuccess ||
        hipStreamCreate(&h->stream3) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) { delete h; g_err = "hipStreamCreate failed"; return ADP_ERR_HIP; }
    if (hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming) != hipSuccess) { adp_destroy(h); g_err = "hipEventCreate failed"; return ADP_ERR_HIP; }
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) h->n_cu = pr.multiProcessorCount; }
    { const char *cv = getenv("ADP_CNN_CONV"); if (cv && !strcmp(cv, "f32")) h->cnn_mode = 0; } // the exact-float32 conv stack (cnn_conv.h)
    *out = h;
    return ADP_OK;
}

int adp_destroy(adp_handle *h)
{
    if (!h) return ADP_OK;
    (void)hipSetDevice(h->device);
    for (int i = 0; i < ADP_MAX_LANES; i++) if (h->lane[i]) { adp_destroy(h->lane[i]); h->lane[i] = nullptr; }
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipEvent_t e : h->ev_sync) (void)hipEventDestroy(e);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    h->mbstat.release(); h->mbparams.release(); h->sphead.release();
    DevBuf *all[] = {&h->cnn_actf[0], &h->cnn_actf[1], &h->series_plan, &h->cnn_wsp, &h->tr_buf, &h->tr_meta, &h->op_arena, &h->op_used, &h->cstat, &h->cnn_w, &h->cnn_act[0], &h->cnn_act[1], &h->cnn_x, &h->cnn_sc, &h->ct_st, &h->ct_lnz, &h->ct_ap, &h->rng0, &h->mbs, &h->ghist, &h->gbelow, &h->gcnt, &h->cbuf, &h->fz, &h->fcnt, &h->n1heavy, &h->ct_pk, &h->ct_pv, &h->ct_out, &h->gstat, &h->down, &h->nvalid, &h->ck, &h->tail, &h->trace, &h->bmax, &h->bmin,
                     &h->t1, &h->adapter_idx, &h->polya_idx, &h->bounds, &h->topk_none, &h->rows, &h->preq, &h->series, &h->have_series, &h->vscratch, &h->pk, &h->pkv, &h->npk,
                     &h->mk, &h->st, &h->sp, &h->any_none, &h->sig_stage, &h->len_stage, &h->bounds_stage};
    for (DevBuf *b : all) b->release();
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->stream3) { (void)hipStreamSynchronize(h->stream3); (void)hipStreamDestroy(h->stream3); }
    for (int i = 0; i < 16; i++) if (h->ev_copy[i]) (void)hipEventDestroy(h->ev_copy[i]);
    for (int i = 0; i < 3; i++) if (h->ev_conv[i]) (void)hipEventDestroy(h->ev_conv[i]);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return ADP_OK;
}

int adp_set_config(adp_handle *h, const adp_cfg *cfg)
{
    if (!h || !cfg) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->cfg = *cfg;
    int rc = geom(h);
    if (rc) return rc;
    for (int i = 0; i < ADP_MAX_LANES; i++) if (h->lane[i]) { h->lane[i]->cfg = *cfg; (void)geom(h->lane[i]); }
    return ADP_OK;
}

void *adp_stream(adp_handle *h) { return h ? (void *)h->stream : nullptr; }

int adp_synchronize(adp_handle *h)
{
    if (!h) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ADP_OK;
}

int adp_set_profiling(adp_handle *h, int on)
{
    if (!h) return ADP_ERR_INVALID;
    h->profiling = on != 0;
    return ADP_OK;
}

} // extern "C"

// ---- profiling helpers -------------------------------------------------------------
static hipEvent_t next_event(adp_handle *h)
{
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_used++];
}
struct Scope {
    adp_handle *h; hipEvent_t b = nullptr; hipStream_t st;
    Scope(adp_handle *h_, const char *name, hipStream_t st_ = nullptr) : h(h_), st(st_ ? st_ : h_->stream)
    {
        if (!h->profiling || !name) return;
        hipEvent_t a = next_event(h);
        b = next_event(h);
        (void)hipEventRecord(a, st);
        h->prof.push_back({name, a, b});
    }
    ~Scope() { if (b) (void)hipEventRecord(b, st); }
};

// ---- input staging -----------------------------------------------------------------
static int stage_inputs(adp_handle *h, const float *signals, const int32_t *full_len, int n, int m, int flags,
                        const float **dsig, const int32_t **dlen)
{
    if (flags & ADP_IN_DEVICE) { *dsig = signals; *dlen = full_len; return 0; }
    if (h->sig_stage.ensure((size_t)n * m * 4) || h->len_stage.ensure((size_t)n * 4)) { g_err = "staging allocation failed"; return ADP_ERR_HIP; }
    HIPCHK(hipMemcpyAsync(h->sig_stage.p, signals, (size_t)n * m * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->len_stage.p, full_len, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    *dsig = h->sig_stage.as<float>();
    *dlen = h->len_stage.as<int32_t>();
    return 0;
}

static int deliver_rows(adp_handle *h, int n, int flags, adp_row *rows_out)
{
    if (!rows_out) return 0;
    HIPCHK(hipMemcpyAsync(rows_out, h->rows.p, (size_t)n * sizeof(adp_row),
                          (flags & ADP_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    return 0;
}

__global__ void k_mb_set_status(MbState *mbs, int n_mb, int status)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_mb) mbs[i].status = status;
}

// a read without any valid pooled block sinks its minibatch (the reference raises there)
__global__ void k_check_empty(const int32_t *nvalid, int n_reads, int mbsize, MbState *mbs)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    if (nvalid[r] <= 0 && mbs[r / mbsize].status == ADP_MB_OK) mbs[r / mbsize].status = ADP_MB_EMPTY_TRACE;
}

__global__ void k_mb_status_out(const MbState *mbs, int n_mb, int32_t *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_mb) out[i] = mbs[i].status;
}

__global__ void k_mb_params_out(const MbState *mbs, int n_mb, double *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_mb) {
        out[4 * i] = (double)mbs[i].med; out[4 * i + 1] = (double)mbs[i].mad;
        out[4 * i + 2] = (double)mbs[i].lo; out[4 * i + 3] = (double)mbs[i].hi;
    }
}

// developer switch ADP_ABLATE (bit mask that skips parts of kernels in timing experiments; results are wrong when set):
// the device copy is refreshed only when the environment value changes
static void sync_ablate(hipStream_t st)
{
    static int current = 0;
    const char *ab = getenv("ADP_ABLATE");
    const int want = ab ? atoi(ab) : 0;
    if (want != current) {
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ablate), &want, sizeof(int), 0, hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
        current = want;
    }
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// ---- the open-pore arena of one API call --------------------------------------------------
// arena_begin: allocated, counter zeroed (on the handle's stream, ahead of every kernel of the call).  arena_end (the call's
// stream(s) drained): how much the call wanted; > capacity = lists were dropped: grow and tell the caller to run again.
static int arena_begin(adp_handle *h)
{
    if (h->op_used.ensure(8) || (h->op_arena.cap == 0 && h->op_arena.ensure((size_t)65536 * 4))) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    HIPCHK(hipMemsetAsync(h->op_used.p, 0, 8, h->stream)); // [0] arena words wanted, [1] the split conv stack's out-of-range flag
    return 0;
}
// -> 0 done, 1 run the call again (arena grown), < 0 error
// (cnn: the call ran the conv stack -- its out-of-range flag is read with the counter; set = repeat the call on the float32 kernels)
static int arena_end(adp_handle *h, bool cnn = false)
{
    const bool conv_flag = cnn && h->cnn_mode == 1 && !h->cnn_redo_f32;
    if (!h->cfg.detect_open_pores && !conv_flag) { h->op_last_used = 0; return 0; }
    unsigned int w[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(w, h->op_used.p, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (conv_flag && w[1]) { h->cnn_redo_f32 = true; return 1; }
    if (!h->cfg.detect_open_pores) { h->op_last_used = 0; return 0; }
    const unsigned int used = w[0];
    h->op_last_used = used;
    if ((size_t)used * 4 <= h->op_arena.cap) return 0;
    if (h->op_arena.ensure((size_t)used * 8)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    return 1;
}

template <int THREADS, int HB, int U>
static int launch_cand_stats2(adp_handle *h, const float *sig, const int32_t *dlen, int n, int m, int kmax, int cap)
{
    typedef Cs2Sh<HB> Sh;
    const unsigned bit = THREADS >= 512 ? 4096u : 8192u;
    if (!(h->attr_done & bit)) { HIPCHK(hipFuncSetAttribute((const void *)k_cand_stats2<THREADS, HB, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Sh))); h->attr_done |= bit; }
    hipLaunchKernelGGL((k_cand_stats2<THREADS, HB, U>), dim3(n), dim3(THREADS), sizeof(Sh), h->stream, sig, dlen, n, m, h->bounds.as<int64_t>(), kmax, h->cfg,
                       (const float *)h->series.as<float>(), cap, (const int8_t *)h->have_series.as<int8_t>(), h->cstat.as<CandStat>());
    return 0;
}

template <class SIG>
static int launch_validate(adp_handle *h, SIG dsig, const int32_t *dlen, int n, int m, int kmax, int mbsize,
                           bool gate_mb)
{
    ValidateInT<SIG> in;
    in.sig = dsig; in.full_len = dlen; in.bounds = h->bounds.as<int64_t>(); in.topk_none = h->topk_none.as<int8_t>();
    in.kmax = kmax; in.n_reads = n; in.m = m; in.mbsize = mbsize;
    in.mbs = gate_mb ? h->mbs.as<MbState>() : nullptr;
    in.scratch = h->vscratch.as<float>(); in.scratch_stride = h->vstride;
    // several candidates per read (the CNN path): moving-window series up to the LARGEST candidate for every read (the
    // window, not MVS_CAP, bounds them), then the order statistics of all candidates in shared sweeps (cand_stats.h)
    // (two shapes of k_cand_stats: big workgroups and wide levels beyond a 32 k preload, small ones below: cand_stats.h)
    const bool multi = std::is_same<SIG, SigF32>::value && kmax > 1 && h->cfg.mvs_detect_check && !h->cfg.mvs_detect_overwrite &&
                       h->cfg.pA_var_window <= MS_HIST && h->cfg.pA_mean_window <= MS_HIST;
    const int cap = multi ? h->vstride : MVS_CAP;
    if (multi && (h->series.ensure((size_t)n * 2 * cap * 4) || h->cstat.ensure((size_t)n * kmax * sizeof(CandStat)))) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    in.series = h->series.as<float>(); in.have_series = h->have_series.as<int8_t>(); in.series_cap = cap;
    in.cstat = multi ? h->cstat.as<CandStat>() : nullptr;
    if (h->cfg.mvs_detect_check && !h->cfg.mvs_detect_overwrite) {
        // one candidate per read (the LLR path): the same pipeline of waves over the slice [adapter end, poly(A) end) where it takes the
        // windows (round 4; the lane-per-read kernel k_mvs_series waits on 64 scattered rows per load instruction); ADP_SERIES_PIPE_LLR=0: as before
        bool piped = false;
        if constexpr (std::is_same<SIG, SigF32>::value) {
        const bool pipe_ok = sp_takes(h->cfg.pA_var_window, h->cfg.pA_mean_window) && env_int("ADP_SERIES_PIPE", 1);
        if (multi || (pipe_ok && kmax == 1 && env_int("ADP_SERIES_PIPE_LLR", 1))) {
            piped = true;
            Scope s(h, multi ? "k_mvs_series_wave" : "k_mvs_series");
            auto ring = [](int w) { int rb = 128; while (rb < w + MS_CHUNK) rb <<= 1; return rb; };
            const size_t lds = (size_t)MS_G * (ring(h->cfg.pA_var_window) + 4 + ring(h->cfg.pA_mean_window) + 4 + 4 * (MS_CHUNK + 4)) * 4; // (two out halves per wave)
            if (lds > h->lds_series_set) { HIPCHK(hipFuncSetAttribute((const void *)k_mvs_series_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); h->lds_series_set = lds; }
            // the plan (slice starts / lengths, have[]) and the order by falling length (validate.h), then the chains
            if (h->series_plan.ensure((size_t)n * 12 + 2 * MS_NBKT * 4)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
            int32_t *pa = h->series_plan.as<int32_t>(), *pn = pa + n, *pp = pn + n;
            uint32_t *pc = reinterpret_cast<uint32_t *>(pp + n);
            HIPCHK(hipMemsetAsync(pc, 0, 2 * MS_NBKT * 4, h->stream));
            hipLaunchKernelGGL(k_series_plan, dim3((n + 255) / 256), dim3(256), 0, h->stream, dlen, n, m, h->bounds.as<int64_t>(), kmax, h->cfg, cap,
                               h->have_series.as<int8_t>(), pa, pn, pc);
            hipLaunchKernelGGL(k_series_order, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, cap, pn, pc, pc + MS_NBKT, pp);
            // round 4: the recurrences as a pipeline of waves (series_pipe.h) for the windows it takes; ADP_SERIES_PIPE=0: one wave per recurrence
            if (pipe_ok) {
                if (!(h->attr_done & 32u)) { HIPCHK(hipFuncSetAttribute((const void *)k_mvs_series_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS_FLOATS * 4)); h->attr_done |= 32u; }
                hipLaunchKernelGGL(k_mvs_series_pipe, dim3((n + SP_G - 1) / SP_G), dim3(SP_THREADS), SP_LDS_FLOATS * 4, h->stream, dsig.base, n, m, pa, pn, pp,
                                   h->cfg.pA_var_window, h->cfg.pA_mean_window, h->series.as<float>(), cap, h->have_series.as<int8_t>());
            } else
            hipLaunchKernelGGL(k_mvs_series_wave, dim3((n + MS_G - 1) / MS_G), dim3(128), lds, h->stream, dsig.base, n, m, pa, pn, pp, h->cfg,
                               h->series.as<float>(), cap, h->have_series.as<int8_t>());
        }
        }
        if (!piped) {
            Scope s(h, "k_mvs_series");
            hipLaunchKernelGGL(k_mvs_series<SIG>, dim3((n + 63) / 64), dim3(64), 0, h->stream, dsig, dlen, n, m, h->bounds.as<int64_t>(), kmax, h->cfg,
                               h->series.as<float>(), cap, h->have_series.as<int8_t>());
        }
        if constexpr (std::is_same<SIG, SigF32>::value) {
        if (multi) {
            Scope s(h, "k_cand_stats");
            // two sweeps per array (cand_stats2.h)
            int rc;
                // (shapes tried on one box, 24 000 reads at the 200 k window / 32 000 at the default one: 512 threads x 8 loads in flight 14.3 ms,
                // x 4 14.7, 1024 threads 18.9-19.8; 256 threads x 4 3.5 ms, x 8 4.4, x 2 3.45, 128 threads 4.1-4.2)
                rc = h->m > 32768 ? launch_cand_stats2<512, 12, 8>(h, dsig.base, dlen, n, m, kmax, cap)
                                  : launch_cand_stats2<256, 10, 4>(h, dsig.base, dlen, n, m, kmax, cap);
            if (rc) return rc;
        }
        }
    } else {
        (void)hipMemsetAsync(h->have_series.p, 0, (size_t)n, h->stream);
    }
    int grid = n < h->vslots ? n : h->vslots;
    sync_ablate(h->stream);
    // open-pore lists longer than a row holds go to the arena of the CALL (arena_begin / arena_end: cumulative over every
    // launch of the call, shared by the lanes of a grouped call; a call that overflowed it is repeated on a larger one)
    adp_handle *a = h->owner ? h->owner : h;
    in.op_arena = a->op_arena.as<int32_t>(); in.op_used = a->op_used.as<unsigned int>(); in.op_cap = (unsigned int)(a->op_arena.cap / 4);
    { Scope s(h, "k_validate");
      hipLaunchKernelGGL(k_validate<SIG>, dim3(grid), dim3(64), 0, h->stream, in, h->cfg, h->rows.as<adp_row>(), h->preq.as<PartReq>()); }
    { Scope s(h, "k_partition_stats");
      // (the kernel is a template on the workgroup size: 512 threads x 2 and 1024 x 1 per CU, and the second pass walking a segment from its
      // end, were measured and dropped in round 5 -- 28.2 / 38.5 against 24.5 ms; +-1 %: profiles/r05_tried_and_dropped.txt)
      hipLaunchKernelGGL((k_partition_stats<SIG, BS_THREADS, 5>), dim3(n), dim3(BS_THREADS), sizeof(BlockScratch), h->stream, dsig, m, h->preq.as<PartReq>(),
                         h->rows.as<adp_row>()); }
    return 0;
}

// N1 for all minibatches (n1_select.h): sampled guess, then ONE verified full pass per statistic when the copied
// bracket holds the rank (k_n1_finish), else the second pass; a missed window falls back to the aligned path.
template <class SIG>
static int launch_n1(adp_handle *h, SIG dsig, int n, int m, int T, int minibatch, int n_mb, bool profile,
                     const int32_t *tails = nullptr) // tails: full_len when ADP_TAILS_NAN holds
{
    hipStream_t st = h->stream;
    MbState *mbs = h->mbs.as<MbState>();
    uint32_t *gh = h->ghist.as<uint32_t>(), *gb = h->gbelow.as<uint32_t>();
    unsigned long long *gc = h->gcnt.as<unsigned long long>();
    const int collect = ((long long)minibatch * T >= (1ll << 24)) ? 1 : 0; // only worth it (and sized) for big minibatches
    if (collect && h->cbuf.ensure((size_t)n_mb * N1_CB_CAP * 4)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    uint32_t *cb = h->cbuf.as<uint32_t>();
    int bpm = 4096 / n_mb; if (bpm > 256) bpm = 256; if (bpm < 4) bpm = 4;
    dim3 hg(bpm, n_mb), pg(n_mb);
    // sample: 1/32 of the minibatch -- of EVERY read `col_div` pieces of ~96 samples, one in each col_div-th of the window
    // at a position that rotates from read to read (short windows: every 32nd read whole)
    const int pdiv = 32;
    int col_div = 1;
    if (T >= 8192) { col_div = T / (pdiv * 96); if (col_div > 64) col_div = 64; if (col_div < 2) col_div = 2; }
    int row_step = (col_div > 1) ? 1 : minibatch / 32; if (row_step < 1) row_step = 1;
    int sb = (minibatch + row_step - 1) / row_step; if (sb > bpm) sb = bpm; if (sb < 1) sb = 1;
    dim3 sg(sb, n_mb);
    const double thr = h->cfg.sig_norm_outlier_thresh;
    // big minibatches: both statistics from ONE pass (n1_fused.h); whatever it cannot verify is left to the passes below
    // (2^22 samples: the reference's defaults -- 1000 reads per minibatch at the preset's 16 000-sample window, 1.6e7 -- lie just
    // under 2^24, where this threshold stood until the end of round 3: the preset ran the three-pass path, 5.3 instead of 3.0 ms per
    // 96 000 reads.  A bracket that misses on a small minibatch costs the fused pass and falls through to those passes.)
    const long long fused_min = getenv("ADP_N1_FUSED_MIN") ? atoll(getenv("ADP_N1_FUSED_MIN")) : (1ll << 22);
    if ((long long)minibatch * T >= fused_min && T >= 64) {
        if (h->cbuf.ensure((size_t)n_mb * N1_CB_CAP * 4) || h->fz.ensure((size_t)n_mb * sizeof(N1Fused)) ||
            h->fcnt.ensure((size_t)n_mb * 8 * N1F_NCNT) || h->n1heavy.ensure((size_t)n_mb * N1H_WORDS * 4)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
        cb = h->cbuf.as<uint32_t>();
        N1Fused *fz = h->fz.as<N1Fused>();
        unsigned long long *fc = h->fcnt.as<unsigned long long>();
        HIPCHK(hipMemsetAsync(fc, 0, (size_t)n_mb * 8 * N1F_NCNT, st));
        uint32_t *hvy = h->n1heavy.as<uint32_t>();
        HIPCHK(hipMemsetAsync(hvy, 0, (size_t)n_mb * N1H_WORDS * 4, st));
        { Scope s(h, !profile ? nullptr : "k_n1 sample passes");
        // the first level of a statistic's sample only picks the 2^21-key window the second level works in: an eighth of the sampled
        // rows does (ADP_N1_S0: 1 = every sampled row, as before); the second level, whose counts place the bracket, keeps them all
        const int s0 = (col_div > 1 && minibatch >= 64) ? env_int("ADP_N1_S0", 8) : 1;
        for (int mode = 0; mode < 2; mode++) {
            hipLaunchKernelGGL((k_n1_hist<0, SIG>), sg, dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mode, mbs, gh, gb, gc, row_step * (s0 > 1 ? s0 : 1), N1_ALWAYS, cb, 0, col_div, pdiv, tails);
            hipLaunchKernelGGL((k_n1_pick<0, N1_SAMPLE>), pg, dim3(256), 0, st, mbs, gh, gb, gc, mode, thr);
            hipLaunchKernelGGL((k_n1_hist<1, SIG>), sg, dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mode, mbs, gh, gb, gc, row_step, N1_ALWAYS, cb, 0, col_div, pdiv, tails);
            hipLaunchKernelGGL((k_n1_pick<1, N1_SAMPLE>), pg, dim3(256), 0, st, mbs, gh, gb, gc, mode, thr);
            hipLaunchKernelGGL(k_n1_fuse_setup, dim3((n_mb + 63) / 64), dim3(64), 0, st, mbs, fz, n_mb, mode);
        }
        // values shared by many of the samples to copy (quantised data): found in the sample, counted instead of copied
        // (a quarter of the sampled rows is plenty to see ties)
        hipLaunchKernelGGL(k_n1_heavy_scan<SIG>, dim3((sb + 3) / 4, n_mb), dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mbs, fz, hvy, row_step * 4, col_div, pdiv, tails);
        hipLaunchKernelGGL(k_n1_heavy_pick, pg, dim3(64), 0, st, mbs, fz, hvy, n_mb); }
        { Scope s(h, !profile ? nullptr : "k_n1_fused");
          hipLaunchKernelGGL(k_n1_fused<SIG>, hg, dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mbs, fz, fc, (float *)cb, hvy, tails); }
        { Scope s(h, !profile ? nullptr : "k_n1_fused_finish");
          hipLaunchKernelGGL(k_n1_fused_finish, pg, dim3(1024), 0, st, mbs, fz, fc, (const float *)cb, thr, (const uint32_t *)hvy); }
    }
    // behind the fused pass these launches only serve the (rare) minibatch it could not settle: with many minibatches in the call
    // such a minibatch would be left to 4096 / n_mb blocks -- one missed bracket among 96 Pareto-length minibatches cost 3.5 ms of a
    // 50 ms step -- so it gets at least 128 (the blocks of settled minibatches return at once)
    if ((long long)minibatch * T >= fused_min && T >= 64 && bpm < 128) hg = dim3(128, n_mb);
    for (int mode = 0; mode < 2; mode++) {
        // guess from a row sample
        hipLaunchKernelGGL((k_n1_hist<0, SIG>), sg, dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mode, mbs, gh, gb, gc, row_step, N1_ALWAYS, cb, 0, col_div, pdiv, tails);
        hipLaunchKernelGGL((k_n1_pick<0, N1_SAMPLE>), pg, dim3(256), 0, st, mbs, gh, gb, gc, mode, thr);
        hipLaunchKernelGGL((k_n1_hist<1, SIG>), sg, dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mode, mbs, gh, gb, gc, row_step, N1_ALWAYS, cb, 0, col_div, pdiv, tails);
        hipLaunchKernelGGL((k_n1_pick<1, N1_SAMPLE>), pg, dim3(256), 0, st, mbs, gh, gb, gc, mode, thr);
        // pass 1 over everything, verified
        { Scope s(h, !profile ? nullptr : (mode ? "k_n1_hist<1> mad" : "k_n1_hist<1> med"));
          hipLaunchKernelGGL((k_n1_hist<1, SIG>), hg, dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mode, mbs, gh, gb, gc, 1, N1_ALWAYS, cb, collect, 1, 8, tails); }
        hipLaunchKernelGGL((k_n1_pick<1, N1_FULL>), pg, dim3(256), 0, st, mbs, gh, gb, gc, mode, thr);
        hipLaunchKernelGGL(k_n1_finish, pg, dim3(1024), 0, st, mbs, cb, gc, gb, mode, thr);
        // fallback (runs only for minibatches whose guess missed)
        hipLaunchKernelGGL((k_n1_hist<0, SIG>), hg, dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mode, mbs, gh, gb, gc, 1, N1_IF_BAD, cb, 0, 1, 8, tails);
        hipLaunchKernelGGL((k_n1_pick<0, N1_FALLBACK>), pg, dim3(256), 0, st, mbs, gh, gb, gc, mode, thr);
        hipLaunchKernelGGL((k_n1_hist<1, SIG>), hg, dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mode, mbs, gh, gb, gc, 1, N1_IF_BAD, cb, 0, 1, 8, tails);
        hipLaunchKernelGGL((k_n1_pick<1, N1_FALLBACK>), pg, dim3(256), 0, st, mbs, gh, gb, gc, mode, thr);
        // pass 2 (only where the bracket did not settle it)
        { Scope s(h, !profile ? nullptr : (mode ? "k_n1_hist<2> mad" : "k_n1_hist<2> med"));
          hipLaunchKernelGGL((k_n1_hist<2, SIG>), hg, dim3(N1_THREADS), 0, st, dsig, n, m, T, minibatch, mode, mbs, gh, gb, gc, 1, N1_IF_NOT_DONE, cb, 0, 1, 8, tails); }
        hipLaunchKernelGGL((k_n1_pick<2, N1_FULL>), pg, dim3(256), 0, st, mbs, gh, gb, gc, mode, thr);
    }
    return 0;
}

// ---- the LLR path over a resident signal matrix: float32 pA (SigF32) or int16 ADC + calibration (SigI16) ----------------
// Three phases per group of minibatches: S (N1 + normalised pooling: streams the signal twice), C (cumulative sums, both
// gains passes, peak picking: float64 ALU / latency, touches only the pooled signal), V (validation + partition statistics:
// streams the signal twice more).  llr_enqueue puts one group on a handle's stream WITHOUT waiting for anything on the host;
// PhaseSync lets the caller order the phases of different groups against each other across streams.
struct PhaseSync {
    hipEvent_t wait[3] = {nullptr, nullptr, nullptr};  // before phase S / C / V: wait for this event (another stream's)
    hipEvent_t done[3] = {nullptr, nullptr, nullptr};  // after phase S / C / V: record this event
};

static __device__ __host__ inline SigF32 sig_from(const SigF32 &s, size_t r0, int m) { return SigF32{s.base + r0 * (size_t)m}; }
static __device__ __host__ inline SigI16 sig_from(const SigI16 &s, size_t r0, int m) { return SigI16{s.base + r0 * (size_t)m, s.scale + r0, s.offset + r0, s.full_len + r0}; }

template <class SIG>
static int llr_enqueue(adp_handle *h, SIG dsig, const int32_t *dlen, int n, int m, int minibatch, int flags,
                       adp_row *rows_dev, int rows_kind /* hipMemcpyKind of the row delivery */, int32_t *mb_status_dev, double *mb_params_dev,
                       int upto, const PhaseSync *ps)
{
    int rc = 0;
    const int n_mb = (n + minibatch - 1) / minibatch;
    h->last_n = n; h->last_nmb = n_mb;
    if (h->mbs.ensure((size_t)n_mb * sizeof(MbState)) || h->ghist.ensure((size_t)n_mb * N1_BINS * 4) || h->gbelow.ensure((size_t)n_mb * 8) || h->gcnt.ensure((size_t)n_mb * 8 * N1_NCNT)) {
        g_err = "device allocation failed"; return ADP_ERR_HIP;
    }
    rc = alloc_all(h, n, true);
    if (rc) return rc;
    hipStream_t st = h->stream;
    sync_ablate(st);
    if (ps && ps->wait[0]) HIPCHK(hipStreamWaitEvent(st, ps->wait[0], 0));
    MbState *mbs = h->mbs.as<MbState>();
    HIPCHK(hipMemsetAsync(mbs, 0, (size_t)n_mb * sizeof(MbState), st));
    HIPCHK(hipMemsetAsync(h->ghist.p, 0, (size_t)n_mb * N1_BINS * 4, st));
    HIPCHK(hipMemsetAsync(h->gbelow.p, 0, (size_t)n_mb * 8, st));
    HIPCHK(hipMemsetAsync(h->gcnt.p, 0, (size_t)n_mb * 8 * N1_NCNT, st));
    const int T = h->T;
    bool sp_forked = false, sp_done = false;
    if (h->L <= 0) {
        hipLaunchKernelGGL(k_mb_set_status, dim3((n_mb + 255) / 256), dim3(256), 0, st, mbs, n_mb, ADP_MB_EMPTY_TRACE);
        if (ps) for (int k = 0; k < 2; k++) { if (k && ps->wait[k]) HIPCHK(hipStreamWaitEvent(st, ps->wait[k], 0)); if (ps->done[k]) HIPCHK(hipEventRecord(ps->done[k], st)); }
    } else {
        // ---- phase S
        const int32_t *tails = (flags & ADP_TAILS_NAN) ? dlen : nullptr;
        rc = launch_n1(h, dsig, n, m, T, minibatch, n_mb, true, tails);
        if (rc) return rc;
        // K1 riding the pooling pass (k_norm_pool<SIG, true>): when both pooling factors agree and min_obs_adapter is a multiple of
        // them the start-peak scan costs no sweep of its own (ADP_SP_FUSED=0: the separate k_start_peak on the side stream)
        const bool sp_fused = upto >= 8 && (flags & ADP_WITH_START_PEAK) && h->layout == ADP_LAYOUT_MINIBATCH &&
                              h->cfg.sp_downscale_factor == h->ds && h->off % h->ds == 0 && T > h->off && env_int("ADP_SP_FUSED", 1) != 0;
        if (sp_fused) {
            if (h->sphead.ensure((size_t)n * sizeof(SpHead))) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
            Scope s(h, "k_sp_head");
            hipLaunchKernelGGL(k_sp_head<SIG>, dim3(n), dim3(64), (size_t)64 * h->cfg.sp_downscale_factor * 4, st, dsig, dlen, n, m, h->cfg, h->off, h->sphead.as<SpHead>());
        }
        if (upto >= 2 && sp_fused) {
            { Scope s(h, "k_norm_pool");
              // (float32 rows only: the int16 rows' conversion pushes the prefetching variant to 112 registers, 4 waves per SIMD -- 24 vs 14 ms)
              auto kern = (h->ds == 10 && std::is_same<SIG, SigF32>::value && env_int("ADP_NP_PREFETCH", 1)) ? k_norm_pool<SIG, true, 5> : k_norm_pool<SIG, true, 0>;
              hipLaunchKernelGGL(kern, dim3(n), dim3(256), (size_t)NP_TILE * h->ds * 4, st, dsig, m, T, h->off, h->ds, h->L, h->Lp,
                                 minibatch, mbs, h->down.as<float>(), h->nvalid.as<int32_t>(), (const int64_t *)nullptr,
                                 dlen, (flags & ADP_TAILS_NAN) ? 1 : 0, h->sphead.as<SpHead>(), (float)h->cfg.open_pore_pa); }
            const int cov0 = h->off / h->ds, cov1 = cov0 + (T - h->off) / h->ds;
            Scope s(h, "k_sp_tail");
            hipLaunchKernelGGL(k_sp_tail<SIG>, dim3(n), dim3(64), (size_t)64 * h->cfg.sp_downscale_factor * 4, st, dsig, dlen, n, m, h->cfg, cov0, cov1,
                               (const SpHead *)h->sphead.as<SpHead>(), h->sp.as<SpOut>());
            sp_done = true;
        } else if (upto >= 2) {
            Scope s(h, "k_norm_pool");
            const int64_t *rng = nullptr;
            if (h->layout == ADP_LAYOUT_SINGLE_READ) {
                // the reference pools the unpadded read: its last, ragged block is filled up with zeros at the READ's end
                // (downscale.py:22-29), which is what the per-read range [0, min(T, full_len)) does here
                if (h->rng0.ensure((size_t)n * 16)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
                std::vector<int64_t> hr((size_t)2 * n);
                for (int i = 0; i < n; i++) { hr[2 * i] = 0; hr[2 * i + 1] = T; }
                HIPCHK(hipMemcpyAsync(h->rng0.p, hr.data(), (size_t)n * 16, hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st)); // (hr goes out of scope; this layout is never grouped)
                rng = h->rng0.as<int64_t>();
            }
            auto kern = (h->ds == 10 && !rng && std::is_same<SIG, SigF32>::value && env_int("ADP_NP_PREFETCH", 1)) ? k_norm_pool<SIG, false, 5> : k_norm_pool<SIG, false, 0>;
            hipLaunchKernelGGL(kern, dim3(n), dim3(256), (size_t)NP_TILE * h->ds * 4, st, dsig, m, T, h->off, h->ds, h->L, h->Lp,
                               minibatch, mbs, h->down.as<float>(), h->nvalid.as<int32_t>(), rng,
                               dlen, (flags & ADP_TAILS_NAN) ? 1 : 0, (SpHead *)nullptr, 0.f);
        }
        if (upto >= 2)
            hipLaunchKernelGGL(k_check_empty, dim3((n + 255) / 256), dim3(256), 0, st, h->nvalid.as<int32_t>(), n, minibatch, mbs);
        if (ps && ps->done[0]) HIPCHK(hipEventRecord(ps->done[0], st));
        if (upto >= 8 && (flags & ADP_WITH_START_PEAK) && !sp_fused) {
            // the start-peak scan depends on nothing computed here: it streams the signal on the side stream while the
            // main stream runs the ALU-bound cumulative sums and gains
            HIPCHK(hipEventRecord(h->ev_fork, st));
            HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
            { Scope s(h, "k_start_peak", h->stream2);
              hipLaunchKernelGGL(k_start_peak<SIG>, dim3(n), dim3(64), (size_t)64 * h->cfg.sp_downscale_factor * 4, h->stream2, dsig, dlen, n, m, h->cfg, h->sp.as<SpOut>()); }
            HIPCHK(hipEventRecord(h->ev_join, h->stream2));
            sp_forked = true;
        }
        // ---- phase C
        if (ps && ps->wait[1]) HIPCHK(hipStreamWaitEvent(st, ps->wait[1], 0));
        if (upto >= 3) {
            Scope s(h, "k_cumsum");
            const int split = env_int("ADP_CUMSUM_GATHER", 1) != 0;
            hipLaunchKernelGGL(k_cumsum, dim3((n + 63) / 64), dim3(64), 0, st, h->down.as<float>(), h->nvalid.as<int32_t>(), h->Lp, n,
                               h->nck, h->ck.as<double2>(), h->tail.as<double2>(), split);
            if (split)
                hipLaunchKernelGGL(k_cumsum_gather, dim3((n + 63) / 64), dim3(64), 0, st, h->down.as<float>(), h->nvalid.as<int32_t>(), h->Lp, n,
                                   h->nck, h->ck.as<double2>(), h->tail.as<double2>());
        }
        if (upto >= 4) {
            Scope s(h, "k_gains<1>");
            hipLaunchKernelGGL(k_gains<1>, dim3((n + GAINS_WPB - 1) / GAINS_WPB), dim3(64 * GAINS_WPB), 0, st, h->down.as<float>(), h->nvalid.as<int32_t>(), h->Lp, h->nck,
                               h->ck.as<double2>(), h->tail.as<double2>(), h->adapter_idx.as<int32_t>(), minibatch, mbs,
                               h->trace.as<double>(), h->bmax.as<double>(), h->bmin.as<double>(), h->nsum, h->t1.as<int2>(), 0, h->pk.as<int32_t>(), h->npk.as<int32_t>(), h->Lp / 2 + 1, h->gstat.as<double>(), n, h->oh1);
        }
        if (upto >= 5) {
            Scope s(h, "k_adapter_peak");
            hipLaunchKernelGGL(k_adapter_peak, dim3(n), dim3(64), 0, st, h->trace.as<double>(), h->nvalid.as<int32_t>(), h->Lp,
                               h->bmax.as<double>(), h->bmin.as<double>(), h->nsum, h->t1.as<int2>(), minibatch, mbs,
                               h->cfg.adapter_peak_prominence, h->cfg.adapter_peak_rel_height,
                               h->cfg.adapter_peak_width / h->ds, h->adapter_idx.as<int32_t>(), h->gstat.as<double>());
        }
        // (ADP_PK_VALUES=0: k_polya_peak gathers the heights from the trace, as before round 4; a handle made without the buffer keeps doing so)
        double *pkvp = (env_int("ADP_PK_VALUES", 1) && h->pkv.cap >= (size_t)n * (h->Lp / 2 + 1) * 8) ? h->pkv.as<double>() : nullptr;
        if (upto >= 6) {
            Scope s(h, "k_gains<2>");
            hipLaunchKernelGGL(k_gains<2>, dim3((n + GAINS_WPB - 1) / GAINS_WPB), dim3(64 * GAINS_WPB), 0, st, h->down.as<float>(), h->nvalid.as<int32_t>(), h->Lp, h->nck,
                               h->ck.as<double2>(), h->tail.as<double2>(), h->adapter_idx.as<int32_t>(), minibatch, mbs,
                               h->trace.as<double>(), h->bmax.as<double>(), h->bmin.as<double>(), h->nsum, h->t1.as<int2>(), 0, h->pk.as<int32_t>(), h->npk.as<int32_t>(), h->Lp / 2 + 1, h->gstat.as<double>(), n, 5, pkvp);
        }
        if (upto >= 7) {
            Scope s(h, "k_polya_peak");
            int grid = n < h->pslots ? n : h->pslots;
            hipLaunchKernelGGL(k_polya_peak, dim3(grid), dim3(64), (size_t)(((h->Lp / 2 + 1) + 8) / 16 + 2) * 4, st, h->trace.as<double>(), h->nvalid.as<int32_t>(), h->Lp,
                               h->bmax.as<double>(), h->bmin.as<double>(), h->nsum, h->adapter_idx.as<int32_t>(), n, minibatch, mbs,
                               h->pk.as<int32_t>(), h->mk.as<uint32_t>(), h->polya_idx.as<int32_t>(), h->npk.as<int32_t>(), pkvp);
        }
        if (ps && ps->done[1]) HIPCHK(hipEventRecord(ps->done[1], st));
    }
    if (upto >= 8) {
        // ---- phase V
        if (ps && ps->wait[2]) HIPCHK(hipStreamWaitEvent(st, ps->wait[2], 0));
        if (h->L > 0)
            hipLaunchKernelGGL(k_llr_bounds, dim3((n + 255) / 256), dim3(256), 0, st, h->adapter_idx.as<int32_t>(),
                               h->polya_idx.as<int32_t>(), n, h->ds, h->pos_off, h->bounds.as<int64_t>(), h->topk_none.as<int8_t>(), h->layout == ADP_LAYOUT_SINGLE_READ ? 1 : 0);
        rc = launch_validate(h, dsig, dlen, n, m, 1, minibatch, true);
        if (rc) return rc;
        if (flags & ADP_WITH_START_PEAK) {
            if (sp_forked) HIPCHK(hipStreamWaitEvent(st, h->ev_join, 0));
            else if (!sp_done) {
                Scope s(h, "k_start_peak");
                hipLaunchKernelGGL(k_start_peak<SIG>, dim3(n), dim3(64), (size_t)64 * h->cfg.sp_downscale_factor * 4, st, dsig, dlen, n, m, h->cfg, h->sp.as<SpOut>());
            }
            hipLaunchKernelGGL(k_sp_decorate, dim3((n + 255) / 256), dim3(256), 0, st, h->sp.as<SpOut>(), h->rows.as<adp_row>(), n, 0,
                               (const int32_t *)nullptr);
        }
        if (rows_dev)
            HIPCHK(hipMemcpyAsync(rows_dev, h->rows.p, (size_t)n * sizeof(adp_row), (hipMemcpyKind)rows_kind, st));
    }
    if (mb_status_dev)
        hipLaunchKernelGGL(k_mb_status_out, dim3((n_mb + 255) / 256), dim3(256), 0, st, mbs, n_mb, mb_status_dev);
    if (mb_params_dev)
        hipLaunchKernelGGL(k_mb_params_out, dim3((n_mb + 255) / 256), dim3(256), 0, st, mbs, n_mb, mb_params_dev);
    if (ps && ps->done[2]) HIPCHK(hipEventRecord(ps->done[2], st));
    HIPCHK(hipGetLastError());
    return ADP_OK;
}

// how a call's minibatches are cut into groups: ADP_GROUPS (unset / 1: one group = the plain serial pipeline, 0: automatic = three
// groups per lane, k: aim at k groups); ADP_LANES (streams the groups rotate over, default 2); ADP_STAGGER (bit p set: phase p of
// group g + 1 starts after phase p of group g; default 1 = the streaming S phases take turns, which keeps neighbouring groups one
// phase apart)

static int lane_get(adp_handle *h, int i, int reads, adp_handle **out)
{
    adp_handle *l = h->lane[i];
    if (!l) {
        int rc = adp_create(h->device, &h->cfg, h->max_reads, h->m, &l);
        if (rc) return rc;
        l->owner = h;
        l->n_cu = h->n_cu;
        h->lane[i] = l;
    }
    l->profiling = h->profiling;
    *out = l;
    (void)reads;
    return 0;
}

template <class SIG>
static int llr_grouped(adp_handle *h, SIG dsig, const int32_t *dlen, int n, int m, int minibatch, int flags, adp_row *rows_out,
                       int32_t *mb_status, int mb_per_group, int n_lanes)
{
    const int n_mb = (n + minibatch - 1) / minibatch;
    const int G = (n_mb + mb_per_group - 1) / mb_per_group;
    const int stagger = env_int("ADP_STAGGER", 1);
    h->last_n = n; h->last_nmb = n_mb; h->last_grouped = true;
    const bool out_dev = (flags & ADP_OUT_DEVICE) != 0;
    if (h->mbstat.ensure((size_t)n_mb * 4) || h->mbparams.ensure((size_t)n_mb * 32) || (rows_out && !out_dev && h->rows.ensure((size_t)n * sizeof(adp_row)))) {
        g_err = "device allocation failed"; return ADP_ERR_HIP;
    }
    while (h->ev_sync.size() < (size_t)G * 3) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->ev_sync.push_back(e);
    }
    adp_handle *lanes[ADP_MAX_LANES];
    for (int i = 0; i < n_lanes; i++) {
        int rc = lane_get(h, i, mb_per_group * minibatch, &lanes[i]);
        if (rc) return rc;
        lanes[i]->prof.clear(); lanes[i]->ev_used = 0;
    }
    adp_row *rows_dev = rows_out ? (out_dev ? rows_out : h->rows.as<adp_row>()) : nullptr;
    for (int attempt = 0; attempt < 3; attempt++) {
        int rc = arena_begin(h);
        if (rc) return rc;
        HIPCHK(hipEventRecord(h->ev_start, h->stream));      // (inputs staged on this stream, arena counter zeroed)
        for (int i = 0; i < n_lanes; i++) HIPCHK(hipStreamWaitEvent(lanes[i]->stream, h->ev_start, 0));
        for (int g = 0; g < G; g++) {
            adp_handle *l = lanes[g % n_lanes];
            const int mb0 = g * mb_per_group, r0 = mb0 * minibatch;
            const int ng = (n - r0) < mb_per_group * minibatch ? (n - r0) : mb_per_group * minibatch;
            PhaseSync ps;
            for (int p = 0; p < 3; p++) {
                ps.done[p] = h->ev_sync[(size_t)g * 3 + p];
                if (g > 0 && (stagger >> p & 1)) ps.wait[p] = h->ev_sync[(size_t)(g - 1) * 3 + p];
            }
            rc = llr_enqueue(l, sig_from(dsig, (size_t)r0, m), dlen + r0, ng, m, minibatch, flags, rows_dev ? rows_dev + r0 : nullptr,
                             hipMemcpyDeviceToDevice, h->mbstat.as<int32_t>() + mb0, h->mbparams.as<double>() + 4 * (size_t)mb0, 8, &ps);
            if (rc) { for (int i = 0; i < n_lanes; i++) (void)hipStreamSynchronize(lanes[i]->stream); return rc; }
        }
        for (int i = 0; i < n_lanes; i++) HIPCHK(hipStreamSynchronize(lanes[i]->stream));
        rc = arena_end(h);
        if (rc < 0) return rc;
        if (rc == 0) break;
        if (attempt == 2) { g_err = "the call's repeats (conv stack out of the float16 range, open-pore arena growth) are used up and the arena is still short"; return ADP_ERR_CAPACITY; }
        for (int i = 0; i < n_lanes; i++) { lanes[i]->prof.clear(); lanes[i]->ev_used = 0; }
    }
    if (rows_out && !out_dev) HIPCHK(hipMemcpyAsync(rows_out, h->rows.p, (size_t)n * sizeof(adp_row), hipMemcpyDeviceToHost, h->stream));
    if (mb_status) HIPCHK(hipMemcpyAsync(mb_status, h->mbstat.p, (size_t)n_mb * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ADP_OK;
}

template <class SIG>
static int llr_pipeline_t(adp_handle *h, SIG dsig, const int32_t *dlen, int n, int m, int minibatch, int flags,
                          adp_row *rows_out, int32_t *mb_status, int upto)
{
    const int n_mb = (n + minibatch - 1) / minibatch;
    // grouped execution is OPT-IN (ADP_GROUPS > 1): measured on MI355X, overlapping the float64 gains of one group with the
    // streaming passes of its neighbour buys nothing (+-1.5 %, profiles/r03_overlap_*: the kernels do run side by side -- two or
    // more resident for 82 % of the busy time -- and stretch each other by the same factor; even a PURE read beside a PURE
    // float64 FMA kernel reaches only 0.81-0.88 of the serial time on this chip), while one launch over all reads of the
    // call has the shortest tails.  What grouping does buy is memory: the lanes' workspace is sized for one group.
    int want = env_int("ADP_GROUPS", 1), n_lanes = env_int("ADP_LANES", 2);
    if (n_lanes < 1) n_lanes = 1;
    if (n_lanes > ADP_MAX_LANES) n_lanes = ADP_MAX_LANES;
    if (want <= 0) want = 3 * n_lanes;
    if (upto == 8 && h->layout == ADP_LAYOUT_MINIBATCH && n_mb >= 2 && want > 1) {
        int per = (n_mb + want - 1) / want;
        if (per < 1) per = 1;
        if ((n_mb + per - 1) / per < n_lanes) n_lanes = (n_mb + per - 1) / per;
        return llr_grouped(h, dsig, dlen, n, m, minibatch, flags, rows_out, mb_status, per, n_lanes);
    }
    h->last_grouped = false;
    for (int attempt = 0; attempt < 3; attempt++) {
        int rc = arena_begin(h);
        if (rc) return rc;
        rc = llr_enqueue(h, dsig, dlen, n, m, minibatch, flags, upto >= 8 ? rows_out : nullptr,
                         (flags & ADP_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, nullptr, nullptr, upto, nullptr);
        if (rc) return rc;
        if (mb_status) {
            // through a small device buffer -> host
            hipLaunchKernelGGL(k_mb_status_out, dim3((n_mb + 255) / 256), dim3(256), 0, h->stream, h->mbs.as<MbState>(), n_mb, h->gbelow.as<int32_t>());
            HIPCHK(hipMemcpyAsync(mb_status, h->gbelow.p, (size_t)n_mb * 4, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        if (upto < 8) break;
        rc = arena_end(h);
        if (rc < 0) return rc;
        if (rc == 0) break;
        if (attempt == 2) { g_err = "the call's repeats (conv stack out of the float16 range, open-pore arena growth) are used up and the arena is still short"; return ADP_ERR_CAPACITY; }
        h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    }
    return ADP_OK;
}

static int llr_pipeline(adp_handle *h, const float *signals, const int32_t *full_len, int n, int m, int minibatch, int flags,
                        adp_row *rows_out, int32_t *mb_status, int upto)
{
    if (!h || !signals || !full_len || n < 1 || minibatch < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    if (n > h->max_reads || m != h->m) { g_err = "n_reads/m exceed the handle's capacity"; return ADP_ERR_CAPACITY; }
    if (h->layout == ADP_LAYOUT_SINGLE_READ && minibatch != 1) { g_err = "the single-read layout normalises every read on its own: minibatch must be 1"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear();
    h->ev_used = 0;
    h->last_grouped = false;
    const float *dsig; const int32_t *dlen;
    int rc = stage_inputs(h, signals, full_len, n, m, flags, &dsig, &dlen);
    if (rc) return rc;
    return llr_pipeline_t(h, SigF32{dsig}, dlen, n, m, minibatch, flags, rows_out, mb_status, upto);
}

extern "C" {

int adp_set_layout(adp_handle *h, int layout)
{
    if (!h || (layout != ADP_LAYOUT_MINIBATCH && layout != ADP_LAYOUT_SINGLE_READ)) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    const int old = h->layout;
    h->layout = layout;
    int rc = geom(h);
    if (rc) { h->layout = old; (void)geom(h); }
    return rc;
}

int adp_detect_llr(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m, int minibatch, int flags,
                   adp_row *rows_out, int32_t *mb_status)
{
    return llr_pipeline(h, signals, full_len, n_reads, m, minibatch, flags, rows_out, mb_status, 8);
}

int adp_detect_llr_i16(adp_handle *h, const int16_t *raw, const int32_t *full_len, const float *scale, const float *offset, int n_reads,
                       int m, int minibatch, int flags, adp_row *rows_out, int32_t *mb_status)
{
    if (!h || !raw || !full_len || !scale || !offset || n_reads < 1 || minibatch < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    if (!(flags & ADP_IN_DEVICE)) { g_err = "adp_detect_llr_i16 takes device pointers (ADP_IN_DEVICE)"; return ADP_ERR_INVALID; }
    if (n_reads > h->max_reads || m != h->m) { g_err = "n_reads/m exceed the handle's capacity"; return ADP_ERR_CAPACITY; }
    if (h->layout == ADP_LAYOUT_SINGLE_READ) { g_err = "the single-read layout takes float32 input"; return ADP_ERR_UNSUPPORTED; }
    if (m & 3) { g_err = "int16 rows need m % 4 == 0 (8-byte aligned rows)"; return ADP_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear();
    h->ev_used = 0;
    h->last_grouped = false;
    // (samples at or beyond min(full_len, m) read as NaN: the padding is implied, so the passes always stop at a read's end)
    return llr_pipeline_t(h, SigI16{raw, scale, offset, full_len}, full_len, n_reads, m, minibatch, flags | ADP_TAILS_NAN, rows_out, mb_status, 8);
}

__global__ void k_debug_log(const double *in, double *out, int n)
{
    __shared__ __attribute__((aligned(16))) double lt_[3 * LOGCR_N];
    for (int i = threadIdx.x; i < 3 * LOGCR_N; i += blockDim.x) lt_[i] = g_logcr_table[i];
    __syncthreads();
    const LDS double *lt = (const LDS double *)lt_;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = log_cr_impl(in[i], lt, [](double u) { return log(u); });
}

// c_llr_trace / c_llr_trace_gains / _gains for a batch of float64 signals (trace_api.h)
int adp_c_llr_trace(adp_handle *h, const double *raw, const int32_t *len, const int32_t *start, const int32_t *end, int n_reads,
                    int L, const adp_trace_args *args, int flags, double *gain_out, double *c_io, double *c2_io)
{
    if (!h || !len || !start || !end || !args || !gain_out || n_reads < 1 || L < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    const bool from_sums = (flags & ADP_TRACE_FROM_SUMS) != 0, in_dev = (flags & ADP_IN_DEVICE) != 0, out_dev = (flags & ADP_OUT_DEVICE) != 0;
    if (from_sums ? (!c_io || !c2_io) : !raw) { g_err = "bad argument: no input signal / sums"; return ADP_ERR_INVALID; }
    if (!from_sums && ((c_io == nullptr) != (c2_io == nullptr))) { g_err = "bad argument: c and c2 go together"; return ADP_ERR_INVALID; }
    const adp_trace_args &a = *args;
    if (a.stride < 1 || a.min_obs < 0 || a.border_trim < 0) { g_err = "stride must be >= 1, offsets >= 0"; return ADP_ERR_INVALID; }
    if (a.polya_early_stopping > 0) {
        if (a.adapter_early_stop_stride < 1 || a.polya_early_stop_stride < 1 || a.adapter_early_stop_window < 0 || a.polya_early_stop_window < 0) { g_err = "early-stop windows / strides out of range"; return ADP_ERR_INVALID; }
        if (a.adapter_early_stop_stride % a.stride || a.polya_early_stop_stride % a.stride) { g_err = "early-stop stride is not a multiple of stride (the reference asserts, _c_llr.pyx:137-138)"; return ADP_ERR_INVALID; }
    } else if (a.adapter_early_stopping > 0) {
        if (a.adapter_early_stop_stride < 1 || a.adapter_early_stop_window < 0) { g_err = "early-stop window / stride out of range"; return ADP_ERR_INVALID; }
        if (a.adapter_early_stop_stride % a.stride) { g_err = "early-stop stride is not a multiple of stride (the reference asserts, _c_llr.pyx:102)"; return ADP_ERR_INVALID; }
    }
    for (int r = 0; r < n_reads; r++)
        if (len[r] < 0 || len[r] > L || start[r] < 0 || start[r] > end[r] || end[r] > len[r]) { g_err = "need 0 <= start <= end <= len <= L for every read"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    const size_t row = (size_t)L * 8, mat = row * n_reads;
    // device staging: [raw | c | c2 | gain] as far as the caller's arrays are host memory
    const int need = (in_dev ? 0 : (from_sums ? 2 : 1)) + ((!from_sums && (!c_io || !out_dev)) ? 2 : 0) + (out_dev ? 0 : 1);
    if (need && h->tr_buf.ensure(mat * need)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    if (h->tr_meta.ensure((size_t)n_reads * 12)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    double *pool = h->tr_buf.as<double>();
    auto take = [&]() { double *p = pool; pool += (size_t)L * n_reads; return p; };
    int32_t *dlen = h->tr_meta.as<int32_t>(), *dstart = dlen + n_reads, *dend = dstart + n_reads;
    HIPCHK(hipMemcpyAsync(dlen, len, (size_t)n_reads * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dstart, start, (size_t)n_reads * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dend, end, (size_t)n_reads * 4, hipMemcpyHostToDevice, h->stream));
    const double *dc, *dc2;
    double *dc_w = nullptr, *dc2_w = nullptr;
    if (from_sums) {
        if (in_dev) { dc = c_io; dc2 = c2_io; }
        else {
            double *t0 = take(), *t1 = take();
            HIPCHK(hipMemcpyAsync(t0, c_io, mat, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(t1, c2_io, mat, hipMemcpyHostToDevice, h->stream));
            dc = t0; dc2 = t1;
        }
    } else {
        const double *draw = raw;
        if (!in_dev) { double *t = take(); HIPCHK(hipMemcpyAsync(t, raw, mat, hipMemcpyHostToDevice, h->stream)); draw = t; }
        if (c_io && out_dev) { dc_w = c_io; dc2_w = c2_io; } else { dc_w = take(); dc2_w = take(); }
        { Scope s(h, "k_trace_cumsum");
          hipLaunchKernelGGL(k_trace_cumsum, dim3(n_reads), dim3(64), 0, h->stream, draw, dlen, L, n_reads, dc_w, dc2_w); }
        dc = dc_w; dc2 = dc2_w;
    }
    double *dg = out_dev ? gain_out : take();
    TraceArgs ta = {a.min_obs, a.border_trim, a.stride, a.adapter_early_stopping, a.adapter_early_stop_window, a.adapter_early_stop_stride,
                    a.polya_early_stopping, a.polya_early_stop_window, a.polya_early_stop_stride};
    { Scope s(h, "k_trace_gains");
      hipLaunchKernelGGL(k_trace_gains, dim3(n_reads), dim3(64), 0, h->stream, dc, dc2, dlen, dstart, dend, L, ta, dg); }
    HIPCHK(hipGetLastError());
    if (!out_dev) {
        HIPCHK(hipMemcpyAsync(gain_out, dg, mat, hipMemcpyDeviceToHost, h->stream));
        if (!from_sums && c_io) {
            HIPCHK(hipMemcpyAsync(c_io, dc_w, mat, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(c2_io, dc2_w, mat, hipMemcpyDeviceToHost, h->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return ADP_OK;
}

int adp_debug_log(adp_handle *h, const double *host_in, double *host_out, int n)
{
    if (!h || !host_in || !host_out || n < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    double *d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t)n * 16));
    HIPCHK(hipMemcpyAsync(d, host_in, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_debug_log, dim3((n + 255) / 256), dim3(256), 0, h->stream, d, d + n, n);
    HIPCHK(hipMemcpyAsync(host_out, d + n, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipFree(d));
    return ADP_OK;
}

// fdiv_shared against the IEEE division for `count` consecutive float bit patterns from `first_bits` on (both signs)
__global__ void k_debug_divcheck(float d, uint32_t first_bits, uint32_t count, unsigned long long *mism)
{
    const float y = 1.0f / d;
    unsigned long long bad = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float a = __uint_as_float(first_bits + i);
        if (a != a || __builtin_isinf(a)) continue;
        const float q = fdiv_shared(a, d, y), w = a / d;
        const float qn = fdiv_shared(-a, d, y), wn = -a / d;
        if (__float_as_uint(q) != __float_as_uint(w) || __float_as_uint(qn) != __float_as_uint(wn)) bad++;
    }
    if (bad) atomicAdd(mism, bad);
}

int adp_debug_divcheck(adp_handle *h, float d, uint32_t first_bits, uint32_t count, uint64_t *mismatches_out)
{
    if (!h || !mismatches_out) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    unsigned long long *dm = nullptr;
    HIPCHK(hipMalloc(&dm, 8));
    HIPCHK(hipMemsetAsync(dm, 0, 8, h->stream));
    hipLaunchKernelGGL(k_debug_divcheck, dim3(4096), dim3(256), 0, h->stream, d, first_bits, count, dm);
    HIPCHK(hipMemcpyAsync(mismatches_out, dm, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipFree(dm));
    return ADP_OK;
}

int adp_debug_llr_upto(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m, int minibatch,
                       int flags, int stage)
{
    return llr_pipeline(h, signals, full_len, n_reads, m, minibatch, flags, nullptr, nullptr, stage);
}

int adp_detect_start_peak(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m, int minibatch,
                          int flags, adp_row *rows_out)
{
    if (!h || !signals || !full_len || n_reads < 1 || minibatch < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    if (n_reads > h->max_reads || m != h->m) { g_err = "n_reads/m exceed the handle's capacity"; return ADP_ERR_CAPACITY; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    const float *dsig; const int32_t *dlen;
    int rc = stage_inputs(h, signals, full_len, n_reads, m, flags, &dsig, &dlen);
    if (rc) return rc;
    hipStream_t st = h->stream;
    rc = alloc_all(h, n_reads < minibatch ? n_reads : minibatch, false);
    if (rc) return rc;
    // the pandas float-column quirk couples the reads of ONE minibatch: process minibatch by minibatch (one arena for the
    // whole call: the offsets in the rows of every minibatch stay valid until the next call)
    for (int attempt = 0; attempt < 3; attempt++) {
        rc = arena_begin(h);
        if (rc) return rc;
        for (int s0 = 0; s0 < n_reads; s0 += minibatch) {
            int n = n_reads - s0 < minibatch ? n_reads - s0 : minibatch;
            const float *sg = dsig + (size_t)s0 * m;
            const int32_t *ln = dlen + s0;
            HIPCHK(hipMemsetAsync(h->any_none.p, 0, 4, st));
            { Scope s(h, "k_start_peak");
              hipLaunchKernelGGL(k_start_peak<SigF32>, dim3(n), dim3(64), (size_t)64 * h->cfg.sp_downscale_factor * 4, st, SigF32{sg}, ln, n, m, h->cfg, h->sp.as<SpOut>()); }
            hipLaunchKernelGGL(k_sp_bounds, dim3((n + 255) / 256), dim3(256), 0, st, h->sp.as<SpOut>(), n, h->bounds.as<int64_t>(),
                               h->topk_none.as<int8_t>(), h->any_none.as<int32_t>());
            rc = launch_validate(h, SigF32{sg}, ln, n, m, 1, minibatch, false);
            if (rc) return rc;
            hipLaunchKernelGGL(k_sp_decorate, dim3((n + 255) / 256), dim3(256), 0, st, h->sp.as<SpOut>(), h->rows.as<adp_row>(), n, 1,
                               h->any_none.as<int32_t>());
            if (rows_out) {
                HIPCHK(hipMemcpyAsync(rows_out + s0, h->rows.p, (size_t)n * sizeof(adp_row),
                                      (flags & ADP_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
            }
            HIPCHK(hipStreamSynchronize(st));
        }
        HIPCHK(hipGetLastError());
        rc = arena_end(h);
        if (rc < 0) return rc;
        if (rc == 0) break;
        if (attempt == 2) { g_err = "the call's repeats (conv stack out of the float16 range, open-pore arena growth) are used up and the arena is still short"; return ADP_ERR_CAPACITY; }
        h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    }
    return ADP_OK;
}

int adp_validate_candidates(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m,
                            const int64_t *bounds, int k, int flags, adp_row *rows_out)
{
    if (!h || !signals || !full_len || !bounds || n_reads < 1 || k < 1 || k > ADP_MAX_CAND) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    if (n_reads > h->max_reads || m != h->m) { g_err = "n_reads/m exceed the handle's capacity"; return ADP_ERR_CAPACITY; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    const float *dsig; const int32_t *dlen;
    int rc = stage_inputs(h, signals, full_len, n_reads, m, flags, &dsig, &dlen);
    if (rc) return rc;
    hipStream_t st = h->stream;
    rc = alloc_all(h, n_reads, false);
    if (rc) return rc;
    for (int attempt = 0; attempt < 3; attempt++) {
        rc = arena_begin(h);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(h->bounds.p, bounds, (size_t)n_reads * (1 + k) * 8,
                              ((flags & ADP_IN_DEVICE) && !(flags & ADP_BOUNDS_HOST)) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(h->topk_none.p, (flags & ADP_TOPK_NONE) ? 1 : 0, (size_t)n_reads, st));
        rc = launch_validate(h, SigF32{dsig}, dlen, n_reads, m, k, n_reads, false);
        if (rc) return rc;
        rc = deliver_rows(h, n_reads, flags, rows_out);
        if (rc) return rc;
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        rc = arena_end(h);
        if (rc < 0) return rc;
        if (rc == 0) break;
        if (attempt == 2) { g_err = "the call's repeats (conv stack out of the float16 range, open-pore arena growth) are used up and the arena is still short"; return ADP_ERR_CAPACITY; }
        h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    }
    return ADP_OK;
}

// C3 on the device: top-k behind given arg-maxes (dapos / dppos device int64 [n]) -> dcand / dcnt in ct_out (asynchronous)
static int cnn_topk_dev(adp_handle *h, const float *scores, const long long *dapos, const long long *dppos, int n, int mbsize, int Lo, int k)
{
    if ((long long)(mbsize < n ? mbsize : n) * Lo >= (1ll << 31)) { g_err = "minibatch * Lo too large for 32-bit flat positions"; return ADP_ERR_UNSUPPORTED; }
    const size_t half = (size_t)Lo / 2 + 1;
    const int wpr = (int)((half + 4 + 15) / 16 + 2);
    const size_t lds = (size_t)wpr * 4;
    if (lds > 60000) { g_err = "Lo too large for the LDS state of k_cnn_topk"; return ADP_ERR_UNSUPPORTED; }
    if (h->ct_pk.ensure((size_t)n * 2 * half * 4) || h->ct_pv.ensure((size_t)n * half * 4) || h->ct_st.ensure((size_t)n * wpr * 4) ||
        h->ct_lnz.ensure((size_t)n * 4) || h->ct_out.ensure(((size_t)n * (k + 1) + 1) * 4)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    int32_t *dcand = h->ct_out.as<int32_t>(), *dcnt = dcand + (size_t)n * k;
    const int n_mb = (n + mbsize - 1) / mbsize;
    hipStream_t st = h->stream;
    hipLaunchKernelGGL(k_cnn_rowlink, dim3(n_mb), dim3(1024), 0, st, dapos, dppos, n, mbsize, h->ct_lnz.as<int32_t>());
    { Scope s(h, "k_cnn_topk");
      hipLaunchKernelGGL(k_cnn_topk, dim3(n), dim3(64), lds, st, scores, dapos, dppos, h->ct_lnz.as<int32_t>(), n, mbsize, Lo, k,
                         h->ct_pk.as<int32_t>(), h->ct_pk.as<int32_t>() + (size_t)n * half, h->ct_pv.as<float>(), h->ct_st.as<uint32_t>(), wpr, dcand, dcnt); }
    return 0;
}

int adp_cnn_topk(adp_handle *h, const float *scores_dev, const int64_t *adapter_pos_dev, const int64_t *polya_pos_dev,
                 int n_reads, int Lo, int k, int32_t *cand_out, int32_t *n_peaks_out)
{
    if (!h || !scores_dev || !adapter_pos_dev || !polya_pos_dev || !cand_out || !n_peaks_out || n_reads < 1 || Lo < 3 ||
        k < 1 || k > ADP_MAX_CAND) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    int rc = cnn_topk_dev(h, scores_dev, (const long long *)adapter_pos_dev, (const long long *)polya_pos_dev, n_reads, n_reads, Lo, k);
    if (rc) return rc;
    hipStream_t st = h->stream;
    int32_t *dcand = h->ct_out.as<int32_t>(), *dcnt = dcand + (size_t)n_reads * k;
    HIPCHK(hipMemcpyAsync(cand_out, dcand, (size_t)n_reads * k * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(n_peaks_out, dcnt, (size_t)n_reads * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return ADP_OK;
}

// cnn_predict + the scaling of cnn_detect: scores (device) -> h->bounds [n, 1 + max(k, 1)] (asynchronous)
static int cnn_predict_dev(adp_handle *h, const float *scores, int n, int mbsize, int Lo, int *kmax_out)
{
    const adp_cfg &c = h->cfg;
    const int k = c.polya_cand_k, kk = k < 1 ? 1 : k;
    if (k > ADP_MAX_CAND) { g_err = "polya_cand_k too large"; return ADP_ERR_UNSUPPORTED; }
    if (n > h->max_reads) { g_err = "n_reads exceeds the handle's capacity"; return ADP_ERR_CAPACITY; }
    if (h->ct_ap.ensure((size_t)n * 16) || h->bounds.ensure((size_t)n * (1 + ADP_MAX_CAND) * 8)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    long long *dap = h->ct_ap.as<long long>(), *dpp = dap + n;
    const int na = (c.max_obs_adapter - c.min_obs_adapter) / c.downscale_factor;
    hipStream_t st = h->stream;
    { Scope s(h, "k_cnn_argmax");
      hipLaunchKernelGGL(k_cnn_argmax, dim3(n), dim3(64), 0, st, scores, n, Lo, na, k, dap, dpp); }
    const int32_t *dcand = nullptr, *dcnt = nullptr;
    if (k > 1) {
        int rc = cnn_topk_dev(h, scores, dap, dpp, n, mbsize, Lo, k);
        if (rc) return rc;
        dcand = h->ct_out.as<int32_t>(); dcnt = dcand + (size_t)n * k;
    }
    const int n_mb = (n + mbsize - 1) / mbsize;
    hipLaunchKernelGGL(k_cnn_bounds, dim3(n_mb), dim3(1024), 0, st, dap, dpp, dcand, dcnt, n, mbsize, k, c.downscale_factor, c.min_obs_adapter,
                       h->bounds.as<int64_t>());
    *kmax_out = kk;
    return 0;
}

int adp_cnn_predict(adp_handle *h, const float *scores_dev, int n_reads, int minibatch, int Lo, int64_t *bounds_out)
{
    if (!h || !scores_dev || !bounds_out || n_reads < 1 || minibatch < 1 || Lo < 3) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    int kk = 1;
    int rc = cnn_predict_dev(h, scores_dev, n_reads, minibatch, Lo, &kk);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(bounds_out, h->bounds.p, (size_t)n_reads * (1 + kk) * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return ADP_OK;
}

// C1 on the handle's stream: pooled values (k_cnn_pool), then their per-read median / MAD scaling in place (k_cnn_prepare)
static int launch_cnn_prepare(adp_handle *h, const float *dsig, int n_reads, int m, int off, int ds, int Lc, float *dout)
{
    if (h->npk.ensure((size_t)n_reads * 4)) { g_err = "device allocation failed"; return ADP_ERR_HIP; } // (npk: unused on this path)
    int32_t *nan_cnt = h->npk.as<int32_t>();
    HIPCHK(hipMemsetAsync(nan_cnt, 0, (size_t)n_reads * 4, h->stream));
    { Scope s(h, "k_cnn_pool");
      hipLaunchKernelGGL(k_cnn_pool, dim3(n_reads), dim3(256), (size_t)4 * 64 * ds * 4, h->stream, dsig, n_reads, m, off, ds, Lc, dout, nan_cnt); }
    { Scope s(h, "k_cnn_prepare");
      hipLaunchKernelGGL(k_cnn_prepare, dim3(n_reads), dim3(64), 0, h->stream, n_reads, Lc, dout, (const int32_t *)nan_cnt); }
    return 0;
}

int adp_cnn_prepare(adp_handle *h, const float *signals, int n_reads, int m, int flags, float *prepared_out)
{
    if (!h || !signals || !prepared_out || n_reads < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    if (m != h->m) { g_err = "m differs from the handle's"; return ADP_ERR_CAPACITY; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    const int off = h->cfg.min_obs_adapter, ds = h->cfg.downscale_factor;
    if (m <= off) { g_err = "preload shorter than min_obs_adapter"; return ADP_ERR_INVALID; }
    const int Lc = (m - off + ds - 1) / ds;
    const float *dsig = signals;
    if (!(flags & ADP_IN_DEVICE)) {
        if (h->sig_stage.ensure((size_t)n_reads * m * 4)) { g_err = "staging allocation failed"; return ADP_ERR_HIP; }
        HIPCHK(hipMemcpyAsync(h->sig_stage.p, signals, (size_t)n_reads * m * 4, hipMemcpyHostToDevice, h->stream));
        dsig = h->sig_stage.as<float>();
    }
    float *dout = prepared_out;
    if (!(flags & ADP_OUT_DEVICE)) {
        if (h->bounds_stage.ensure((size_t)n_reads * Lc * 4)) { g_err = "staging allocation failed"; return ADP_ERR_HIP; }
        dout = h->bounds_stage.as<float>();
    }
    { int rc2 = launch_cnn_prepare(h, dsig, n_reads, m, off, ds, Lc, dout); if (rc2) return rc2; }
    if (!(flags & ADP_OUT_DEVICE))
        HIPCHK(hipMemcpyAsync(prepared_out, dout, (size_t)n_reads * Lc * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return ADP_OK;
}

// offsets (floats) of the layers inside the weight buffer: w0 [64][1][7], b0 [64], w1 [64][64][7], b1 [64], w2, b2, w3 [64][2][7], b3 [2]
#define CNN_W0 0
#define CNN_B0 (CNN_W0 + CNN_C * CNN_K)
#define CNN_W1 (CNN_B0 + CNN_C)
#define CNN_B1 (CNN_W1 + CNN_C * CNN_C * CNN_K)
#define CNN_W2 (CNN_B1 + CNN_C)
#define CNN_B2 (CNN_W2 + CNN_C * CNN_C * CNN_K)
#define CNN_W3 (CNN_B2 + CNN_C)
#define CNN_B3 (CNN_W3 + CNN_C * 2 * CNN_K)
#define CNN_W3S (CNN_B3 + 2)                 // layer 3's weights times 1 / CNS_ASCALE (exact: a power of two) for the split stack's last layer
#define CNN_WTOTAL (CNN_W3S + CNN_C * 2 * CNN_K)

int adp_cnn_set_weights(adp_handle *h, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                        const float *b2, const float *w3, const float *b3)
{
    if (!h || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !w3 || !b3) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    if (h->cnn_w.ensure((size_t)CNN_WTOTAL * 4)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    std::vector<float> all((size_t)CNN_WTOTAL);
    memcpy(&all[CNN_W0], w0, sizeof(float) * CNN_C * CNN_K); memcpy(&all[CNN_B0], b0, sizeof(float) * CNN_C);
    memcpy(&all[CNN_W1], w1, sizeof(float) * CNN_C * CNN_C * CNN_K); memcpy(&all[CNN_B1], b1, sizeof(float) * CNN_C);
    memcpy(&all[CNN_W2], w2, sizeof(float) * CNN_C * CNN_C * CNN_K); memcpy(&all[CNN_B2], b2, sizeof(float) * CNN_C);
    memcpy(&all[CNN_W3], w3, sizeof(float) * CNN_C * 2 * CNN_K); memcpy(&all[CNN_B3], b3, sizeof(float) * 2);
    for (int i = 0; i < CNN_C * 2 * CNN_K; i++) all[CNN_W3S + i] = w3[i] * (1.0f / CNS_ASCALE);
    HIPCHK(hipMemcpyAsync(h->cnn_w.p, all.data(), (size_t)CNN_WTOTAL * 4, hipMemcpyHostToDevice, h->stream));
    // the 64 -> 64 layers once more as split float16 B fragments (cnn_conv_split.h), scaled by a power of two per layer so that the
    // largest weight lands in [2^13, 2^14)
    if (h->cnn_wsp.ensure(((size_t)2 * CNS_WSP_LAYER + CNS_W3SP + CNS_W0SP) * 2)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    for (int layer = 0; layer < 2; layer++) {
        const float *wl = &all[layer ? CNN_W2 : CNN_W1];
        float mx = 0.f;
        for (int i = 0; i < CNN_C * CNN_C * CNN_K; i++) { const float a = fabsf(wl[i]); if (a > mx && a < INFINITY) mx = a; }
        int e = 0;
        if (mx > 0.f) (void)frexpf(mx, &e); // mx = f 2^e, f in [0.5, 1)
        int se = 14 - e;
        if (se > 100) se = 100;
        if (se < -100) se = -100;
        h->cnn_sw[layer] = ldexpf(1.0f, se);
        hipLaunchKernelGGL(k_cns_split_weights, dim3((2 * CNS_KSTEPS * 64 + 255) / 256), dim3(256), 0, h->stream,
                           h->cnn_w.as<float>() + (layer ? CNN_W2 : CNN_W1), h->cnn_sw[layer], h->cnn_wsp.as<_Float16>() + (size_t)layer * CNS_WSP_LAYER);
    }
    { // layer 3 as A fragments of one more GEMM in layer 2's kernel (k_cns_split_w3), scaled like the others
        float mx = 0.f;
        for (int i = 0; i < CNN_C * 2 * CNN_K; i++) { const float a = fabsf(w3[i]); if (a > mx && a < INFINITY) mx = a; }
        int e = 0;
        if (mx > 0.f) (void)frexpf(mx, &e);
        int se = 14 - e;
        if (se > 100) se = 100;
        if (se < -100) se = -100;
        h->cnn_sw[2] = ldexpf(1.0f, se);
        hipLaunchKernelGGL(k_cns_split_w3, dim3(1), dim3(256), 0, h->stream, h->cnn_w.as<float>() + CNN_W3, h->cnn_sw[2], h->cnn_wsp.as<_Float16>() + (size_t)2 * CNS_WSP_LAYER);
    }
    { // layer 0 as A fragments of a small GEMM in layer 1's prologue (k_cns_split_w0)
        float mx = 0.f;
        for (int i = 0; i < CNN_C * CNN_K; i++) { const float a = fabsf(w0[i]); if (a > mx && a < INFINITY) mx = a; }
        int e = 0;
        if (mx > 0.f) (void)frexpf(mx, &e);
        int se = 14 - e;
        if (se > 100) se = 100;
        if (se < -100) se = -100;
        h->cnn_sw[3] = ldexpf(1.0f, se);
        hipLaunchKernelGGL(k_cns_split_w0, dim3(1), dim3(128), 0, h->stream, h->cnn_w.as<float>() + CNN_W0, h->cnn_sw[3],
                           h->cnn_wsp.as<_Float16>() + (size_t)2 * CNS_WSP_LAYER + CNS_W3SP);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->cnn_have_w = true;
    return ADP_OK;
}

extern "C++" {
template <int NT>
static int launch_conv64(adp_handle *h, const float *in, float *out, const float *w, const float *b, int n, int L1, int Lpad, int tiles)
{
    const size_t lds = (size_t)2 * CNN_C * (64 * NT + 8) * 4;
    const unsigned bit = 4u << (NT - 2);
    if (!(h->attr_done & bit)) { HIPCHK(hipFuncSetAttribute((const void *)k_cnn_conv64<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); h->attr_done |= bit; }
    long long total = (long long)n * tiles;
    int grid = (int)(total < h->n_cu ? total : h->n_cu);
    hipLaunchKernelGGL(k_cnn_conv64<NT>, dim3(grid), dim3(256), lds, h->stream, in, out, w, b, n, L1, Lpad, tiles);
    return 0;
}
} // extern "C++"

extern "C++" {
template <int NT, bool LAST, bool FIRST = false>
static int launch_conv64s(adp_handle *h, const _Float16 *in, _Float16 *out, const _Float16 *wsp, const float *b, float sw, int n, int L1,
                          int Lrows, int tiles, int32_t *flag, const _Float16 *w3sp = nullptr, const float *b3 = nullptr, float inv_s3 = 0.f,
                          float *scores = nullptr, int Lo = 0, CnsFirst first = CnsFirst{})
{
    // (LAST: layer 3 in the epilogue -- tiles that advance by 64 NT - 2 positions, the taps' partial sums of a tile in LDS behind the two tile buffers)
    const size_t lds = (size_t)2 * (((size_t)(64 * NT + 6) * CNS_ROWB + 1023) / 1024 * 1024) + (LAST ? (size_t)2 * CNN_K * 64 * NT * 4 : 0);
    const unsigned bit = (FIRST ? 16384u : LAST ? 512u : 64u) << (NT - 2);
    if (!(h->attr_done & bit)) { HIPCHK(hipFuncSetAttribute((const void *)k_cnn_conv64s<NT, LAST, FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); h->attr_done |= bit; }
    long long total = (long long)n * tiles;
    if (total >= (1ll << 31)) { g_err = "too many tiles for one launch of k_cnn_conv64s"; return ADP_ERR_UNSUPPORTED; }
    int grid = (int)(total < h->n_cu ? total : h->n_cu);
    hipLaunchKernelGGL((k_cnn_conv64s<NT, LAST, FIRST>), dim3(grid), dim3(256), lds, h->stream, in, out, wsp, b, sw, 1.0f / sw, n, L1, Lrows, tiles, flag,
                       w3sp, b3, inv_s3, scores, Lo, first);
    return 0;
}
} // extern "C++"

// the conv stack on split float16 operands (cnn_conv_split.h); the out-of-range flag is the second word of the call's arena counter
static int cnn_forward_split(adp_handle *h, adp_handle *wh, const float *prepared, int n_reads, int Lc, int L1, int Lo, int NT, float *scores_out)
{
    const int PB = 64 * NT, tiles = (L1 + PB - 1) / PB, Lrows = CNS_FRONT + tiles * PB + 4;
    const size_t per_read = (size_t)Lrows * CNS_ROWB;
    size_t cap_reads = ((size_t)4 << 30) / per_read; // two activation buffers of at most 4 GiB each
    if (cap_reads < 1) cap_reads = 1;
    if (cap_reads > 65535) cap_reads = 65535;        // (the first / last layer's kernels take the chunk's reads as grid.y)
    int C = (int)((size_t)n_reads < cap_reads ? (size_t)n_reads : cap_reads);
    if (h->cnn_Lpad != Lrows || h->cnn_L1 != L1 || h->cnn_chunk < C) {
        for (int k = 0; k < 2; k++) {
            if (h->cnn_act[k].ensure((size_t)C * per_read + CNS_SLACK)) { g_err = "device allocation failed"; return ADP_ERR_HIP; } // (+ what the tile DMAs read past the last read's rows)
            HIPCHK(hipMemsetAsync(h->cnn_act[k].p, 0, h->cnn_act[k].cap, h->stream)); // the padding rows are never written again
        }
        h->cnn_Lpad = Lrows; h->cnn_L1 = L1; h->cnn_chunk = (int)((h->cnn_act[0].cap - CNS_SLACK) / per_read); if (h->cnn_chunk > 65535) h->cnn_chunk = 65535;
    }
    C = h->cnn_chunk < n_reads ? h->cnn_chunk : n_reads;
    if (wh->op_used.ensure(8)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    int32_t *flag = wh->op_used.as<int32_t>() + 1;
    const float *W = wh->cnn_w.as<float>();
    const _Float16 *wsp = wh->cnn_wsp.as<_Float16>();
    _Float16 *bufs[2] = {h->cnn_act[0].as<_Float16>(), h->cnn_act[1].as<_Float16>()};
    // The chunks' LAST layer runs on the side stream beside the NEXT chunk's first layer (round 4): layer 3 only reads split rows (4.4 TB/s),
    // layer 0 only writes them (3.5 TB/s), neither touches the matrix cores -- together they move more bytes per second than in turn.  The
    // two activation buffers swap roles from chunk to chunk (X: layer 0's output and layer 2's; Y: layer 1's), so layer 0 of chunk c + 1
    // writes what layer 2 of chunk c has just finished READING while layer 3 of chunk c reads the other buffer; layer 1 of chunk c + 1 waits
    // for that layer 3 (it overwrites its input).  ADP_CNN_OVERLAP=0: everything in turn on one stream, as before.
    // ADP_CNN_FOLD=1 (round 5): layer 3 in layer 2's epilogue (k_cnn_conv64s<NT, true>): no rows of layer 2 in HBM, no k_cnn_conv_out_s, nothing
    // left to run beside the next chunk's first layer -- one stream, the buffers keep their roles
    const bool fold = env_int("ADP_CNN_FOLD", 1) != 0;
    // ADP_CNN_FUSE_IN=1 (round 5): layer 0 in layer 1's prologue, on the matrix cores (k_cnn_conv64s<NT, false, true>): no k_cnn_conv_in_s, no
    // rows of layer 0 in HBM
    const bool fuse_in = env_int("ADP_CNN_FUSE_IN", 1) != 0;
    const int tiles_last = (L1 + PB - 3) / (PB - 2); // tiles of the folded layer: they advance by PB - 2 positions
    const int tiles_first = (L1 + PB - 7) / (PB - 6); // ... of the layer that makes its input rows: by PB - 6
    const bool overlap = env_int("ADP_CNN_OVERLAP", 1) != 0 && !fold && !fuse_in;
    if (overlap && !h->ev_conv[0]) {
        for (int i = 0; i < 3; i++) if (hipEventCreateWithFlags(&h->ev_conv[i], hipEventDisableTiming) != hipSuccess) { g_err = "hipEventCreate failed"; return ADP_ERR_HIP; }
    }
    int chunk = 0;
    bool out_pending = false; // a layer 3 on the side stream that the main stream has not waited for yet
    // an error return from inside the chunk loop must not leave that layer 3 running: the caller's repeat (or its float32 fallback)
    // reuses the activation buffers and the scores on the main stream
    struct SideJoin { hipStream_t side; bool *pending; ~SideJoin() { if (*pending) (void)hipStreamSynchronize(side); } } side_join{h->stream2, &out_pending};
    for (int s0 = 0; s0 < n_reads; s0 += C, chunk++) {
        const int n = n_reads - s0 < C ? n_reads - s0 : C;
        const float *x = prepared + (size_t)s0 * Lc;
        float *sc = scores_out + (size_t)s0 * 2 * Lo;
        _Float16 *A = bufs[overlap ? (chunk & 1) : 0], *B = bufs[overlap ? 1 - (chunk & 1) : 1];
        if (!fuse_in) {
            Scope s(h, "k_cnn_conv_in");
            hipLaunchKernelGGL(k_cnn_conv_in_s, dim3((L1 + CNS_IN_P * CNS_IN_T - 1) / (CNS_IN_P * CNS_IN_T), n), dim3(256), 0, h->stream, x, Lc, L1, Lrows, W + CNN_W0, W + CNN_B0, A, flag);
        }
        if (out_pending) { HIPCHK(hipStreamWaitEvent(h->stream, h->ev_conv[1 + ((chunk - 1) & 1)], 0)); out_pending = false; } // (layer 1 overwrites what that layer 3 reads)
        for (int layer = 0; layer < 2; layer++) {
            Scope s(h, layer ? "k_cnn_conv64 (layer 2)" : "k_cnn_conv64 (layer 1)");
            const _Float16 *in = layer ? B : A; _Float16 *out = layer ? A : B;
            const _Float16 *w = wsp + (size_t)layer * CNS_WSP_LAYER;
            const float *b = W + (layer ? CNN_B2 : CNN_B1);
            const float sw = wh->cnn_sw[layer];
            int rc;
            if (layer == 1 && fold) {
                const _Float16 *w3sp = wsp + (size_t)2 * CNS_WSP_LAYER;
                const float inv_s3 = 1.0f / (wh->cnn_sw[2] * CNS_ASCALE);
                rc = NT == 4 ? launch_conv64s<4, true>(h, in, out, w, b, sw, n, L1, Lrows, tiles_last, flag, w3sp, W + CNN_B3, inv_s3, sc, Lo)
                   : NT == 3 ? launch_conv64s<3, true>(h, in, out, w, b, sw, n, L1, Lrows, tiles_last, flag, w3sp, W + CNN_B3, inv_s3, sc, Lo)
                             : launch_conv64s<2, true>(h, in, out, w, b, sw, n, L1, Lrows, tiles_last, flag, w3sp, W + CNN_B3, inv_s3, sc, Lo);
            } else if (layer == 0 && fuse_in) {
                const CnsFirst first{x, Lc, wsp + (size_t)2 * CNS_WSP_LAYER + CNS_W3SP, W + CNN_B0, wh->cnn_sw[3]};
                rc = NT == 4 ? launch_conv64s<4, false, true>(h, in, out, w, b, sw, n, L1, Lrows, tiles_first, flag, nullptr, nullptr, 0.f, nullptr, 0, first)
                   : NT == 3 ? launch_conv64s<3, false, true>(h, in, out, w, b, sw, n, L1, Lrows, tiles_first, flag, nullptr, nullptr, 0.f, nullptr, 0, first)
                             : launch_conv64s<2, false, true>(h, in, out, w, b, sw, n, L1, Lrows, tiles_first, flag, nullptr, nullptr, 0.f, nullptr, 0, first);
            } else
                rc = NT == 4 ? launch_conv64s<4, false>(h, in, out, w, b, sw, n, L1, Lrows, tiles, flag)
                   : NT == 3 ? launch_conv64s<3, false>(h, in, out, w, b, sw, n, L1, Lrows, tiles, flag)
                             : launch_conv64s<2, false>(h, in, out, w, b, sw, n, L1, Lrows, tiles, flag);
            if (rc) return rc;
        }
        if (fold) continue;
        const bool last = s0 + C >= n_reads;
        hipStream_t so = (overlap && !last) ? h->stream2 : h->stream; // (the last chunk's layer 3 has nothing to run beside)
        if (so != h->stream) { HIPCHK(hipEventRecord(h->ev_conv[0], h->stream)); HIPCHK(hipStreamWaitEvent(so, h->ev_conv[0], 0)); }
        { Scope s(h, "k_cnn_conv_out", so);
          hipLaunchKernelGGL(k_cnn_conv_out_s, dim3((L1 + CNS_OUT_P - 1) / CNS_OUT_P, n), dim3(CNS_OUT_P), 0, so, A, L1, Lrows, Lo, W + CNN_W3S, W + CNN_B3, sc); }
        if (so != h->stream) { HIPCHK(hipEventRecord(h->ev_conv[1 + (chunk & 1)], so)); out_pending = true; }
    }
    if (out_pending) { HIPCHK(hipStreamWaitEvent(h->stream, h->ev_conv[1 + ((chunk - 1) & 1)], 0)); out_pending = false; }
    return 0;
}

// the conv stack over device buffers, asynchronous on the handle's stream
static int cnn_forward_dev(adp_handle *h, const float *prepared, int n_reads, int Lc, float *scores_out)
{
    adp_handle *wh = h->owner ? h->owner : h; // (a lane runs with its parent's weights)
    if (!wh->cnn_have_w) { g_err = "adp_cnn_set_weights has not been called"; return ADP_ERR_INVALID; }
    const int L1 = (Lc + 2 * 3 - CNN_K) / 3 + 1, Lo = (L1 - 1) * 3 - 2 * 3 + CNN_K;
    // positions per workgroup step: the NT (32-position tiles per wave) that wastes least of the last step
    int NT = 4; { long long best = -1; for (int nt = 4; nt >= 2; nt--) { const long long pb = 64 * nt, cover = (L1 + pb - 1) / pb * pb; if (best < 0 || cover < best) { best = cover; NT = nt; } } }
    if (wh->cnn_mode == 1 && !wh->cnn_redo_f32) return cnn_forward_split(h, wh, prepared, n_reads, Lc, L1, Lo, NT, scores_out);
    const int PB = 64 * NT, tiles = (L1 + PB - 1) / PB, Lpad = tiles * PB + 8;
    size_t per_read = (size_t)CNN_C * Lpad * 4;
    size_t cap_reads = ((size_t)4 << 30) / per_read; // two activation buffers of at most 4 GiB each
    if (cap_reads < 1) cap_reads = 1;
    if (cap_reads > 65535) cap_reads = 65535;        // (the first / last layer's kernels take the chunk's reads as grid.y)
    int C = (int)((size_t)n_reads < cap_reads ? (size_t)n_reads : cap_reads);
    // (buffers of its own: a call that falls back from the split stack -- an activation beyond the float16 range -- and the split
    // call after it used to share one pair and cleared up to 2 x 4 GiB of padding on every change of kind)
    if (h->cnn_f_Lpad != Lpad || h->cnn_f_L1 != L1 || h->cnn_f_chunk < C) {
        for (int k = 0; k < 2; k++) {
            if (h->cnn_actf[k].ensure((size_t)C * per_read)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
            HIPCHK(hipMemsetAsync(h->cnn_actf[k].p, 0, h->cnn_actf[k].cap, h->stream)); // the padding columns are never written again
        }
        h->cnn_f_Lpad = Lpad; h->cnn_f_L1 = L1; h->cnn_f_chunk = (int)(h->cnn_actf[0].cap / per_read); if (h->cnn_f_chunk > 65535) h->cnn_f_chunk = 65535;
    }
    C = h->cnn_f_chunk < n_reads ? h->cnn_f_chunk : n_reads;
    const float *W = wh->cnn_w.as<float>();
    float *A = h->cnn_actf[0].as<float>(), *B = h->cnn_actf[1].as<float>();
    for (int s0 = 0; s0 < n_reads; s0 += C) {
        const int n = n_reads - s0 < C ? n_reads - s0 : C;
        const float *x = prepared + (size_t)s0 * Lc;
        float *sc = scores_out + (size_t)s0 * 2 * Lo;
        { Scope s(h, "k_cnn_conv_in");
          hipLaunchKernelGGL(k_cnn_conv_in, dim3((L1 + 255) / 256, n), dim3(256), 0, h->stream, x, Lc, L1, Lpad, W + CNN_W0, W + CNN_B0, A); }
        for (int layer = 0; layer < 2; layer++) {
            Scope s(h, layer ? "k_cnn_conv64 (layer 2)" : "k_cnn_conv64 (layer 1)");
            const float *in = layer ? B : A; float *out = layer ? A : B;
            const float *w = W + (layer ? CNN_W2 : CNN_W1), *b = W + (layer ? CNN_B2 : CNN_B1);
            int rc = NT == 4 ? launch_conv64<4>(h, in, out, w, b, n, L1, Lpad, tiles)
                   : NT == 3 ? launch_conv64<3>(h, in, out, w, b, n, L1, Lpad, tiles)
                             : launch_conv64<2>(h, in, out, w, b, n, L1, Lpad, tiles);
            if (rc) return rc;
        }
        { Scope s(h, "k_cnn_conv_out");
          hipLaunchKernelGGL(k_cnn_conv_out, dim3((L1 + 255) / 256, n), dim3(256), 0, h->stream, A, L1, Lpad, Lo, W + CNN_W3, W + CNN_B3, sc); }
    }
    return 0;
}

int adp_cnn_forward(adp_handle *h, const float *prepared, int n_reads, int Lc, float *scores_out)
{
    if (!h || !prepared || !scores_out || n_reads < 1 || Lc < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    h->cnn_redo_f32 = false;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (h->op_used.ensure(8)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
        HIPCHK(hipMemsetAsync(h->op_used.as<int32_t>() + 1, 0, 4, h->stream));
        int rc = cnn_forward_dev(h, prepared, n_reads, Lc, scores_out);
        if (rc) return rc;
        HIPCHK(hipGetLastError());
        int32_t flag = 0;
        if (h->cnn_mode == 1 && !h->cnn_redo_f32) HIPCHK(hipMemcpyAsync(&flag, h->op_used.as<int32_t>() + 1, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (!flag) break;
        h->cnn_redo_f32 = true; // an activation left the float16 range: once more on the float32 kernels
        h->prof.clear(); h->ev_used = 0;
    }
    return ADP_OK;
}

// combined_detect_cnn up to (not including) the short-read fallback: C1 prepare -> C2 conv stack -> C3 predict -> V1 with the
// k candidates, enqueued on the handle's stream without waiting for anything on the host.  bounds_dst (may be NULL): int64
// [n, 1 + max(k, 1)] what cnn_detect returns, copied with bounds_kind.
static int cnn_enqueue(adp_handle *h, const float *dsig, const int32_t *dlen, int n_reads, int m, int minibatch, adp_row *rows_dst,
                       int rows_kind, int64_t *bounds_dst, int bounds_kind)
{
    const int off = h->cfg.min_obs_adapter, ds = h->cfg.downscale_factor;
    const int Lc = (m - off + ds - 1) / ds, L1 = (Lc - 1) / 3 + 1, Lo = 3 * L1 - 2;
    if (h->cnn_x.ensure((size_t)n_reads * Lc * 4) || h->cnn_sc.ensure((size_t)n_reads * 2 * Lo * 4)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    hipStream_t st = h->stream;
    int rc = alloc_all(h, n_reads, false);
    if (rc) return rc;
    rc = launch_cnn_prepare(h, dsig, n_reads, m, off, ds, Lc, h->cnn_x.as<float>());
    if (rc) return rc;
    rc = cnn_forward_dev(h, h->cnn_x.as<float>(), n_reads, Lc, h->cnn_sc.as<float>());
    if (rc) return rc;
    int kk = 1;
    rc = cnn_predict_dev(h, h->cnn_sc.as<float>(), n_reads, minibatch, Lo, &kk);
    if (rc) return rc;
    if (bounds_dst) HIPCHK(hipMemcpyAsync(bounds_dst, h->bounds.p, (size_t)n_reads * (1 + kk) * 8, (hipMemcpyKind)bounds_kind, st));
    HIPCHK(hipMemsetAsync(h->topk_none.p, 0, (size_t)n_reads, st));
    rc = launch_validate(h, SigF32{dsig}, dlen, n_reads, m, kk, n_reads, false);
    if (rc) return rc;
    if (rows_dst) HIPCHK(hipMemcpyAsync(rows_dst, h->rows.p, (size_t)n_reads * sizeof(adp_row), (hipMemcpyKind)rows_kind, st));
    HIPCHK(hipGetLastError());
    return ADP_OK;
}

// Chunks of whole minibatches over two lanes, free-running: while one lane's chunk is in the conv stack (matrix cores, one
// wave per SIMD, little HBM traffic), the other's is in the candidate validation -- the moving-window recurrences (latency of one
// wave's instruction stream, ~11 ms per launch whatever the number of reads), the order-statistics sweeps (HBM), the partition
// statistics.  Reads of different minibatches never interact on this path (find_peaks / row compaction are per minibatch).
static int cnn_grouped(adp_handle *h, const float *dsig, const int32_t *dlen, int n, int m, int minibatch, int flags, adp_row *rows_out,
                       int64_t *bounds_out, int mb_per_group, int n_lanes)
{
    const int n_mb = (n + minibatch - 1) / minibatch;
    const int G = (n_mb + mb_per_group - 1) / mb_per_group;
    const int k = h->cfg.polya_cand_k, kk = k < 1 ? 1 : k;
    const bool out_dev = (flags & ADP_OUT_DEVICE) != 0;
    h->last_n = n; h->last_nmb = n_mb; h->last_grouped = true;
    if ((rows_out && !out_dev && h->rows.ensure((size_t)n * sizeof(adp_row))) || (bounds_out && h->bounds.ensure((size_t)n * (1 + ADP_MAX_CAND) * 8))) {
        g_err = "device allocation failed"; return ADP_ERR_HIP;
    }
    adp_handle *lanes[ADP_MAX_LANES];
    for (int i = 0; i < n_lanes; i++) {
        int rc = lane_get(h, i, mb_per_group * minibatch, &lanes[i]);
        if (rc) return rc;
        lanes[i]->prof.clear(); lanes[i]->ev_used = 0;
    }
    adp_row *rows_dev = rows_out ? (out_dev ? rows_out : h->rows.as<adp_row>()) : nullptr;
    int64_t *bounds_dev = bounds_out ? h->bounds.as<int64_t>() : nullptr;
    for (int attempt = 0; attempt < 3; attempt++) {
        int rc = arena_begin(h);
        if (rc) return rc;
        HIPCHK(hipEventRecord(h->ev_start, h->stream));
        for (int i = 0; i < n_lanes; i++) HIPCHK(hipStreamWaitEvent(lanes[i]->stream, h->ev_start, 0));
        for (int g = 0; g < G; g++) {
            adp_handle *l = lanes[g % n_lanes];
            const int r0 = g * mb_per_group * minibatch;
            const int ng = (n - r0) < mb_per_group * minibatch ? (n - r0) : mb_per_group * minibatch;
            rc = cnn_enqueue(l, dsig + (size_t)r0 * m, dlen + r0, ng, m, minibatch, rows_dev ? rows_dev + r0 : nullptr, hipMemcpyDeviceToDevice,
                             bounds_dev ? bounds_dev + (size_t)r0 * (1 + kk) : nullptr, hipMemcpyDeviceToDevice);
            if (rc) { for (int i = 0; i < n_lanes; i++) (void)hipStreamSynchronize(lanes[i]->stream); return rc; }
        }
        for (int i = 0; i < n_lanes; i++) HIPCHK(hipStreamSynchronize(lanes[i]->stream));
        rc = arena_end(h, true);
        if (rc < 0) return rc;
        if (rc == 0) break;
        if (attempt == 2) { g_err = "the call's repeats (conv stack out of the float16 range, open-pore arena growth) are used up and the arena is still short"; return ADP_ERR_CAPACITY; }
        for (int i = 0; i < n_lanes; i++) { lanes[i]->prof.clear(); lanes[i]->ev_used = 0; }
    }
    if (rows_out && !out_dev) HIPCHK(hipMemcpyAsync(rows_out, h->rows.p, (size_t)n * sizeof(adp_row), hipMemcpyDeviceToHost, h->stream));
    if (bounds_out) HIPCHK(hipMemcpyAsync(bounds_out, h->bounds.p, (size_t)n * (1 + kk) * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ADP_OK;
}

// combined_detect_cnn up to its short-read fallback in one call.  bounds_out (host, may be NULL): int64 [n, 1 + max(k, 1)], what
// cnn_detect returns.
int adp_detect_cnn(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m, int minibatch, int flags,
                   adp_row *rows_out, int64_t *bounds_out)
{
    if (!h || !signals || !full_len || n_reads < 1 || minibatch < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    if (n_reads > h->max_reads || m != h->m) { g_err = "n_reads/m exceed the handle's capacity"; return ADP_ERR_CAPACITY; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    if (m <= h->cfg.min_obs_adapter) { g_err = "preload shorter than min_obs_adapter"; return ADP_ERR_INVALID; }
    if (!h->cnn_have_w) { g_err = "adp_cnn_set_weights has not been called"; return ADP_ERR_INVALID; }
    h->cnn_redo_f32 = false;
    const float *dsig; const int32_t *dlen;
    int rc = stage_inputs(h, signals, full_len, n_reads, m, flags, &dsig, &dlen);
    if (rc) return rc;
    // ADP_CNN_GROUPS (opt-in): unset / 1 = one chunk on one stream; 0 = automatic (chunks of about a quarter of the call, at most
    // what the conv stack's activation buffers hold, over two lanes); k = aim at k chunks.  Measured at the 200 k window (8000
    // reads, profiles/r03_cnn_chunks.txt): 4 chunks over 2 lanes 107.5 ms against 95.4 ms for one chunk -- the moving-window
    // series costs the same ~11 ms per LAUNCH whatever the number of reads (one wave's instruction stream per chain), so
    // chunking multiplies it, and the conv stack stretches by as much as its neighbours gain; equal at the default window.
    const int n_mb = (n_reads + minibatch - 1) / minibatch;
    int want = env_int("ADP_CNN_GROUPS", 1), n_lanes = env_int("ADP_CNN_LANES", 2);
    if (n_lanes < 1) n_lanes = 1;
    if (n_lanes > ADP_MAX_LANES) n_lanes = ADP_MAX_LANES;
    if (want <= 0) want = 4;
    if (n_mb >= 2 && want > 1) {
        int per = (n_mb + want - 1) / want;
        // a chunk beyond the activation buffers' capacity would be cut again inside the conv stack: keep chunks below it
        const int off = h->cfg.min_obs_adapter, ds = h->cfg.downscale_factor;
        const int Lc = (m - off + ds - 1) / ds, L1 = (Lc - 1) / 3 + 1;
        const size_t per_read = (size_t)CNN_C * ((size_t)L1 + 264) * 4;
        long long cap_mb = (long long)(((size_t)4 << 30) / per_read) / minibatch;
        if (cap_mb >= 1 && per > cap_mb) per = (int)cap_mb;
        if (per < 1) per = 1;
        const int G = (n_mb + per - 1) / per;
        if (G < n_lanes) n_lanes = G;
        if (G >= 2) return cnn_grouped(h, dsig, dlen, n_reads, m, minibatch, flags, rows_out, bounds_out, per, n_lanes);
    }
    for (int attempt = 0; attempt < 3; attempt++) {
        rc = arena_begin(h);
        if (rc) return rc;
        rc = cnn_enqueue(h, dsig, dlen, n_reads, m, minibatch, rows_out, (flags & ADP_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                         bounds_out, hipMemcpyDeviceToHost);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        rc = arena_end(h, true);
        if (rc < 0) return rc;
        if (rc == 0) break;
        if (attempt == 2) { g_err = "the call's repeats (conv stack out of the float16 range, open-pore arena growth) are used up and the arena is still short"; return ADP_ERR_CAPACITY; }
        h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    }
    return ADP_OK;
}

__global__ void k_refine_out(const MbState *mbs, const int32_t *nvalid, const int32_t *polya_idx, const int64_t *ranges, int n,
                             int ds, int64_t *out, int32_t *status)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int st = 0;
    if (mbs[r].status == ADP_MB_MAD_ZERO) st = ADP_F_EXC_MAD_ZERO;
    else if (nvalid[r] <= 0) st = ADP_F_EXC_EMPTY_TRACE;
    int p = polya_idx[r];
    out[r] = (st == 0 && p > 0) ? (int64_t)p * ds + ranges[2 * r] : 0;
    status[r] = st;
}

int adp_llr_refine_polya(adp_handle *h, const float *signals, const int32_t *full_len, int n, int m, const int64_t *ranges,
                         int flags, int64_t *polya_out, int32_t *status_out)
{
    if (!h || !signals || !full_len || !ranges || !polya_out || !status_out || n < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    if (n > h->max_reads || m != h->m) { g_err = "n_reads/m exceed the handle's capacity"; return ADP_ERR_CAPACITY; }
    HIPCHK(hipSetDevice(h->device));
    h->prof.clear(); h->ev_used = 0; h->last_grouped = false;
    const float *dsig; const int32_t *dlen;
    int rc = stage_inputs(h, signals, full_len, n, m, flags, &dsig, &dlen);
    if (rc) return rc;
    hipStream_t st = h->stream;
    rc = alloc_all(h, n, true);
    if (rc) return rc;
    if (h->mbs.ensure((size_t)n * sizeof(MbState)) || h->ghist.ensure((size_t)n * N1_BINS * 4) || h->gbelow.ensure((size_t)n * 8) || h->gcnt.ensure((size_t)n * 8 * N1_NCNT) ||
        h->bounds_stage.ensure((size_t)n * 16 + (size_t)n * 12)) { g_err = "device allocation failed"; return ADP_ERR_HIP; }
    int64_t *drng = h->bounds_stage.as<int64_t>();
    int64_t *dout = drng + 2 * (size_t)n;
    int32_t *dstat = reinterpret_cast<int32_t *>(dout + n);
    HIPCHK(hipMemcpyAsync(drng, ranges, (size_t)n * 16, (flags & ADP_IN_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    MbState *mbs = h->mbs.as<MbState>();
    HIPCHK(hipMemsetAsync(mbs, 0, (size_t)n * sizeof(MbState), st));
    HIPCHK(hipMemsetAsync(h->ghist.p, 0, (size_t)n * N1_BINS * 4, st));
    HIPCHK(hipMemsetAsync(h->gbelow.p, 0, (size_t)n * 8, st));
    HIPCHK(hipMemsetAsync(h->adapter_idx.p, 0, (size_t)n * 4, st));
    HIPCHK(hipMemsetAsync(h->gcnt.p, 0, (size_t)n * 8 * N1_NCNT, st));
    rc = launch_n1(h, SigF32{dsig}, n, m, h->T, 1, n, false); // per-read normalisation: every read is its own minibatch
    if (rc) return rc;
    hipLaunchKernelGGL(k_norm_pool<SigF32>, dim3(n), dim3(256), (size_t)NP_TILE * h->ds * 4, st, SigF32{dsig}, m, h->T, h->off, h->ds, h->L, h->Lp, 1, mbs,
                       h->down.as<float>(), h->nvalid.as<int32_t>(), (const int64_t *)drng, dlen);
    hipLaunchKernelGGL(k_cumsum, dim3((n + 63) / 64), dim3(64), 0, st, h->down.as<float>(), h->nvalid.as<int32_t>(), h->Lp, n, h->nck,
                       h->ck.as<double2>(), h->tail.as<double2>(), 1);
    hipLaunchKernelGGL(k_cumsum_gather, dim3((n + 63) / 64), dim3(64), 0, st, h->down.as<float>(), h->nvalid.as<int32_t>(), h->Lp, n, h->nck,
                       h->ck.as<double2>(), h->tail.as<double2>());
    hipLaunchKernelGGL(k_gains<1>, dim3((n + GAINS_WPB - 1) / GAINS_WPB), dim3(64 * GAINS_WPB), 0, st, h->down.as<float>(), h->nvalid.as<int32_t>(), h->Lp, h->nck,
                       h->ck.as<double2>(), h->tail.as<double2>(), h->adapter_idx.as<int32_t>(), 1, mbs, h->trace.as<double>(),
                       h->bmax.as<double>(), h->bmin.as<double>(), h->nsum, h->t1.as<int2>(), 1, h->pk.as<int32_t>(), h->npk.as<int32_t>(), h->Lp / 2 + 1, (double *)nullptr, n, 5);
    int grid = n < h->pslots ? n : h->pslots;
    hipLaunchKernelGGL(k_polya_peak, dim3(grid), dim3(64), (size_t)(((h->Lp / 2 + 1) + 8) / 16 + 2) * 4, st, h->trace.as<double>(), h->nvalid.as<int32_t>(), h->Lp,
                       h->bmax.as<double>(), h->bmin.as<double>(), h->nsum, h->adapter_idx.as<int32_t>(), n, 1, mbs,
                       h->pk.as<int32_t>(), h->mk.as<uint32_t>(), h->polya_idx.as<int32_t>(), h->npk.as<int32_t>());
    hipLaunchKernelGGL(k_refine_out, dim3((n + 255) / 256), dim3(256), 0, st, mbs, h->nvalid.as<int32_t>(), h->polya_idx.as<int32_t>(),
                       (const int64_t *)drng, n, h->ds, dout, dstat);
    HIPCHK(hipMemcpyAsync(polya_out, dout, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(status_out, dstat, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return ADP_OK;
}

int adp_open_pores_arena(adp_handle *h, int32_t *out, uint64_t cap, uint64_t *used)
{
    if (!h || !used || (cap && !out)) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    *used = h->op_last_used;
    const uint64_t n = h->op_last_used < cap ? h->op_last_used : cap;
    if (n) {
        HIPCHK(hipMemcpyAsync(out, h->op_arena.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return ADP_OK;
}

int adp_synth_fill(adp_handle *h, float *dev_signals, const int32_t *dev_full_len, int n, int m, uint32_t seed,
                   uint32_t first_read, int decorate)
{
    if (!h || !dev_signals || n < 1 || m < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    const double ih_sd = std::sqrt(8.0 * (256.0 * 256.0 - 1.0) / 12.0);
    SynthConst k;
    k.sd_adapter = (float)(7.0 / ih_sd); k.sd_polya = (float)(2.5 / ih_sd); k.sd_rna = (float)(3.0 / ih_sd);
    k.sd_level = (float)(14.0 / ih_sd);
    hipLaunchKernelGGL(k_synth, dim3((m + 255) / 256, n), dim3(256), 0, h->stream, dev_signals, dev_full_len, n, m, seed, first_read,
                       decorate, k);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return ADP_OK;
}

int adp_dev_alloc(adp_handle *h, uint64_t bytes, void **out)
{
    if (!h || !out) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMalloc(out, bytes));
    return ADP_OK;
}
int adp_dev_free(adp_handle *h, void *p)
{
    if (!h) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipFree(p));
    return ADP_OK;
}
int adp_memcpy_h2d(adp_handle *h, void *dst, const void *src, uint64_t bytes)
{
    if (!h) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ADP_OK;
}
// grid = (ceil(m / 1024), n); block = 256: four samples per thread
__global__ void __launch_bounds__(256) k_calibrate_i16(const int16_t *__restrict__ raw, const int32_t *__restrict__ full_len,
                                                        const float *__restrict__ scale, const float *__restrict__ offset, int m,
                                                        float *__restrict__ out)
{
    const int r = blockIdx.y;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= m) return;
    const int len = full_len[r] < m ? full_len[r] : m;
    const float sc = scale[r], of = offset[r];
    const int16_t *src = raw + (size_t)r * m + i0;
    float *dst = out + (size_t)r * m + i0;
    const float nanv = __builtin_nanf("");
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (i0 + j < m) dst[j] = (i0 + j < len) ? sc * ((float)src[j] + of) : nanv;
}

int adp_calibrate_i16(adp_handle *h, const int16_t *raw, const int32_t *full_len, const float *scale, const float *offset,
                      int n_reads, int m, float *signals_out)
{
    if (!h || !raw || !full_len || !scale || !offset || !signals_out || n_reads < 1 || m < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_calibrate_i16, dim3((m + 1023) / 1024, n_reads), dim3(256), 0, h->stream, raw, full_len, scale, offset, m, signals_out);
    HIPCHK(hipGetLastError());
    return ADP_OK;
}

int adp_expand_ragged(adp_handle *h, const void *packed, int is_int16, const int64_t *offsets, const int32_t *full_len,
                      const float *scale, const float *offset, int n_reads, int m, float *signals_out)
{
    if (!h || !packed || !offsets || !full_len || !signals_out || n_reads < 1 || m < 1 || (is_int16 && (!scale || !offset))) {
        g_err = "bad argument"; return ADP_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(h->device));
    const dim3 grid((m + 1023) / 1024, n_reads);
    if (is_int16)
        hipLaunchKernelGGL(k_expand_ragged<int16_t>, grid, dim3(256), 0, h->stream, (const int16_t *)packed, offsets, full_len, scale, offset, m, signals_out);
    else
        hipLaunchKernelGGL(k_expand_ragged<float>, grid, dim3(256), 0, h->stream, (const float *)packed, offsets, full_len,
                           (const float *)nullptr, (const float *)nullptr, m, signals_out);
    HIPCHK(hipGetLastError());
    return ADP_OK;
}

int adp_expand_ragged_i16(adp_handle *h, const int16_t *packed, const int64_t *offsets, const int32_t *full_len, int n_reads, int m,
                          int16_t *raw_out)
{
    if (!h || !packed || !offsets || !full_len || !raw_out || n_reads < 1 || m < 1) { g_err = "bad argument"; return ADP_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_expand_ragged_raw, dim3((m + 1023) / 1024, n_reads), dim3(256), 0, h->stream, packed, offsets, full_len, m, raw_out);
    HIPCHK(hipGetLastError());
    return ADP_OK;
}

int adp_host_alloc(adp_handle *h, uint64_t bytes, void **out)
{
    if (!h || !out) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return ADP_OK;
}
int adp_host_free(adp_handle *h, void *p)
{
    if (!h) return ADP_ERR_INVALID;
    HIPCHK(hipHostFree(p));
    return ADP_OK;
}
int adp_memcpy_h2d_async(adp_handle *h, void *dst, const void *src_pinned, uint64_t bytes)
{
    if (!h) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, h->stream3));
    return ADP_OK;
}
int adp_copy_mark(adp_handle *h, int slot)
{
    if (!h || slot < 0 || slot >= 16) return ADP_ERR_INVALID;
    if (!h->ev_copy[slot]) HIPCHK(hipEventCreateWithFlags(&h->ev_copy[slot], hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->ev_copy[slot], h->stream3));
    return ADP_OK;
}
int adp_copy_wait(adp_handle *h, int slot)
{
    if (!h || slot >= 16) return ADP_ERR_INVALID;
    if (slot < 0) { HIPCHK(hipStreamSynchronize(h->stream3)); return ADP_OK; }
    if (!h->ev_copy[slot]) return ADP_ERR_INVALID;
    HIPCHK(hipEventSynchronize(h->ev_copy[slot]));
    return ADP_OK;
}
int adp_memcpy_d2h(adp_handle *h, void *dst, const void *src, uint64_t bytes)
{
    if (!h) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ADP_OK;
}

int adp_kernel_times(adp_handle *h, const char **names_out, float *ms_out, int cap)
{
    if (!h) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    int k = 0;
    const adp_handle *src[1 + ADP_MAX_LANES] = {h};
    int ns = 1;
    if (h->last_grouped) for (int i = 0; i < ADP_MAX_LANES; i++) if (h->lane[i]) src[ns++] = h->lane[i];
    for (int q = 0; q < ns; q++) {
        for (const ProfEntry &e : src[q]->prof) {
            if (k >= cap) break;
            float ms = 0.f;
            HIPCHK(hipEventSynchronize(e.b));
            HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
            names_out[k] = e.name;
            ms_out[k] = ms;
            k++;
        }
    }
    return k;
}

int adp_debug_fetch(adp_handle *h, int what, void *host_out, uint64_t bytes)
{
    if (!h || !host_out) return ADP_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    const void *src = nullptr;
    switch (what) {
    case 0: {
        if (h->last_grouped) {
            if (bytes > (uint64_t)h->last_nmb * 32) return ADP_ERR_INVALID;
            HIPCHK(hipMemcpyAsync(host_out, h->mbparams.p, bytes, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            return ADP_OK;
        }
        if (h->ghist.cap < (size_t)h->last_nmb * 32) return ADP_ERR_INVALID;
        hipLaunchKernelGGL(k_mb_params_out, dim3((h->last_nmb + 255) / 256), dim3(256), 0, h->stream, h->mbs.as<MbState>(),
                           h->last_nmb, h->ghist.as<double>());
        src = h->ghist.p; break; }
    case 1: src = h->nvalid.p; break;
    case 2: src = h->down.p; break;
    case 3: src = h->trace.p; break;
    case 4: src = h->adapter_idx.p; break;
    case 5: src = h->polya_idx.p; break;
    case 6: { int32_t lp = h->Lp; if (bytes < 4) return ADP_ERR_INVALID; memcpy(host_out, &lp, 4); return ADP_OK; }
    case 7: src = h->t1.p; break;
    case 9: src = h->have_series.p; if (bytes > h->have_series.cap) return ADP_ERR_INVALID; break; // 1: the read's moving-window series were prepared by a series kernel
    case 8: { if (bytes < 64 || bytes > sizeof(unsigned long long) * ADP_NDBG) return ADP_ERR_INVALID;
              HIPCHK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dbg), bytes, 0, hipMemcpyDeviceToHost));
              static unsigned long long tally[ADP_NTALLY][8];
              HIPCHK(hipMemcpyFromSymbol(tally, HIP_SYMBOL(g_bs_tally), sizeof(tally), 0, hipMemcpyDeviceToHost));
              unsigned long long *o = (unsigned long long *)host_out;
              for (int j = 0; j < 5; j++) { o[j] = 0; for (int i = 0; i < ADP_NTALLY; i++) o[j] += tally[i][j]; }
              return ADP_OK; }
    default: return ADP_ERR_INVALID;
    }
    // (the stage buffers belong to the plain pipeline: a grouped call keeps them in its lanes, a handle that has not run the LLR path has none)
    if (!src || (what != 0 && h->last_grouped)) { g_err = "no stage buffers of a plain (one-group) LLR call on this handle"; return ADP_ERR_INVALID; }
    HIPCHK(hipMemcpyAsync(host_out, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (what == 0) HIPCHK(hipMemsetAsync(h->ghist.p, 0, (size_t)h->last_nmb * 32, h->stream));
    return ADP_OK;
}

} // extern "C"
