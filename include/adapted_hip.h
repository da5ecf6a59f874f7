/*
 * adapted_hip.h -- C ABI of libadapted_hip.so: the MI355X (gfx950) implementation of the
 * `adapted detect` signal-segmentation hot path.
 *
 * Drop-in boundary.  The reference (KleistLab/ADAPTed v0.2.4) exposes this path as Python
 * operators over one dense minibatch; each entry point below replaces one of them
 * (file:line under the reference tree):
 *
 *   adp_detect_llr          <- combined_detect_llr2(batch f32[N,m], full_lens i32[N], spc)
 *                              adapted/detect/combined.py:122-227   (called from
 *                              adapted/file_proc.py:230-235), which itself wraps the only
 *                              native code of the reference, adapted/detect/_c_llr.pyx:
 *                              c_llr_trace :202-236, c_llr_trace_gains :176-199, _gains :67-88
 *   adp_detect_start_peak   <- combined_detect_start_peak(...)   adapted/detect/combined.py:312-355
 *                              (detect_rna_start_peak, adapted/detect/start_peak.py:7-119)
 *   adp_detect_cnn          <- combined_detect_cnn(...)          adapted/detect/combined.py:230-250 (up to its short-read
 *                              fallback :251-301 = adp_llr_refine_polya + adp_validate_candidates)
 *   adp_cnn_prepare         <- prepare_data(...)                 adapted/detect/cnn.py:70-82
 *   adp_cnn_forward         <- BoundariesCNN.forward / cnn_score adapted/detect/cnn.py:16-52, 85-98 (hand-written conv
 *                              stack; default: float16 matrix cores on split float32 operands at float32 accuracy,
 *                              adapted_amd/csrc/cnn_conv_split.h, with the exact float32 MFMA kernels of cnn_conv.h behind
 *                              it for activations outside the float16 range -- or for every call when the handle was
 *                              created with ADP_CNN_CONV=f32 in the environment; PyTorch is not involved)
 *   adp_cnn_predict         <- cnn_predict + cnn_detect scaling  adapted/detect/cnn.py:101-182
 *   adp_validate_candidates <- the validate_boundaries loop of combined_detect_cnn
 *                              adapted/detect/combined.py:243-305
 *   adp_cfg                 <- SigProcConfig                     adapted/config/sig_proc.py:161-221
 *   adp_row                 <- DetectResults                     adapted/container_types.py:23-92
 *
 * Conventions: plain C, caller-allocated outputs, no exceptions.  Every function returns 0
 * or a negative ADP_ERR_* code.  A handle owns one device, its HIP streams and its workspace
 * (allocated by the first call that needs it, for the reads of that call, and kept: a call may
 * therefore return ADP_ERR_HIP for an allocation that adp_create did not attempt); it is not
 * thread-safe, distinct handles are independent.  Every call returns with its work finished.  `signals` is the reference's
 * minibatch layout (adapted/file_proc.py:143-190): row-major float32 [n_reads, m], NaN from
 * each read's end to m; full_len[i] is the read's true length (may exceed m).  A call may
 * carry several minibatches: reads [k*minibatch, (k+1)*minibatch) form minibatch k, and the
 * LLR path normalises with ONE median/MAD per minibatch exactly as the reference does
 * (adapted/detect/normalize.py:15-22 on the 2-D array).
 */
#ifndef ADAPTED_HIP_H
#define ADAPTED_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADP_ABI_VERSION 3

/* error codes */
#define ADP_OK 0
#define ADP_ERR_INVALID (-1)       /* bad argument */
#define ADP_ERR_HIP (-2)           /* a HIP runtime call failed (see adp_last_error) */
#define ADP_ERR_CAPACITY (-3)      /* n_reads / m exceed what the handle was created for */
#define ADP_ERR_UNSUPPORTED (-4)   /* config outside the implemented surface */
/* per-minibatch status (mb_status[]): the reference raises and drops the minibatch */
#define ADP_MB_OK 0
#define ADP_MB_MAD_ZERO 1          /* "MAD normalization failed: scale is 0" normalize.py:56-59 */
#define ADP_MB_EMPTY_TRACE 2       /* a read has no valid pooled block: np.argmin of an empty
                                      trace, adapted/detect/llr.py:136 */

/* memory-kind flags */
#define ADP_IN_DEVICE 1            /* signals / full_len are device pointers */
#define ADP_OUT_DEVICE 2           /* rows_out is a device pointer */
#define ADP_WITH_START_PEAK 4      /* LLR path: also fill the start_peak_* columns (extension) */
#define ADP_TOPK_NONE 8            /* adp_validate_candidates: polya_end_topk is None (k must be 1) */
#define ADP_BOUNDS_HOST 16         /* adp_validate_candidates: `bounds` is host memory even though ADP_IN_DEVICE is set */
#define ADP_TAILS_NAN 32           /* adp_detect_llr: the caller guarantees that row r is NaN from min(full_len[r], m) on (the
                                      padding of adapted/file_proc.py:170-174): the streaming passes then stop at the read's
                                      end instead of reading the padding.  Without the flag the rows are taken as they are. */

/* SigProcConfig, flattened.  Ranges are [lo, hi] with -inf/+inf for "None". */
typedef struct adp_cfg {
    int32_t min_obs_adapter, max_obs_adapter, min_obs_polya, downscale_factor, max_obs_trace;
    int32_t primary_method;            /* 0 llr, 1 cnn, 2 start_peak: names the *_adapter_end columns */
    double sig_norm_outlier_thresh;
    double adapter_peak_prominence, adapter_peak_rel_height;
    int32_t adapter_peak_width;
    int32_t mvs_detect_check, mvs_detect_overwrite, search_window;
    int32_t pA_mean_window, pA_var_window, median_shift_window, polyA_window;
    double pA_mean_range[2], pA_var_range[2], median_shift_range[2];
    double polyA_med_range[2], polyA_local_range[2], pA_mean_adapter_med_scale_range[2];
    int32_t detect_open_pores, real_signal_check, mean_window, max_obs_local_range;
    double mean_start_range[2], mean_end_range[2], local_range[2], adapter_mad_range[2];
    int32_t detect_med_shift, med_shift_window;
    double med_shift_range[2];
    int32_t sp_downscale_factor, start_peak_max_idx, sp_offset1, sp_offset2;
    double open_pore_pa;
    int32_t polya_cand_k, fallback_to_llr_short_reads;
} adp_cfg;

/* numeric columns of a result row, in DetectResults field order */
enum adp_col {
    ADP_C_SIGNAL_LEN, ADP_C_PRELOADED,
    ADP_C_ADAPTER_START, ADP_C_ADAPTER_END, ADP_C_ADAPTER_LEN, ADP_C_ADAPTER_MEAN, ADP_C_ADAPTER_STD,
    ADP_C_ADAPTER_MED, ADP_C_ADAPTER_MAD,
    ADP_C_POLYA_START, ADP_C_POLYA_END, ADP_C_POLYA_LEN, ADP_C_POLYA_MEAN, ADP_C_POLYA_STD,
    ADP_C_POLYA_MED, ADP_C_POLYA_MAD,
    ADP_C_RNA_START, ADP_C_RNA_LEN, ADP_C_RNA_MEAN, ADP_C_RNA_STD, ADP_C_RNA_MED, ADP_C_RNA_MAD,
    ADP_C_SP_IDX, ADP_C_SP_PA, ADP_C_SP_NEXT_IDX, ADP_C_SP_NEXT_PA, ADP_C_SP_OPEN_PORE_IDX,
    ADP_C_MED_SHIFT, ADP_C_PRIMARY_ADAPTER_END, ADP_C_PRIMARY_POLYA_END,
    ADP_C_MVS_MEAN, ADP_C_MVS_VAR, ADP_C_MVS_POLYA_MED, ADP_C_MVS_LOCAL_RANGE, ADP_C_MVS_MED_SHIFT,
    ADP_C_REAL_MEAN_START, ADP_C_REAL_MEAN_END, ADP_C_REAL_LOCAL_RANGE,
    ADP_C_MVS_ADAPTER_END,             /* mvs_detect_overwrite: position found by the MVS scan, 0 if none (combined.py:523) */
    ADP_NCOL
};

/* fail codes <-> the reference's fail_reason strings (adapted/detect/combined.py:396-580) */
enum adp_fail {
    ADP_F_NONE = 0,
    ADP_F_NO_ADAPTER = 1,       /* "No adapter detected (primary)" */
    ADP_F_ADAPTER_MAD = 2,      /* "adapter MAD check failed" */
    ADP_F_OPEN_PORE = 3,        /* "Open pore too close to boundary" */
    ADP_F_REAL_RANGE = 4,       /* "Real signal check failed" */
    ADP_F_NO_POLYA = 5,         /* "No polya detected (primary)" */
    ADP_F_MVS_NOT_ENOUGH = 6,   /* "MVS polya check failed: not enough signal" */
    ADP_F_MVS_CHECKS = 7,       /* "MVS polya check failed: " + names from mvs_fail_mask */
    ADP_F_MED_SHIFT = 8,        /* "Median shift check failed" */
    /* 9..14: the reference raised inside the per-read try block; the row is all-None */
    ADP_F_EXC_TOPK_NONE = 9,    /* "'NoneType' object is not iterable" (combined.py:464) */
    ADP_F_EXC_SLICE = 10,       /* "slice indices must be integers or None or have an __index__ method" */
    ADP_F_EXC_MOVE_WINDOW = 11, /* bottleneck: moving window larger than the slice */
    ADP_F_EXC_PA_RANGE = 12,    /* "pA_mean_range is not specified" (combined.py:462) */
    ADP_F_EXC_EMPTY_TRACE = 13, /* "attempt to get argmin of an empty sequence" (CNN fallback, llr.py:136) */
    ADP_F_EXC_MAD_ZERO = 14,    /* "MAD normalization failed: scale is 0" (CNN fallback, normalize.py:56-59) */
    ADP_F_NO_ADAPTER_MVS = 15   /* "No adapter detected in range (mvs_detect)" (mvs_detect_overwrite, combined.py:540) */
};
#define ADP_F_IS_EXCEPTION(code) ((code) >= 9 && (code) <= 14)
/* adp_row.mvs_fail_mask bit 8: mvs_llr_polya_end_to_early_stop (mvs_detect_overwrite moved the adapter end past the
 * poly(A) end; the reference then takes Boundaries.trace_early_stop_pos, None on every v0.2.4 path: polya_end is None) */
#define ADP_MVS_TO_EARLY_STOP 256

#define ADP_MAX_CAND 16
#define ADP_MAX_OPEN_PORES 16

/* one read's result: fixed width so that rows can be gathered across GPUs as bytes */
typedef struct adp_row {
    double col[ADP_NCOL];      /* ints exactly, float32 statistics widened exactly */
    uint64_t present;          /* bit c set <=> col[c] is not None */
    int32_t success;
    int32_t fail_code;         /* enum adp_fail */
    int32_t mvs_fail_mask;     /* bit0 mean, bit1 var, bit2 med, bit3 range, bit4 shift failed; bit8 ADP_MVS_TO_EARLY_STOP */
    int32_t start_peak_type;   /* 0 None, 1 "open pore in adapter", 2 "potential concatemer adapter-only read" */
    int32_t n_cand;            /* polya_candidates length; -1 <=> None */
    int32_t n_open_pores;      /* open_pores length (may exceed ADP_MAX_OPEN_PORES: see open_pores_more); -1 <=> None */
    int64_t cand[ADP_MAX_CAND];
    int32_t open_pores[ADP_MAX_OPEN_PORES]; /* the first ADP_MAX_OPEN_PORES positions */
    int32_t open_pores_more;   /* n_open_pores > ADP_MAX_OPEN_PORES: the WHOLE list lies at this offset of the call's
                                  open-pore arena (adp_open_pores_arena); -1 otherwise */
    int32_t reserved_;
} adp_row;

typedef struct adp_handle adp_handle;

int adp_abi_version(void);
int adp_sizeof_cfg(void);
int adp_sizeof_row(void);
const char *adp_last_error(void);

/* Number of visible HIP devices, or a negative error. */
int adp_device_count(void);

/* Create a handle on `device` able to process up to max_reads reads of m samples per call. */
int adp_create(int device, const adp_cfg *cfg, int max_reads, int m, adp_handle **out);
int adp_destroy(adp_handle *h);
int adp_set_config(adp_handle *h, const adp_cfg *cfg);
/* Layout of the LLR path.  ADP_LAYOUT_MINIBATCH (default): combined_detect_llr2, adapted/detect/combined.py:122-227.
 * ADP_LAYOUT_SINGLE_READ: combined_detect_llr, adapted/detect/combined.py:39-119 (API only in v0.2.4) -- every read
 * normalised on its own (adp_detect_llr with minibatch = 1), pooled from sample 0 with offset_head = 5 +
 * min_obs_adapter // ds, min_obs_adapter still added to the positions, no poly(A) search behind a candidate at index 0,
 * the ragged last pooled block filled up with zeros at the READ's end. */
#define ADP_LAYOUT_MINIBATCH 0
#define ADP_LAYOUT_SINGLE_READ 1
int adp_set_layout(adp_handle *h, int layout);
/* The handle's main HIP stream (hipStream_t as void*): staging copies, single-group calls and the non-LLR entry points are
 * ordered on it; asynchronous helpers (adp_calibrate_i16, adp_expand_ragged*) enqueue on it and a following detect call is
 * ordered behind them. */
void *adp_stream(adp_handle *h);
int adp_synchronize(adp_handle *h);

/* LLR primary + validation (+ optional start-peak columns) over n_reads reads.
 * rows_out: adp_row[n_reads]; mb_status: int32[ceil(n_reads/minibatch)] (host, may be NULL).
 * Rows of a dropped minibatch are zeroed with fail_code = 0 and success = 0.
 * Optional grouped execution (environment, read per call): ADP_GROUPS=k (k > 1; 0 = automatic) cuts a call that carries two or
 * more minibatches into about k groups of whole minibatches that run software-pipelined over ADP_LANES (default 2, <= 4)
 * internal streams, each with a workspace for ONE group -- while one group streams the signal for its normalisation, pooling
 * or partition statistics, its neighbour runs the float64 gains and the peak picking on its pooled trace; ADP_STAGGER=bits
 * orders phase p (bit 0 streaming-in, 1 gains, 2 validation) of consecutive groups.  Minibatches are independent of each
 * other, so the rows are the same bytes whatever the grouping (tests/test_gpu_grouped.py).  The default is one group: on
 * MI355X the overlap buys no time (profiles/r03_overlap_*), a smaller workspace is what grouping is for.
 * N1 (the minibatch's median / MAD): minibatches of at least ADP_N1_FUSED_MIN samples (default 2^22 = minibatch x max_obs_trace;
 * the reference's defaults are 1000 x 16 000) take ONE verified pass over the signal, smaller ones three or four; the values are
 * the exact np.nanmedian / MAD either way. */
int adp_detect_llr(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m,
                   int minibatch, int flags, adp_row *rows_out, int32_t *mb_status);

/* The same over RAW samples (an EXTENSION of the boundary: the reference's operators take calibrated
 * float32): raw = DEVICE int16 [n_reads, m] as the sequencer stores them, scale / offset = DEVICE float32 [n_reads] (pod5's
 * per-read calibration), full_len = DEVICE int32 [n_reads]; flags must hold ADP_IN_DEVICE.  Every kernel that touches the
 * signal forms pA = scale * (float32(adc) + offset) in registers (both operations rounded to float32, never fused --
 * bit-identical to adp_calibrate_i16's output) and treats samples at or beyond min(full_len, m) as the NaN padding of
 * adapted/file_proc.py:170-174, so rows are identical to adp_calibrate_i16 + adp_detect_llr while every streaming pass
 * moves 2 bytes per sample instead of 4.  m must be a multiple of 4. */
int adp_detect_llr_i16(adp_handle *h, const int16_t *raw, const int32_t *full_len, const float *scale, const float *offset,
                       int n_reads, int m, int minibatch, int flags, adp_row *rows_out, int32_t *mb_status);

/* Start-peak primary + validation. */
int adp_detect_start_peak(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads,
                          int m, int minibatch, int flags, adp_row *rows_out);

/* CNN head, C1: the network's input.
 * prepared_out: float32 [n_reads, Lc] with Lc = ceil((m - min_obs_adapter)/downscale_factor). */
int adp_cnn_prepare(adp_handle *h, const float *signals, int n_reads, int m, int flags, float *prepared_out);
/* C2, the conv net itself (BoundariesCNN adapted/detect/cnn.py:16-52; cnn_score :85-98), hand-written for gfx950
 * (adapted_amd/csrc/cnn_conv.h: the two 64 -> 64 layers on the float32 matrix cores, fixed accumulation order).
 * adp_cnn_set_weights: HOST pointers in the layout of the reference's state dict ("0.weight" [64,1,7], "0.bias" [64],
 * "2.weight" [64,64,7], "2.bias", "4.weight" [64,64,7], "4.bias", "6.weight" [64,2,7], "6.bias" [2]).
 * adp_cnn_forward: prepared = DEVICE float32 [n_reads, Lc] (adp_cnn_prepare's output), scores_out = DEVICE float32
 * [n_reads, 2, Lo], Lo = 3 * ((Lc - 1) / 3 + 1) - 2 -- what model(x) returns in the reference. */
int adp_cnn_set_weights(adp_handle *h, const float *w0, const float *b0, const float *w1, const float *b1, const float *w2,
                        const float *b2, const float *w3, const float *b3);
int adp_cnn_forward(adp_handle *h, const float *prepared, int n_reads, int Lc, float *scores_out);
/* C3, cnn_predict (adapted/detect/cnn.py:101-160) and the scaling of cnn_detect (:165-182) on the device
 * (adapted_amd/csrc/cnn_topk.h).  scores: DEVICE float32 [n, 2, Lo] (the conv net's output).  Reads [q * minibatch,
 * (q + 1) * minibatch) are one call of the reference: its find_peaks runs over that minibatch's FLATTENED scores and the
 * candidates of the i-th read with peaks go to row i of the minibatch (:150-158), both reproduced.
 * bounds_out: HOST int64 [n, 1 + max(k, 1)], k = cfg.polya_cand_k: adapter end and the k poly(A) candidates in samples
 * (index * downscale_factor + min_obs_adapter, a value equal to min_obs_adapter -> 0), i.e. what cnn_detect returns.
 * Exact ties between peaks closer than 5 samples: the later index counts as higher (scipy leaves it to an unstable sort). */
int adp_cnn_predict(adp_handle *h, const float *scores_dev, int n_reads, int minibatch, int Lo, int64_t *bounds_out);
/* The k > 1 part alone, behind given arg-maxes (tests): adapter_pos / polya_pos DEVICE int64 [n]; host outputs cand int32
 * [n, k] (positions inside the read, zero padded) and n_peaks int32 [n] (peaks of the read after the distance rule).
 * All n reads are one minibatch. */
int adp_cnn_topk(adp_handle *h, const float *scores_dev, const int64_t *adapter_pos_dev, const int64_t *polya_pos_dev,
                 int n_reads, int Lo, int k, int32_t *cand_out, int32_t *n_peaks_out);
/* combined_detect_cnn (adapted/detect/combined.py:230-309) in one call, up to its short-read fallback (the caller applies
 * that with adp_llr_refine_polya + adp_validate_candidates, :251-301): prepare_data -> conv net -> cnn_predict -> the
 * validate_boundaries loop.  Needs adp_cnn_set_weights.  bounds_out (HOST, may be NULL) as adp_cnn_predict. */
int adp_detect_cnn(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m, int minibatch, int flags,
                   adp_row *rows_out, int64_t *bounds_out);
/* Validate with explicit primary boundaries: bounds int64 [n_reads, 1 + k] = adapter_end, k poly(A)
 * candidates (0 terminates), exactly what cnn_detect_boundaries hands to validate_boundaries. */
int adp_validate_candidates(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m,
                            const int64_t *bounds, int k, int flags, adp_row *rows_out);

/* CNN fallback for short reads (adapted/detect/combined.py:251-301): per-read normalisation, LLR trace
 * (offsets 5/5) over [ranges[2i], ranges[2i+1]) and P4 on it.  polya_out[i] = new poly(A) end in samples or 0;
 * status_out[i] = 0 or the ADP_F_EXC_* code of the exception the reference would have raised. */
int adp_llr_refine_polya(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m,
                         const int64_t *ranges, int flags, int64_t *polya_out, int32_t *status_out);

/* The open-pore arena of the LAST detect / validate call on this handle: find_open_pores (adapted/detect/anomalies.py:15-35)
 * returns a list without a length limit, and the CSV prints all of it; rows with more than ADP_MAX_OPEN_PORES entries keep
 * their whole list here (row.open_pores_more = offset, row.n_open_pores = length).  ONE arena per API call, whatever the
 * number of minibatches (or internal groups) the call carries: the offsets of all its rows index the same array.  Copies
 * min(used, cap) int32 entries to the HOST buffer `out` and the number in use to *used (a call that finds the arena too
 * small grows it and runs again before it returns: a call never loses entries). */
int adp_open_pores_arena(adp_handle *h, int32_t *out, uint64_t cap, uint64_t *used);

/* Synthetic squiggles generated on the device (bit-identical to adapted_amd/synth.py).
 * dev_signals: device float32 [n, m]; dev_full_len: device int32 [n] or NULL (=> all m). */
int adp_synth_fill(adp_handle *h, float *dev_signals, const int32_t *dev_full_len, int n, int m,
                   uint32_t seed, uint32_t first_read, int decorate);

/* Device memory helpers so callers need no HIP binding of their own. */
int adp_dev_alloc(adp_handle *h, uint64_t bytes, void **out);
int adp_dev_free(adp_handle *h, void *p);
int adp_memcpy_h2d(adp_handle *h, void *dst, const void *src, uint64_t bytes);
int adp_memcpy_d2h(adp_handle *h, void *dst, const void *src, uint64_t bytes);
/* int16 ingestion (an EXTENSION of the boundary: the reference's operators take calibrated float32):
 * raw ADC samples int16 [n, m] + per-read calibration -> the float32 [n, m] NaN-padded minibatch the detect entry points
 * take, on the device, so that only 2 bytes per sample cross PCIe.  pA = scale * (float32(adc) + offset), both
 * operations rounded to float32 -- pod5's calibrate_signal_array; pod5 is not in this build's container, so the formula
 * is unpinned against it and stated here.  Samples at or beyond min(full_len, m) become NaN (minibatch layout B0,
 * adapted/file_proc.py:170-174).  All pointers are DEVICE pointers. */
int adp_calibrate_i16(adp_handle *h, const int16_t *raw, const int32_t *full_len, const float *scale, const float *offset,
                      int n_reads, int m, float *signals_out);

/* Ragged ingestion (extension, next to adp_calibrate_i16): the reads of a minibatch packed back to back -- read r's first
 * min(full_len[r], m) samples at packed[offsets[r] ...], float32 pA or (is_int16) raw ADC samples with per-read
 * calibration -- are laid out as the float32 [n, m] NaN-padded minibatch on the device.  Only the samples that exist
 * cross PCIe: with heavy-tailed read lengths most of the padded matrix is NaN (84 % for BASELINE configs[4]).
 * All pointers are DEVICE pointers; scale / offset may be NULL for float32 input. */
int adp_expand_ragged(adp_handle *h, const void *packed, int is_int16, const int64_t *offsets, const int32_t *full_len,
                      const float *scale, const float *offset, int n_reads, int m, float *signals_out);

/* Packed raw samples -> the raw int16 [n, m] matrix adp_detect_llr_i16 reads (zeros behind each read; DEVICE pointers). */
int adp_expand_ragged_i16(adp_handle *h, const int16_t *packed, const int64_t *offsets, const int32_t *full_len, int n_reads, int m,
                          int16_t *raw_out);

/* Streaming input (adapted_amd/pipeline.py): page-locked host staging memory, and a host-to-device copy on the handle's
 * COPY stream (neither the compute stream nor its side stream), so that the copy of the next minibatch overlaps the
 * detect call of the current one.  adp_copy_mark(slot) marks the copies issued so far (slot in [0, 16));
 * adp_copy_wait(slot) blocks the host until the copies marked by that slot have landed (slot < 0: all copies). */
int adp_host_alloc(adp_handle *h, uint64_t bytes, void **out);
int adp_host_free(adp_handle *h, void *p);
int adp_memcpy_h2d_async(adp_handle *h, void *dst, const void *src_pinned, uint64_t bytes);
int adp_copy_mark(adp_handle *h, int slot);
int adp_copy_wait(adp_handle *h, int slot);

/* The reference's native module (adapted/detect/_c_llr.pyx, imported at adapted/detect/llr.py:18) as a batched call:
 * `c_llr_trace(raw_signal, start, end, min_obs, border_trim, stride, adapter_early_stopping, adapter_early_stop_window,
 * adapter_early_stop_stride, polya_early_stopping, polya_early_stop_window, polya_early_stop_stride, return_c_c2)` (:202-236)
 * for n_reads float64 signals -- raw [n_reads, L], read r valid in [0, len[r]) -- with `_gains` (:67-88) and both early-stopping
 * forms (`_gains_w_early_stop` :91-122, `_gains_w_polya_early_stop` :125-173) behind the same dispatch (:190-197).
 *   gain_out [n_reads, L]   zeros outside the computed points, as np.zeros_like(c)
 *   c_io, c2_io [n_reads, L] np.cumsum(raw), np.cumsum(raw * raw): outputs (may be NULL), or -- with ADP_TRACE_FROM_SUMS, the
 *                           form of `c_llr_trace_gains` (:176-199) and `_gains` -- the INPUTS (raw is ignored, may be NULL)
 *   len, start, end         HOST int32 [n_reads] (checked here: 0 <= start <= end <= len <= L)
 *   flags                   ADP_IN_DEVICE: raw (or the sums) are device pointers; ADP_OUT_DEVICE: gain_out (and the sums when
 *                           they are outputs) are device pointers
 * Returns ADP_ERR_INVALID where the reference asserts (an early-stop stride that is not a multiple of `stride`) or would
 * index outside its arrays.  The product path does not go through here (adp_detect_llr has its own fused passes). */
typedef struct adp_trace_args {
    int32_t min_obs, border_trim, stride;
    int32_t adapter_early_stopping, adapter_early_stop_window, adapter_early_stop_stride;
    int32_t polya_early_stopping, polya_early_stop_window, polya_early_stop_stride;
} adp_trace_args;
#define ADP_TRACE_FROM_SUMS 64
int adp_c_llr_trace(adp_handle *h, const double *raw, const int32_t *len, const int32_t *start, const int32_t *end, int n_reads,
                    int L, const adp_trace_args *args, int flags, double *gain_out, double *c_io, double *c2_io);

/* Per-kernel timing of the LAST detect call, measured with HIP events on the handle's stream.
 * Enable with adp_set_profiling(h, 1).  names_out: up to cap pointers to static strings. */
int adp_set_profiling(adp_handle *h, int on);
int adp_kernel_times(adp_handle *h, const char **names_out, float *ms_out, int cap);

/* Debug/inspection of intermediate stages of the last LLR call (tests only).
 * what: 0 norm params double[4*n_minibatch]; 1 n_valid int32[n]; 2 pooled float32[n*Lp];
 *       3 trace float64[n*Lp] (pass-2 trace after a full call); 4 adapter idx int32[n];
 *       5 polya idx int32[n]; 6 Lp (int32[1]); 7 the T1 pairs int32[2*n];
 *       8 process-wide tallies uint64[8..48] (cumulative): 0 large S1 segments, 1 MAD brackets that held, 2 generic median
 *         selections, 3 MADs not predicted, 4 MAD bracket overflows (0-4 are kept on 256 cache lines by workgroup and summed
 *         here: per-workgroup device atomics on one address serialise chip-wide), 5 single-pass N1 attempts, 6 / 7 their
 *         median / MAD misses, 22 / 23 N1 heavy keys / samples; the rest: phase cycles of -DADP_PHASE_TIMING builds */
int adp_debug_fetch(adp_handle *h, int what, void *host_out, uint64_t bytes);
/* Run the LLR pipeline only up to a stage (1 N1, 2 pool, 3 cumsum, 4 gains1, 5 adapter, 6 gains2, 7 polya) */
int adp_debug_llr_upto(adp_handle *h, const float *signals, const int32_t *full_len, int n_reads, int m,
                       int minibatch, int flags, int stage);

/* fdiv_shared (adapted_amd/csrc/common.h: float32 division by a divisor shared by many numerators) against the IEEE
 * division, for `count` consecutive float bit patterns from first_bits on and their negatives (tests only) */
int adp_debug_divcheck(adp_handle *h, float d, uint32_t first_bits, uint32_t count, uint64_t *mismatches_out);
/* log_cr (adapted_amd/csrc/log_cr.h), the logarithm of the gains kernels, applied to n host doubles (tests only) */
int adp_debug_log(adp_handle *h, const double *host_in, double *host_out, int n);

#ifdef __cplusplus
}
#endif
#endif /* ADAPTED_HIP_H */
