/*
 * goofer_hip.h — C ABI of libgoofer_hip.so, the MI355X (gfx950) backend for GOOFER's per-frame
 * source-filter resampler loop.
 *
 * The reference is pure Python: its "FFI" for this path is the set of numpy calls made by
 * GOOFER.py / SillySampler.py.  Each entry point below names the reference function it replaces
 * (file:line under the reference repo).  Conventions:
 *   - plain C, no torch types: device pointers are raw (e.g. torch.Tensor.data_ptr()), the stream is
 *     a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); NULL = default stream
 *   - every call is asynchronous on that stream; the caller owns all buffers; the library only
 *     allocates per-plan tables and scratch owned by the handle (goofer_reserve grows it, never
 *     inside a timed region if the caller reserved up front)
 *   - return 0 on success, negative GOOFER_E* on failure; goofer_last_error() gives the text
 *   - ragged batches are CSR: sample_off[n_notes+1], frame_off[n_notes+1] (int64, device memory)
 *   - device matrices are [frames x bins] row-major (the reference is [bins, frames]) with an
 *     explicit row stride `ld` in elements (ld >= n_bins; 516 keeps fp32 rows 16-byte aligned)
 *   - one handle per device per host thread; no global state
 */
#ifndef GOOFER_HIP_H
#define GOOFER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOOFER_OK 0
#define GOOFER_EINVAL (-1)   /* bad argument / unsupported geometry */
#define GOOFER_EHIP (-2)     /* a HIP runtime call failed */
#define GOOFER_ENOPLAN (-3)  /* goofer_plan() has not been called */
#define GOOFER_ENOMEM (-4)

typedef struct goofer_ctx goofer_ctx;

/* Per-note scalars of one synthesize() call — the keyword arguments of GOOFER.py:971-983 that the
 * sampler actually varies (SillySampler.py:1006-1035), plus the V/B/U mix of :1142-1151. */
typedef struct {
    float pitch_shift;          /* f0 *= pitch_shift (fp32)                       GOOFER.py:995  */
    float formant_shift;        /* 'g' flag: uniform warp ratio, 1 = off          GOOFER.py:1016 */
    double f_shift[4];          /* 'fa'..'fd': per-formant ratios, all 1 = off    GOOFER.py:1004 */
    float uv_strength;          /* default 0.75                                    GOOFER.py:1181 */
    float breath_strength;      /* default 0.1                                     GOOFER.py:1180 */
    float normalize;            /* 0..1 exponent of 1/peak                         GOOFER.py:1208 */
    int32_t apply_brightness;   /* default 1                                       GOOFER.py:1131 */
    int32_t cut_below_f0;       /* cut_subharm_below_f0, default 1                 GOOFER.py:1113 */
    float mix_harm, mix_breath, mix_unvoiced, volume;   /* V, (B+100)/100, (U+100)/100, volume */
    uint32_t seed[2];           /* per-note Philox key (lo, hi), XORed with the batch seed: a note's  */
                                /* noise never depends on where it sits in a batch                    */
    float vol_jitter_harm;      /* 'sr': volume_jitter_strength_harm, 0 = off     GOOFER.py:1185-1191 */
    float vol_jitter_breath;    /*       volume_jitter_strength_breath                                */
    float subharm_weight;       /* 'sg': +12 st pulse layer weight, 0 = off       GOOFER.py:1076-1097 */
    /* The two f0 jitter strengths are python floats in the reference and scale a factor that is rounded into the fp32 f0   */
    /* the pulse onsets are derived from: they stay fp64 (an fp32 strength moves a fifth of the f0 samples by an ulp).    */
    double f0_jitter;           /* 'sh': f0_jitter_strength, 0 = off              GOOFER.py:1069-1071 */
    double subharm_f0_jitter;   /* jitter of the f0 the sub-harmonic layer tracks; applied IN PLACE after the pulse   */
                                /* train, so the later per-frame f0 picks see it too (the reference's aliasing) :1078-1080 */
} goofer_note_params;

/* One ragged batch of notes for goofer_synth_batch.  All pointers are device memory. */
typedef struct {
    int32_t n_notes;
    int32_t n_bins;             /* n_fft/2 + 1 */
    int32_t ld;                 /* row stride of env / phi in floats */
    int32_t mix_only;           /* 1: only `mix` (and `rec`) are final; harm / uv / bre are scratch: the stems   */
                                /* before the peak gain, and (round 6) UNDEFINED wherever the noise walker found  */
                                /* a stem exactly zero over a whole hop and did not store it                      */
    int64_t total_frames;       /* frame_off[n_notes]: sum over notes of 1 + n_samples/hop */
    int64_t total_samples;      /* sample_off[n_notes] */
    int64_t total_env_rows;     /* env_off[n_notes] */
    const int64_t *sample_off;  /* [n_notes+1] */
    const int64_t *frame_off;   /* [n_notes+1] synthesis frames (pulse STFT frames)            */
    const int64_t *env_off;     /* [n_notes+1] envelope rows; a note's rows are truncated or   */
                                /*             edge-repeated to its frame count (GOOFER.py:1115-1119) */
    const float *env;           /* [total_env_rows x ld] spectral envelope                     */
    const double *formants;     /* [total_env_rows x 4] F1..F4 in Hz, or NULL when no f_shift  */
    const float *f0;            /* [total_samples] Hz, 0 where unvoiced                        */
    const float *mask;          /* [total_samples] voicing mask                                */
    const float *phi;           /* [total_frames x ld] random phases, or NULL: on-device Philox */
    const float *env_noise;     /* [total_env_rows x ld] noise envelope (sigma-1.75 blur of env along bins,    */
                                /* GOOFER.py:993) when the caller has it already — gf.synthesize's time stretch  */
                                /* resamples it separately from the warped env — else NULL: blurred in-kernel    */
    const goofer_note_params *params;  /* [n_notes] */
    uint64_t seed;              /* Philox key when phi == NULL */
    float transition_sigma;     /* noise_transition_smoothness of this call (default 100; the 'sa'  */
    float vol_jitter_speed;     /* layer uses 1) — one value per batch            GOOFER.py:1179 */
                                /* vol_jitter_speed: Hz of the volume_vibrato sinusoid (see volume_vibrato)       */
    /* jitter flags: standard-normal draws supplied by the caller (the reference takes them from the legacy  */
    /* global np.random stream), [total_samples] fp64 each, NULL when no note of the batch uses the flag      */
    const double *noise_f0;     /* for notes with f0_jitter > 0                   GOOFER.py:666        */
    const double *noise_vol_h;  /* harmonic volume jitter draw                    GOOFER.py:653        */
    const double *noise_vol_b;  /* breath volume jitter draw                                           */
    const double *noise_subharm;/* draw of subharm_f0_jitter (between the f0 and the volume draws), or NULL  :1079   */
    float f0_jitter_sigma;      /* sr / (6 * f0_jitter_speed)  samples            GOOFER.py:667        */
    float vol_jitter_sigma;     /* sr / (6 * volume_jitter_speed)                 GOOFER.py:654        */
    /* sub-harmonic pulse layer ('sg'): notes with params.subharm_weight > 0         GOOFER.py:1076-1097  */
    double subharm_ratio;       /* 2^(subharm_semitones / 12); 0 disables the layer for the batch         */
    double subharm_more[15];    /* further ratios when subharm_semitones is a list (0 = unused): one phase      */
                                /* tracker each, pulses summed before the joint max-normalisation  :672-736     */
    double subharm_vib_rate;    /* Hz                                              GOOFER.py:748-766       */
    double subharm_vib_depth;
    double subharm_vib_delay;   /* seconds of linear fade-in                                              */
    int32_t subharm_vibrato;    /* 0 / 1                                                                  */
    int32_t volume_vibrato;     /* 1: the volume jitter is a sinusoid, no noise draws needed  GOOFER.py:643-652  */
    int32_t unit_pitch_shift;   /* 1: the caller vouches params[i].pitch_shift == 1 for every note (the resampler  */
    int32_t no_warp;            /* path: pitch lives in the curve), so f0 needs no scaling pass          :995    */
                                /* no_warp = 1: the caller vouches that no note of the batch warps its envelope    */
                                /* (every f_shift == 1 and formant_shift == 1): kernels that could warp rows in    */
                                /* LDS need not reserve those rows (a note that warps anyway is rendered unwarped) */
    float *harm, *uv, *bre;     /* [total_samples] stems, gain-normalised like the reference   */
    float *rec;                 /* [total_samples] harm+uv+bre (reconstruct), may be NULL      */
    float *mix;                 /* [total_samples] (harm*V + bre*B + uv*U)*volume, may be NULL */
    const double *f0_64;        /* NULL, or [total_samples] the same f0 as a float64 array: what f0_interp IS in      */
                                /* gf.synthesize behind its time stretch (interp1d returns float64, GOOFER.py:1053).  */
                                /* The f0 jitters then multiply THIS array (`f0` becomes its float32 cast, which is   */
                                /* what the pulse train is handed, :1071-1074) and the sub-harmonic phase trackers    */
                                /* accumulate it (:1077-1097): on a float32 copy one event in ~10^5 lands a sample    */
                                /* off.  Final values (no pitch_shift is applied to it); read only with noise_f0 or   */
                                /* a sub-harmonic ratio set; not modified (the library works on a copy)                */
} goofer_batch;

/* One note's assembly plan (host-computed scalars, SillySampler.py:449-855).  "Logical" source rows /
 * samples are those of the feature arrays after the optional R1 reversal. */
typedef struct {
    int64_t knot_off;        /* first fp16 element of the note's source knot rows, [n_src_rows x K]      */
    int64_t edit_off;        /* first row of this note in the edited-row scratch                          */
    int64_t tap_off;         /* first row of this note in tap_idx / tap_w                                 */
    int64_t env_off;         /* first output envelope row (== fst_tracks row)                             */
    int64_t src_sample_off;  /* first element of the note's source voicing mask                           */
    int64_t out_sample_off;  /* first output sample                                                       */
    int64_t ylen;            /* source length in samples                                                  */
    int64_t bend_off;        /* first element of the note's pitch curve in goofer_assembly.bend           */
    double es_amount;        /* sharpen amount 5*|es|                                                     */
    double vel_factor;       /* 2^(1 - velocity/100)                                                      */
    double pitch_m;          /* MIDI note number                                                          */
    double pitch_t;          /* 't' flag / 100, added after bend/100 + pitch_m                            */
    double tick_dt;          /* 60 / (tempo * 96) seconds per pitch-bend tick                             */
    double fst[4];           /* formant-strength values (python floats in the reference)                  */
    int32_t K, lerp_plan, n_src_rows, reverse;
    int32_t row_lo, n_edit;  /* edited window: logical rows [row_lo, row_lo + n_edit)                     */
    int32_t tilt;            /* br tilt table index or -1                                                 */
    int32_t es_mode;         /* 0 off, 1 smooth, 2 sharpen                                                */
    int32_t es_taps_off, es_radius;
    int32_t fw_plan;         /* fw plan index or -1                                                       */
    int32_t n_out_rows, env_f64;
    int32_t n_out, n_pre, s_pre, s_tail, tail_len, want_samples, n_before_vel, pre_new;
    int32_t vel_active, force_voiced, n_bend, reserved;
    /* vocal fry ('vf' / 'vh' / 'vl', SillySampler.py:883-997); all ranges in output samples          */
    double fry_hz;           /* 'vh' base frequency (>= 1)                                                */
    int32_t fry_dir;         /* 0 off, +1 fry from the start, -1 fry towards the end                      */
    int32_t fry_const_lo, fry_const_hi;   /* f0 = fry_hz * (mask > 0)                                     */
    int32_t fry_glide_lo, fry_glide_hi;   /* f0 = (1 - w) * fry_hz * (mask > 0) + w * f0, w linspace      */
    int32_t fry_a, fry_b, fry_fade;       /* fry mask = 1 on [a, b) with linear fades of fry_fade samples */
    /* pitch dynamics ('pd', SillySampler.py:857-881)                                                     */
    int32_t pd_on;           /* write midi_curve - pd_base (fp32) to goofer_assembly.bend_out             */
    int32_t reserved4;
    double pd_base;          /* pitch_m + t/100                                                           */
} goofer_note_plan;

/* A batch of note assemblies.  All pointers are device memory. */
typedef struct {
    int32_t n_notes, n_bins, ld, sr, max_K, reserved;
    int64_t total_edit_rows, total_out_rows, total_samples;
    const goofer_note_plan *notes;   /* [n_notes] */
    const uint16_t *knots;           /* fp16 log-knot values, per note [n_src_rows x K]                   */
    const int32_t *lerp_idx;         /* [n_lerp_plans x n_bins] 2-tap lerp plans (GOOFER.py:84-90)        */
    const float *lerp_w0, *lerp_w1;
    const float *tilts;              /* [n_tilts x n_bins] br tilt curves (SillySampler.py:506-510)       */
    const double *es_taps;           /* concatenated Gaussian taps of the es flag                          */
    const int32_t *fw_lo, *fw_hi;    /* [n_fw x n_bins] fw plans (SillySampler.py:555-564)                */
    const double *fw_frac;
    const int32_t *tap_idx;          /* [total_out_rows x 4] logical source rows                           */
    const double *tap_w;             /* [total_out_rows x 4]                                               */
    const float *fst_tracks;         /* [total_out_rows x 4] sanitised + smoothed F1..F4 (Hz)              */
    const float *mask_src;           /* source voicing masks, concatenated                                 */
    const double *bend;              /* pitch curve per tick in MIDI semitones, concatenated: bend/100 + pitch_m */
                                     /* (+ t/100), fp64 like the reference builds it (SillySampler.py:838-846)    */
    float *edit_rows;                /* scratch [total_edit_rows x ld] (NULL: handle-owned)                */
    float *env_out;                  /* [total_out_rows x ld] assembled envelope                           */
    float *f0_out, *mask_out;        /* [total_samples]                                                    */
    float *bend_out;                 /* [total_samples] pitch-bend semitones of the 'pd' notes, or NULL    */
    int32_t any_fry, reserved5;      /* some note has fry_a < fry_b: run the envelope fry warp             */
    /* 'sj' growl layer (SillySampler.py:1061-1065): its f0 is f0_new * (0.5 * 2^noise) with f0_new still fp64, rounded */
    /* to fp32 ONCE inside synthesize.  f0_mul [total_samples] fp64 factors (anything on notes without the layer) and   */
    /* f0_mul_out [total_samples] = (float)(fp64 f0 * f0_mul), or both NULL.  Scaling the rounded f0_out instead moves    */
    /* two thirds of the layer's f0 samples by an ulp, enough to flip a pulse onset in ~3 % of such notes.              */
    const double *f0_mul;
    float *f0_mul_out;
} goofer_assembly;

/* One time-varying one-pole cascade (dynamic_butter_filter, SillySampler.py:95-174): `order` sections of
 * y = y' + a (x - y') (low-pass) or y = a (y' + x - x') (high-pass), each restarted from zero, the per-sample
 * coefficient following cutoff_factor * (5-tap box-smoothed f0 reference), clamped to [60|20 Hz, 0.45 sr]. */
typedef struct {
    int64_t src_off, dst_off;   /* element offsets into src / dst (equal offsets with src == dst: in place)  */
    int64_t f0_off;             /* element offset of the note's f0 reference                                 */
    int32_t n;                  /* samples                                                                   */
    int32_t order;              /* 1..12 (two chained order-6 calls are one order-12 cascade)                */
    int32_t highpass;           /* 0 low-pass, 1 high-pass                                                   */
    int32_t f0_mode;            /* 0: f0, 1: max(f0, 120), 2: ones (cutoff_factor is then the cutoff in Hz)  */
    float cutoff_factor;
    int32_t reserved;
} goofer_onepole_job;

/* Per-note settings of the sample-domain post chain (SillySampler.py:1037-1182). */
typedef struct {
    int64_t su_off, sj_off, sa_off;  /* first sample of the note in su_harm / sj_harm / sa_uv+sa_bre, -1: flag off */
    float su_gain;              /* 'su' / 100                                              :1037-1059        */
    float sj_mix;               /* 'sj' / 100                                              :1061-1081        */
    float sa_mix;               /* 'sa' / 100                                              :1153-1172        */
    float sd_strength;          /* 'sd' value                                              :1101-1112        */
    float tension;              /* 'st' / 100, in [-1, 1]                                  :1114-1140        */
    float pitch_dyn;            /* 'pd' / 100                                              :857-881, 1174-1182 */
    int32_t fry_a, fry_b, fry_fade;   /* fry mask range (see goofer_note_plan); a >= b: off :1083-1099       */
    int32_t reserved;
} goofer_post_note;

typedef struct {
    int32_t n_notes, reserved;
    int64_t total_samples;
    const int64_t *sample_off;          /* [n_notes+1] device                                                */
    const int64_t *sample_off_host;     /* [n_notes+1] the same offsets in HOST memory                       */
    const goofer_note_params *params;   /* [n_notes] device (mix_harm / mix_breath / mix_unvoiced / volume)  */
    const goofer_post_note *notes;      /* [n_notes] HOST memory: the library builds its job lists from it   */
    const float *f0, *mask;             /* [total_samples] assembled f0 / voicing mask                       */
    const float *bend;                  /* [total_samples] goofer_assembly.bend_out, NULL without 'pd'       */
    float *harm, *uv, *bre;             /* [total_samples] stems of the main synth call, edited in place     */
    float *su_harm, *sj_harm;           /* harmonic stems of the extra synth calls (compact, filtered in place) */
    const float *sa_uv, *sa_bre;        /* noise stems of the all-voiced 'sa' synth call (compact)           */
    float *mix;                         /* [total_samples] out; only notes with some post flag are rewritten */
} goofer_post;

/* ---- lifetime ------------------------------------------------------------------------------ */
int goofer_create(int device_id, goofer_ctx **out);
void goofer_destroy(goofer_ctx *ctx);
const char *goofer_last_error(const goofer_ctx *ctx);
const char *goofer_version(void);

/* Tables per (sr, n_fft, hop): sqrt-Hann window, bin freqs, boost, brightness curves, FFT twiddles
 * (GOOFER.py:12-46, 585-595).  n_fft: 512, 768, 1024, 1536, 2048 (radix plans; the stem walkers run at 1024 with hop 256, the
 * fused overlap-add at the three powers of two) or any other even size in [64, 2048] (Bluestein's chirp-z transform through
 * power-of-two transforms, one kernel per reference step); re-planning replaces the tables. */
int goofer_plan(goofer_ctx *ctx, int sr, int n_fft, int hop);

/* Pre-size handle-owned scratch for batches up to these totals (else grown on demand). */
int goofer_reserve(goofer_ctx *ctx, int64_t max_frames, int64_t max_samples, int64_t max_notes);

/* ---- single-kernel entry points (unit parity + roofline runs) ------------------------------ */

/* gf.stft (GOOFER.py:355-370): per note reflect-pad n_fft/2, frame, sqrt-Hann, rFFT.
 * x [total_samples] -> S [total_frames x ldc] complex64 (interleaved re,im; ldc in complex elements). */
int goofer_rfft_frames(goofer_ctx *ctx, const float *x, const int64_t *sample_off, const int64_t *frame_off,
                       int n_notes, int64_t total_frames, float *S, int ldc, void *stream);

/* gf.istft + _overlap_add (GOOFER.py:372-413): irFFT, windowed OLA normalised by the summed squared
 * window, trim n_fft/2, zero-pad to the note length.  S [total_frames x ldc] -> y [total_samples]. */
int goofer_irfft_ola(goofer_ctx *ctx, const float *S, int ldc, const int64_t *sample_off, const int64_t *frame_off,
                     int n_notes, int64_t total_frames, int64_t total_samples, float *y, void *stream);

/* The LF glottal-pulse model of gf.pulse_train_numba — its keyword arguments Ra, Rg, Rk (GOOFER.py:474, 508-519: opening
 * phase sin^2 up to Tp = Ra T, return phase exp(-Rg tau) cos(pi tau / 2) up to Tc = Tp + Rk (T - Tp)).  A plan starts with
 * 0.02 / 1.7 / 0.8, the values gf.synthesize passes (GOOFER.py:1074); this call rebuilds the plan's pulse tables for other
 * values (it synchronises the device) and they hold until the next goofer_plan.  Every pulse the handle makes afterwards —
 * goofer_pulse_train and the synthesis — uses them. */
int goofer_pulse_model(goofer_ctx *ctx, double Ra, double Rg, double Rk);

/* gf.pulse_train_numba (GOOFER.py:473-554) with the plan's pulse model (goofer_pulse_model): f0 [total_samples] -> pulse. */
int goofer_pulse_train(goofer_ctx *ctx, const float *f0, const int64_t *sample_off, int n_notes,
                       int64_t total_samples, float *pulse, void *stream);

/* gf.gaussian_filter1d(axis=bins) (GOOFER.py:241-261): fp64 taps [2*radius+1] (host memory, the
 * caller normalises them exactly as the reference does), numpy-'reflect' padding, fp64 accumulate.
 * in/out [rows x ld] fp32. */
int goofer_gauss_bins(goofer_ctx *ctx, const float *in, float *out, int64_t rows, int n_bins, int ld,
                      const double *taps, int radius, void *stream);

/* gf.warp_env_by_formants then gf.shift_formants (GOOFER.py:840-875, 618-627) on rows of env.
 * formants [rows x 4] fp64 (may be NULL when all f_shift == 1); f_shift NULL = no anchor warp;
 * ratio 1 = no uniform warp.  Each stage rounds to fp32 like the reference. */
int goofer_warp_bins(goofer_ctx *ctx, const float *in, float *out, int64_t rows, int n_bins, int ld,
                     const double *formants, const double *f_shift, double ratio, void *stream);

/* gf.decode_env_from_knots (GOOFER.py:149-168): knots fp16 [rows x K] (frames-major) -> env fp32
 * [rows x ld]; hz_knots fp32 [K] in host memory. */
int goofer_knot_decode(goofer_ctx *ctx, const uint16_t *knots_f16, int K, const float *hz_knots,
                       int64_t rows, float *env, int n_bins, int ld, void *stream);

/* ---- analysis half that does not need Praat (GOOFER.py:942-946, 97-147) -------------------------------- */

/* mag = abs(S) + 1e-8 per row (fp32).  S [rows x ldc] complex64 -> mag [rows x ld]. */
int goofer_mag_rows(goofer_ctx *ctx, const float *S, int ldc, int64_t rows, int n_bins, float *mag, int ld, void *stream);

/* gaussian_filter1d(axis=bins) with the reference's fp64 result kept: in fp32 [rows x ld] -> out fp64 [rows x ld64];
 * taps fp64 [2*radius+1] in HOST memory. */
int goofer_gauss_bins_f64(goofer_ctx *ctx, const float *in, int ld, double *out, int ld64, int64_t rows, int n_bins,
                          const double *taps, int radius, void *stream);

/* One candidate of compress_env_to_knots' search: max over the probe rows and all bins of
 * abs(exp(lerp(log knots)) - env) / (env + 1e-8), with knots = log(max(env, 1e-8)) sampled at knot_bin (all device
 * arrays; hz_knots fp32 [K] in HOST memory defines the lerp).  *max_rel_err is written on the host (synchronises). */
int goofer_knot_fit_error(goofer_ctx *ctx, const double *env, int ld64, const int64_t *probe_rows, int n_probe, int n_bins,
                          const int32_t *knot_bin, int K, const float *hz_knots, double *max_rel_err, void *stream);

/* knot_vals_log = log(max(env, 1e-8))[knot_bin] as fp16, [rows x K] (frames-major). */
int goofer_knot_gather(goofer_ctx *ctx, const double *env, int ld64, int64_t rows, const int32_t *knot_bin, int K,
                       uint16_t *knots_f16, void *stream);

/* ---- the hot path --------------------------------------------------------------------------- */

/* gf.synthesize for a ragged batch (GOOFER.py:971-1220) + the V/B/U mix (SillySampler.py:1142-1151). */
int goofer_synth_batch(goofer_ctx *ctx, const goofer_batch *batch, void *stream);

/* SillySampler.resample up to (not including) the synthesize call: knot decode, br / es / fw on the source
 * rows, slicing + loop modes + velocity stretch as a 4-tap frame gather, formant-strength gain, per-sample
 * voicing mask and pitch curve (SillySampler.py:449-855).  Outputs feed goofer_synth_batch directly. */
int goofer_assemble_batch(goofer_ctx *ctx, const goofer_assembly *assembly, void *stream);

/* SillySampler.resample for one batch as ONE call (SillySampler.py:449-1151 without the post chain): exactly
 * goofer_assemble_batch followed by goofer_synth_batch — same kernels, same results — for a batch whose
 * assembly->f0_out is batch->f0.  With both descriptors in hand everything they point to must already be enqueued on
 * `stream` (or complete) when the call is made; that lets the synthesis' pulse chain (f0 scaling, the sequential
 * phase walk, pulse placement) start on the handle's side stream as soon as the assembled f0 exists, beside the
 * envelope assembly, instead of when the synthesis' first kernel is reached in stream order. */
int goofer_render_batch(goofer_ctx *ctx, const goofer_assembly *assembly, const goofer_batch *batch, void *stream);

/* gf.stretch_feature (GOOFER.py:597-616): np.interp(linspace(0,1,rows_out), linspace(0,1,rows_in), column) along
 * axis 0 of a [rows x n_cols] fp32 matrix with row strides ld_in / ld_out (n_cols = 1, ld = 1: a 1-D array). */
/* Host-side helper of the note planner (no device involved): Gaussian FIR along the rows of an fp64 [rows x T] matrix, numpy
 * 'reflect' padding, taps applied in ascending order — the sigma-4 smoothing of the formant tracks, SillySampler.py:264-283. */
int goofer_host_gauss_rows(const double *x, int64_t rows, int T, const double *taps, int radius, double *out);

/* ---- host-side note planner (pure CPU code: no device, no handle) ----------------------------------------------------
 * What SillySampler.resample decides about a note before it touches an array (SillySampler.py:449-500 cut points, :625-696
 * loop modes, :698-788 sample counts + velocity stretch, :714-763 / 264-283 / 791-806 formant tracks, :883-955 fry ranges),
 * for a whole batch in one call; the arithmetic of goofer_amd/sampler.py's numpy planner, bit for bit. */
typedef struct {
    double offset, length, consonant, cutoff;   /* seconds: the request's offset / length / consonant / cutoff / 1000     */
    double vel_factor;                          /* 2 ** (1 - velocity / 100)                                :765           */
    double fry, fry_glide;                      /* 'vf' (clipped to +-100), 'vl'                            :883-888       */
    int64_t ylen;                               /* samples of the source                                                   */
    int32_t sr, n_src_frames;                   /* n_src_frames = rows of the source envelope = 1 + ylen / hop             */
    int32_t loop_mode, reverse;                 /* 0 concat (L0), 1 mirror mean (L1), 2 stretch (L2); 'R1'                 */
    const double *tracks[4];                    /* the source's F1..F4 tracks (Hz, fp64, HOST memory)                      */
    int32_t track_len[4];                       /* their lengths; -1: the source has no such track                         */
    int32_t fst_skip[4];                        /* != 0: this formant's 'fst' strength is off for the note (|s| < 1e-6), so */
                                                /* its fst_tracks column is never read: left 0, the sigma-4 blur not run    */
} goofer_plan_request;

typedef struct {
    int64_t start_sample, consonant_sample, end_sample;     /* cut points                                   :453-487       */
    int64_t tap_off;                            /* first row of the note in the arrays of goofer_host_plans_view           */
    double vel_factor;                          /* the stretch factor when vel_active, else 1                              */
    int32_t status;                             /* 0 planned; 1: a case the reference answers with an exception (empty     */
                                                /* tail to loop, negative length): nothing else of the record is valid     */
    int32_t start_frame, consonant_frame, end_frame;
    int32_t n_rows;                             /* envelope frames the reference would assemble                            */
    int32_t n_out_rows;                         /* rows planned: n_rows, or 1 + n_out / hop when trim_rows cut the rest     */
    int32_t row_lo, row_hi;                     /* source rows [row_lo, row_hi) the planned rows read                      */
    int32_t env_f64;                            /* the reference holds this note's envelope in float64 (lerps / fades)     */
    int32_t n_out, n_pre, s_pre, s_tail, tail_len, want_samples, n_before_vel, pre_new, vel_active;   /* goofer_note_plan */
    int32_t fry_dir, fry_const_lo, fry_const_hi, fry_glide_lo, fry_glide_hi, fry_a, fry_b, fry_fade;  /* goofer_note_plan */
    int32_t reserved;
} goofer_plan_geometry;

typedef struct goofer_host_plans goofer_host_plans;

/* Plan n_notes notes.  gauss_taps[2 * gauss_radius + 1]: the sigma-4 taps of sanitize_smooth_formant (:281), made by the
 * caller the way numpy makes them.  trim_rows != 0: plan only the envelope rows gf.synthesize can reach (GOOFER.py:1115-1119).
 * n_threads <= 0: up to eight host threads.  The result is freed with goofer_host_plans_free. */
int goofer_host_plan_notes(const goofer_plan_request *req, int n_notes, int hop, int trim_rows, const double *gauss_taps,
                           int gauss_radius, int n_threads, goofer_host_plans **out);
/* The same plans written into memory of the caller — pinned staging buffers one H2D copy ships, re-used from batch to batch.
 * geometry[n_notes]; the four row arrays hold row_capacity rows x 4 (layout as goofer_host_plans_view); *rows_out = rows of the
 * batch.  Returns 0 with the arrays filled, 1 when row_capacity is too small (*rows_out and the geometry's status / n_out_rows /
 * tap_off are written: grow, call again), or a negative error code as goofer_host_plan_notes. */
int goofer_host_plan_into(const goofer_plan_request *req, int n_notes, int hop, int trim_rows, const double *gauss_taps,
                          int gauss_radius, int n_threads, goofer_plan_geometry *geometry, int64_t row_capacity, int32_t *tap_idx,
                          double *tap_w, double *formants, float *fst_tracks, int64_t *rows_out);
/* The planned batch: geometry[n_notes]; rows = sum of n_out_rows over the notes with status 0; tap_idx / tap_w [rows x 4]
 * (goofer_assembly.tap_idx / tap_w), formants [rows x 4] fp64 (goofer_batch.formants), fst_tracks [rows x 4] fp32
 * (goofer_assembly.fst_tracks).  HOST memory owned by the handle.  Any out pointer may be NULL. */
int goofer_host_plans_view(const goofer_host_plans *plans, const goofer_plan_geometry **geometry, int64_t *rows,
                           const int32_t **tap_idx, const double **tap_w, const double **formants, const float **fst_tracks);
void goofer_host_plans_free(goofer_host_plans *plans);

/* UTAU pitch-bend strings of a batch (SillySampler.py:56-84): text = the strings back to back, text_off[n + 1] their bounds.
 * Writes out_off[n + 1] and, when out != NULL, the decoded cents (at most `capacity` values); returns the number of values,
 * or -(i + 1) when string i is not well formed (the caller's character loop then raises what the reference raises). */
int64_t goofer_host_decode_bends(const char *text, const int64_t *text_off, int n, float *out, int64_t capacity, int64_t *out_off);

/* float() of n plain decimal strings back to back (the numeric resampler arguments of a batch, SillySampler.py:289-298):
 * out[i] / ok[i] = 1 for [+-]digits[.digits][e[+-]digits]; ok[i] = 0 for anything else (the caller asks Python's float()).
 * strip_bang: skip leading '!' (the tempo argument).  Returns how many strings were not taken, or -1 on a null argument. */
int goofer_host_parse_floats(const char *text, const int64_t *text_off, int n, int strip_bang, double *out, unsigned char *ok);

/* Host arrays back to back into one block (pinned staging memory) on `threads` threads: piece i, nbytes[i] bytes at src[i], lands
 * behind the pieces in front of it.  The upload path of fresh voicebank samples (goofer_amd/render.py SourceArena: the reference
 * re-reads a sample's features from disk per note, GOOFER.py:1227-1262; here they go to HBM once).  Pure CPU code.  Returns the
 * bytes written, or a negative error (GOOFER_EINVAL: more than `capacity` bytes, null pieces). */
int64_t goofer_host_pack(const void *const *src, const int64_t *nbytes, int64_t count, void *dst, int64_t capacity, int threads);

/* Synchronise the device and report errors the asynchronous batch calls detect on the device (today: a note with more
 * pulse onsets than its n / 2 + 16 onset slots, GOOFER.py:493 with f0 above sr / 2).  0, or GOOFER_EINVAL + goofer_last_error. */
int goofer_check(goofer_ctx *ctx);

/* Cumulative per-handle counters kept on the device (the call synchronises): "pulse_scanned_notes" = notes whose pulse onsets
 * (GOOFER.py:487-493) were taken from the parallel phase scan, "pulse_fallback_notes" = those of them that had to be walked
 * sequentially afterwards because a sample's phase lay within the scan's rounding band of an integer. */
int goofer_counter(goofer_ctx *ctx, const char *name, int64_t *value);

/* gf.smooth_mask_ds (GOOFER.py:556-569) on its own, for a ragged batch of voicing masks (sample_off[n_notes + 1]): decimate by
 * 4, Gaussian max(1, sigma / 4) in fp64, np.interp back on float32 linspace grids -> out[total_samples] fp32.
 * fast_interp != 0 selects the interpolant form the stem walkers use (must give the same bits). */
int goofer_smooth_mask_ds(goofer_ctx *ctx, const float *mask, const int64_t *sample_off, int n_notes, int64_t total_samples,
                          float sigma, int fast_interp, float *out, void *stream);

int goofer_stretch_rows(goofer_ctx *ctx, const float *in, int64_t ld_in, int64_t rows_in, float *out, int64_t ld_out,
                        int64_t rows_out, int n_cols, void *stream);

/* gf.gaussian_filter1d along the last axis of ragged fp64 rows (GOOFER.py:241-261: numpy 'reflect' padding, fp64
 * accumulate in tap order): row r is in[row_off[r] .. row_off[r+1]) (device CSR); taps are HOST memory, 2*radius+1. */
int goofer_gauss_rows_f64(goofer_ctx *ctx, const double *in, const int64_t *row_off, int n_rows, int64_t total, const double *taps,
                          int radius, double *out, void *stream);

/* apply_vocal_roughness (GOOFER.py:901-940; gf.synthesize's roughness_on layer) for a ragged batch: out = y + alpha_slewed *
 * highpass(y * (1 + sum_k h_k cos(2 pi cumsum(f_mod_k) / sr)) - y), f_mod_k = max(f0 / k * (1 + noise_amp * noise_k), 0) * mask.
 * noise_s [n_k x total_samples] fp64 = the smoothed noises (make_smooth_noise), alpha_slewed [total_samples] fp32 = the slewed
 * alpha * mask — both Gaussian filters of host draws / the mask (goofer_gauss_rows_f64); k_list / h_list are HOST arrays. */
int goofer_vocal_roughness(goofer_ctx *ctx, const float *y, const float *f0, const float *mask, const double *noise_s, int n_k,
                           const double *k_list, const double *h_list, double noise_amp, double hp_fc, const float *alpha_slewed,
                           const int64_t *sample_off, int n_notes, int64_t total_samples, float *out, void *stream);

/* dynamic_butter_filter (SillySampler.py:95-174) for a list of jobs (device array): src -> dst, fp32. */
int goofer_onepole_cascade(goofer_ctx *ctx, const float *src, float *dst, const float *f0, const goofer_onepole_job *jobs,
                           int n_jobs, void *stream);

/* The sample-domain post chain after the synth calls (SillySampler.py:1037-1182): su / sj layers, fry blend,
 * sd dryness, st tension, V/B/U mix, sa whisper blend, pd gain.  Notes without any of these keep their mix. */
int goofer_post_batch(goofer_ctx *ctx, const goofer_post *post, void *stream);

/* Finished audio as the reference's wav holds it (soundfile's default PCM_16, SillySampler.py:1184-1185): out[i] =
 * int16(round_half_even(clip(x[i], -1, 1 - 2^-15) * 32768)), the arithmetic of goofer_amd.render.write_wav.  Two bytes per
 * sample across PCIe instead of four. */
int goofer_pcm16(goofer_ctx *ctx, const float *x, int64_t n, int16_t *out, void *stream);

/* ---- measurement / test hooks --------------------------------------------------------------- */

/* HIP-event timing of every stage of goofer_synth_batch on the caller's stream: begin() arms up to
 * max_steps batches, end() synchronises and writes the summed milliseconds of each stage
 * (goofer_profile_stage_name(i), i < 18); returns the number of batches recorded.  Stages 15..17 are the assembly's three
 * large kernels ("env_edit", "env_rows", "sample_assemble"), each bracketed on the stream it runs on. */
int goofer_profile_begin(goofer_ctx *ctx, int max_steps);
int goofer_profile_end(goofer_ctx *ctx, double *ms_per_stage, int n_stages);
const char *goofer_profile_stage_name(int stage);
const char *goofer_profile_stage_name_ex(const goofer_ctx *ctx, int stage);   /* names of the path the last profiled batch took */

/* Options (each is an A/B reference a parity test needs; every setting but td_blur and value_f64 produces the same stems):
 *   "fused_ola" 1 (default): irFFT of the three stems + overlap-add + gains in one kernel; 0: separate kernels
 *   "overlap"   1 (default): pulse chain on the handle's side stream beside the aperiodic branch; 0: one stream
 *   "stems"     1 (default): stem-split walker kernels where the geometry allows (hop == n_fft / 4); 0: one kernel per
 *               reference step up to the spectra, then the fused overlap-add
 *   "skip_zero" 1 (default): a noise-stem transform whose stem gain is exactly zero over every sample it reaches is skipped
 *               (the walkers decide per hop; the n_fft 2048 pipeline per frame up front); 0: every frame runs every transform
 *   "td_blur"   1 (default): the stem walkers fold the voiced frames' 5-tap bin blur into the synthesis window (agrees with
 *               0, the blur over the bins, to fp32 rounding — the only option that is not bit-identical)
 *   "pulse_scan" 1 (default): pulse onsets from a parallel fp64 phase scan wherever its rounding band provably cannot move
 *               floor(phase), the sequential walk only for the remaining notes; 0: the sequential walk for every note;
 *               2: the scan kernel walks every note (tests the hand-over)
 *   "value_f64" 0 (default): k_env_edit's VALUE arithmetic (fw interpolation, es blur, knot exp) in fp32 — everything that
 *               decides an index, a threshold or the pitch curve stays fp64; 1: round 4's fp64
 *               arithmetic there (agrees to 2e-8 sample-RMS on the 1024-note batch; not bit-identical)
 *   "sa_fast"   1 (default): k_sample_assemble's branch-free path with all of a thread's loads in flight together; 0: per sample
 *   "prof_only" s >= 0: goofer_profile_begin .. end record the events of stage s only; -1 (default): every stage     */
int goofer_set_option(goofer_ctx *ctx, const char *name, int value);

/* Copy a plan table (0 window, 1 freqs, 2 boost, 3 bright_harm, 4 bright_breath, 5 pulse peak) or an
 * intermediate of the last synth batch to HOST memory; return element count / byte size. Tests only. */
int goofer_debug_table(goofer_ctx *ctx, int which, float *host_out, int capacity);
/* sizeof of the ABI structs (0 note_params, 1 batch, 2 note_plan, 3 assembly, 4 onepole_job, 5 post_note, 6 post,
 * 7 plan_request, 8 plan_geometry) so bindings can verify layout */
int goofer_sizeof(int which);
int64_t goofer_debug_fetch(goofer_ctx *ctx, int which, void *host_out, int64_t capacity_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GOOFER_HIP_H */
