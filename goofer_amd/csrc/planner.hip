// Host-side note planner of goofer_assemble_batch (no device code, no HIP call): everything SillySampler.resample decides
// about a note before it touches an array, for a whole batch in one call.
//
//   cut points and clipped slices                                   SillySampler.py:453-500
//   loop-mode frame taps: L0 cross-faded repeats, L1 mirror mean, L2 stretch           :625-696
//   sample counts, velocity prefix stretch (a second 2-tap lerp on top of the first)   :698-788
//   the F1..F4 tracks handed to gf.synthesize and the repaired + sigma-4 smoothed tracks of the 'fst' gain   :714-763, 264-283, 791-806
//   vocal-fry sample ranges                                         :883-955
//
// goofer_amd/sampler.py holds the same decisions as numpy code (pinned by the reference's 53 index-plan fixtures); this file
// is that arithmetic note by note in C++ — fp64 / fp32 operations in the order numpy performs them, so the two agree to the
// bit (tests/test_planner_native.py) — because a render job of notes that all differ spent 0.15 ms per note in numpy
// dispatch, forty times the device time of the note.  Notes are independent: the batch is split over a few host threads.
//
// What numpy does that matters here:
//   linspace(a, b, n)   = arange(n) * ((b - a) / (n - 1)) + a, last element set to b; n == 1: [a]
//   np.interp           = slope * (x - xp[j]) + fp[j] in fp64, fp[j] itself on an exact hit, fp[-1] at / beyond the right end
//   Python round()      = round half to even (nearbyint in the default rounding mode); int() truncates; // floors
//   float32 <op> Python float  = float32 arithmetic (NEP 50 weak scalars)
#include <algorithm>
#include <cfenv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/goofer_hip.h"

namespace {

struct tap4 {
    int32_t i[4];
    double w[4];
};

inline int64_t floor_div(int64_t a, int64_t b) { return a / b - ((a % b != 0) && ((a < 0) != (b < 0))); }
inline int64_t py_round(double x) { return (int64_t)std::nearbyint(x); }

// x[a:b] on a sequence of length n (Python slice semantics), stop >= start
inline void clip_slice(int64_t a, int64_t b, int64_t n, int64_t &lo, int64_t &hi)
{
    auto fix = [n](int64_t v) {
        if (v < 0) {
            v += n;
            return v < 0 ? (int64_t)0 : v;
        }
        return v > n ? n : v;
    };
    lo = fix(a);
    hi = std::max(lo, fix(b));
}

inline void linspace(double a, double b, int64_t n, std::vector<double> &out)
{
    out.resize((size_t)std::max<int64_t>(n, 0));
    if (n <= 0) return;
    if (n == 1) {
        out[0] = a;
        return;
    }
    const double delta = b - a, step = delta / (double)(n - 1);
    if (step == 0.0) {
        for (int64_t i = 0; i < n; ++i) out[i] = ((double)i / (double)(n - 1)) * delta + a;
    } else {
        for (int64_t i = 0; i < n; ++i) out[i] = (double)i * step + a;
    }
    out[n - 1] = b;
}

// np.interp(x_new, x_old, .) as (j, 1 - c, c): the planner's _interp_taps
struct lerp_tap {
    int32_t j;
    double w0, w1;
};
inline lerp_tap interp_tap(const std::vector<double> &x_old, double x)
{
    const int64_t n = (int64_t)x_old.size();
    if (n == 1) return {0, 1.0, 0.0};
    int64_t j = (int64_t)(std::upper_bound(x_old.begin(), x_old.end(), x) - x_old.begin()) - 1;
    j = std::min(std::max<int64_t>(j, 0), std::max<int64_t>(n - 2, 0));
    double c = (x - x_old[j]) / (x_old[j + 1] - x_old[j]);
    if (x == x_old[j]) c = 0.0;
    if (x >= x_old[n - 1]) {
        j = n - 2;
        c = 1.0;
    }
    return {(int32_t)j, 1.0 - c, c};
}

// the planner's _interp_rows for one fp64 row (np.interp's formula + interp1d's linear extrapolation, GOOFER.py:204-205)
inline double interp_row(const std::vector<double> &x_old, const double *y, double x)
{
    const int64_t n = (int64_t)x_old.size();
    if (n == 1) return y[0];
    if (x < x_old[0]) {
        const double sl = (y[1] - y[0]) / (x_old[1] - x_old[0] + 1e-10);
        return y[0] + sl * (x - x_old[0]);
    }
    if (x > x_old[n - 1]) {
        const double sr = (y[n - 1] - y[n - 2]) / (x_old[n - 1] - x_old[n - 2] + 1e-10);
        return y[n - 1] + sr * (x - x_old[n - 1]);
    }
    if (x >= x_old[n - 1]) return y[n - 1];
    int64_t j = (int64_t)(std::upper_bound(x_old.begin(), x_old.end(), x) - x_old.begin()) - 1;
    j = std::min(std::max<int64_t>(j, 0), n - 2);
    if (x == x_old[j]) return y[j];
    const double slope = (y[j + 1] - y[j]) / (x_old[j + 1] - x_old[j]);
    return slope * (x - x_old[j]) + y[j];
}

// _prefix_positions: where output frame i of the velocity-stretched note sits on the frame axis before the stretch
inline void prefix_positions(int64_t n, int64_t pre_len, double factor, std::vector<double> &pos)
{
    const int64_t pre_new = std::max<int64_t>(1, py_round((double)pre_len * factor));
    const int64_t m = pre_new + (n - pre_len);
    pos.resize((size_t)std::max<int64_t>(m, 0));
    for (int64_t i = 0; i < m; ++i) pos[i] = i < pre_new ? (double)i / factor : (double)(i - pre_new) + (double)pre_len;
}

// pad_trim_to_len (GOOFER.py:64-70) on a non-empty vector
template <typename T> inline void fit_len(std::vector<T> &v, int64_t T_)
{
    if ((int64_t)v.size() >= T_) v.resize((size_t)std::max<int64_t>(T_, 0));
    else v.resize((size_t)T_, v.back());
}

struct scratch {
    std::vector<tap4> stage1, taps;
    std::vector<double> xo, xn, pos, f, tmp, pad;
    std::vector<float> tr, lp, canon, work;
    std::vector<int32_t> rows;
};

// acc[t] = sum_j taps[j] * pp[t + j], every frame adding its products in ascending tap order (product rounded, then added: the
// arithmetic of the numpy planner's loop, so the sums are the same bits whatever the vector width).  Taps outside, frames
// inside: a plain vector loop over t; built for AVX2 as well (chosen at load time), four frames per instruction.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2", "default")))
#endif
void fir_rows(const double *pp, const double *taps, int n_taps, int64_t T, double *acc)
{
    // sixteen frames at a time, their sums in registers across the taps (round 6: one pass over the taps per block instead of a
    // pass over the whole track per tap, which loaded and stored every partial sum 33 times)
    constexpr int W = 16;
    int64_t t0 = 0;
    for (; t0 + W <= T; t0 += W) {
        double a[W];
        const double *p0 = pp + t0;
        for (int u = 0; u < W; ++u) a[u] = taps[0] * p0[u];
        for (int j = 1; j < n_taps; ++j) {
            const double kj = taps[j];
            const double *pj = p0 + j;
            for (int u = 0; u < W; ++u) {
                const double prod = kj * pj[u];
                a[u] = a[u] + prod;
            }
        }
        for (int u = 0; u < W; ++u) acc[t0 + u] = a[u];
    }
    for (int64_t t = t0; t < T; ++t) {
        double a = taps[0] * pp[t];
        for (int j = 1; j < n_taps; ++j) {
            const double prod = taps[j] * pp[t + j];
            a = a + prod;
        }
        acc[t] = a;
    }
}

// sanitize_smooth_formant's repair (SillySampler.py:264-279) on fp32 values x[0..T): returns false when every value is bad
bool repair_track(float *x, int64_t T, float min_hz, float max_hz, scratch &s)
{
    int64_t n_bad = 0;
    for (int64_t i = 0; i < T; ++i) n_bad += (!std::isfinite(x[i]) || x[i] < min_hz || x[i] > max_hz) ? 1 : 0;
    if (n_bad == 0) return true;
    if (n_bad == T) return false;
    // good positions (as float32, like the reference's astype) and values
    std::vector<double> &gx = s.xo, &gy = s.tmp;
    gx.clear();
    gy.clear();
    std::vector<char> bad((size_t)T);
    for (int64_t i = 0; i < T; ++i) {
        bad[i] = (!std::isfinite(x[i]) || x[i] < min_hz || x[i] > max_hz) ? 1 : 0;
        if (!bad[i]) {
            gx.push_back((double)(float)i);
            gy.push_back((double)x[i]);
        }
    }
    const int64_t ng = (int64_t)gx.size();
    const float x0 = (float)gx[0], xl = (float)gx[ng - 1], y0 = (float)gy[0], yl = (float)gy[ng - 1];
    float sl = 0.f, sr = 0.f;
    if (ng > 1) {      // float32 slopes of the two ends: (y1 - y0) / (x1 - x0 + 1e-10) with every operand float32
        sl = ((float)gy[1] - y0) / (((float)gx[1] - x0) + 1e-10f);
        sr = (yl - (float)gy[ng - 2]) / ((xl - (float)gx[ng - 2]) + 1e-10f);
    }
    for (int64_t i = 0; i < T; ++i) {
        if (!bad[i]) continue;
        const float q = (float)i;
        if (ng == 1) {
            x[i] = y0;
        } else if (q < x0) {
            x[i] = y0 + sl * (q - x0);
        } else if (q > xl) {
            x[i] = yl + sr * (q - xl);
        } else {
            const double qd = (double)q;
            int64_t j = (int64_t)(std::upper_bound(gx.begin(), gx.end(), qd) - gx.begin()) - 1;
            const double slope = (gy[j + 1] - gy[j]) / (gx[j + 1] - gx[j]);
            x[i] = (float)(slope * (qd - gx[j]) + gy[j]);
        }
    }
    return true;
}

// Cut points and clipped slices of a note (:453-500): what both passes of the planner start from.
struct cuts {
    int64_t s0, s1, s2, fr0, fr1, fr2;
    int64_t f0a, f0b, f1a, f1b, n_pre_f, n_tail, want_f, want_s;
    int64_t s0a, s1a, n_pre, tail_len;
};
// false: a case the reference answers with an exception (or Python slicing of negative counts)
inline bool plan_cuts(const goofer_plan_request &r, int hop, cuts &c)
{
    const double sr = (double)r.sr;
    const int64_t ylen = r.ylen, T_src = r.n_src_frames;
    const double total = (double)ylen / sr;
    const double a0 = r.offset;
    const double b0 = r.cutoff < 0 ? (r.offset - r.cutoff) : (total - r.cutoff);
    double off = r.offset, cut = r.cutoff;
    if (r.reverse) {
        const double L = b0 - a0;
        off = total - b0;
        cut = total - (off + L);
    }
    c.s0 = (int64_t)(off * sr);
    c.s1 = c.s0 + (int64_t)(r.consonant * sr);
    c.s2 = (int64_t)((cut < 0 ? (off - cut) : (total - cut)) * sr);
    c.fr0 = floor_div(c.s0, hop);
    c.fr1 = floor_div(c.s1, hop);
    c.fr2 = floor_div(c.s2, hop);
    clip_slice(c.fr0, c.fr1, T_src, c.f0a, c.f0b);
    clip_slice(c.fr1, c.fr2, T_src, c.f1a, c.f1b);
    c.n_pre_f = c.f0b - c.f0a;
    c.n_tail = c.f1b - c.f1a;
    c.want_f = (int64_t)std::ceil(r.length * sr / (double)hop);
    c.want_s = (int64_t)(r.length * sr);
    int64_t s0b, s1b;
    clip_slice(c.s0, c.s1, ylen, c.s0a, s0b);
    clip_slice(c.s1, c.s2, ylen, c.s1a, s1b);
    c.n_pre = s0b - c.s0a;
    c.tail_len = s1b - c.s1a;
    return !(c.want_f < 0 || c.want_s < 0 || (c.n_tail < c.want_f && c.n_tail == 0) || (c.tail_len < c.want_s && c.tail_len == 0));
}

// Where stage 1 puts its rows: the note's tap list — or a counter, for the pass that only sizes the batch.
struct tap_list {
    static constexpr bool counting = false;
    std::vector<tap4> &st;
    void row(int64_t q) { st.push_back(tap4{{(int32_t)q, (int32_t)q, 0, 0}, {1.0, 0.0, 0.0, 0.0}}); }
    void tap(const tap4 &t) { st.push_back(t); }
    void many(int64_t) {}
};
struct tap_count {
    static constexpr bool counting = true;
    int64_t n = 0;
    void row(int64_t) { ++n; }
    void tap(const tap4 &) { ++n; }
    void many(int64_t k) { n += std::max<int64_t>(k, 0); }
};

// -- loop-mode frame taps (:631-696): one source for the rows and for their count
template <class Sink> void loop_taps(const goofer_plan_request &r, const cuts &c, Sink &st, scratch &s, bool &f64)
{
    const int64_t f0a = c.f0a, f0b = c.f0b, f1a = c.f1a, n_tail = c.n_tail, want_f = c.want_f;
    f64 = false;
    if (Sink::counting) st.many(f0b - f0a);
    else for (int64_t q = f0a; q < f0b; ++q) st.row(q);
    if (n_tail >= want_f) {
        if (Sink::counting) st.many(want_f);
        else for (int64_t q = 0; q < want_f; ++q) st.row(f1a + q);
    } else if (r.loop_mode == 2) {
        const int64_t n_new = (int64_t)((double)n_tail * ((double)want_f / (double)n_tail));
        if (Sink::counting) {
            st.many(n_new);
        } else {
            linspace(0.0, 1.0, n_tail, s.xo);
            linspace(0.0, 1.0, n_new, s.xn);
            for (int64_t q = 0; q < n_new; ++q) {
                const lerp_tap t = interp_tap(s.xo, s.xn[q]);
                const int32_t j1 = n_tail == 1 ? t.j : t.j + 1;
                st.tap(tap4{{(int32_t)(f1a + t.j), (int32_t)(f1a + j1), 0, 0}, {t.w0, t.w1, 0.0, 0.0}});
            }
        }
        f64 = n_new > 0;                                       // Taps(.., True) of an empty stretch is not counted (len 0)
    } else if (r.loop_mode == 1) {
        const int64_t reps = want_f / n_tail, rem = want_f % n_tail;
        if (Sink::counting) {
            st.many(reps * n_tail + rem);
        } else {
            for (int64_t cc = 0; cc <= reps; ++cc) {
                const int64_t m = cc < reps ? n_tail : rem;
                for (int64_t q = 0; q < m; ++q) st.tap(tap4{{(int32_t)(f1a + q), (int32_t)(f1a + n_tail - 1 - q), 0, 0}, {0.5, 0.5, 0.0, 0.0}});
            }
        }
    } else {
        // L0: every repeat but the last is "the tail, its last k frames cross-faded into the first k of a fresh copy, then the
        // rest of that copy" — and the fresh copy is appended again as the next chunk (:657-672)
        const int64_t n = n_tail, reps = want_f / n, rem = want_f % n;
        auto rows = [&](int64_t a, int64_t b) {                // tail[a:b]
            if (Sink::counting) st.many(b - a);
            else for (int64_t q = a; q < b; ++q) st.row(f1a + q);
        };
        auto faded = [&](int64_t m, int64_t k) {               // tail[:n-k] + cross-fade(k) + first m frames of the tail from k on
            rows(0, n - k);
            if (k > 0) {
                if (Sink::counting) {
                    st.many(k);
                } else {
                    linspace(0.0, 1.0, k, s.xo);               // up
                    linspace(1.0, 0.0, k, s.xn);               // down
                    for (int64_t q = 0; q < k; ++q) st.tap(tap4{{(int32_t)(f1a + n - k + q), (int32_t)(f1a + q), 0, 0}, {s.xn[q], s.xo[q], 0.0, 0.0}});
                }
                f64 = true;
            }
            rows(k, m);
        };
        const int64_t k = std::min<int64_t>(8, n / 2);
        for (int64_t cc = 0; cc + 1 < reps; ++cc) {
            if (k == 0) rows(0, n);                            // a one-frame tail: the chunk is the tail itself
            else faded(n, k);
        }
        if (rem) {
            const int64_t kr = std::min<int64_t>(8, rem / 2);
            if (kr > 0) {
                faded(rem, kr);
            } else {
                rows(0, n);
                rows(0, rem);
            }
        } else {
            rows(0, n);
        }
    }
}

// Sample counts and the velocity prefix stretch's geometry (:698-788) from the note's T_target = n1 frames; returns the
// frames of the stretched note (T_full) and sets g.n_out_rows (what synthesize can reach when trim_rows).
inline int64_t plan_sizes(const goofer_plan_request &r, const cuts &c, int64_t n1, int hop, int trim_rows, goofer_plan_geometry &g)
{
    g.n_pre = (int32_t)c.n_pre; g.tail_len = (int32_t)c.tail_len; g.want_samples = (int32_t)c.want_s;
    g.s_pre = (int32_t)c.s0a; g.s_tail = (int32_t)c.s1a;
    g.n_before_vel = (int32_t)(c.n_pre + c.want_s);
    const double vel = r.vel_factor;
    const bool vel_active = std::fabs(vel - 1.0) > 1e-6 && c.n_pre_f > 1 && c.n_pre > 1;
    g.vel_active = vel_active ? 1 : 0;
    g.vel_factor = vel_active ? vel : 1.0;
    int64_t T_full = n1;
    if (vel_active) {
        const int64_t pre_new_f = std::max<int64_t>(1, py_round((double)c.n_pre_f * vel));   // (prefix_positions' frame count)
        T_full = std::max<int64_t>(pre_new_f + (n1 - c.n_pre_f), 0);
        const int64_t pre_new = std::max<int64_t>(1, py_round((double)c.n_pre * vel));
        g.pre_new = (int32_t)pre_new;
        g.n_out = (int32_t)(pre_new + c.want_s);
    } else {
        g.pre_new = (int32_t)c.n_pre;
        g.n_out = g.n_before_vel;
    }
    g.n_rows = (int32_t)T_full;
    int64_t T_env = T_full;
    if (trim_rows && g.n_out > 0 && T_env > 1 + g.n_out / hop) T_env = 1 + g.n_out / hop;   // gf.synthesize never reads further (GOOFER.py:1115-1119)
    g.n_out_rows = (int32_t)T_env;
    return T_full;
}

// Pass 1: a note's status and row count (g.status, g.n_out_rows), nothing else of g is final.
inline void plan_count(const goofer_plan_request &r, int hop, int trim_rows, goofer_plan_geometry &g, scratch &s)
{
    std::memset(&g, 0, sizeof(g));
    cuts c;
    if (!plan_cuts(r, hop, c)) {
        g.status = 1;
        return;
    }
    tap_count k;
    bool f64;
    loop_taps(r, c, k, s, f64);
    plan_sizes(r, c, k.n, hop, trim_rows, g);
}

// Where a note's rows go: the caller's arrays at the note's first row (four values per row each).
struct row_dest {
    int32_t *tap_idx;
    double *tap_w, *F;
    float *fst;
};

// Pass 2, one note: everything, its rows written at `d`.  status 1: a case the reference answers with an exception (or Python
// slicing of negative counts) — the caller re-plans the batch with the numpy planner, which raises what the reference raises.
// status 2: the two passes disagree about the note's row count (a defect here; the caller takes the numpy planner as well).
void plan_one(const goofer_plan_request &r, int hop, int trim_rows, const double *gtaps, int gradius, int64_t rows_expected,
              goofer_plan_geometry &g, const row_dest &d, scratch &s)
{
    std::memset(&g, 0, sizeof(g));
    const double sr = (double)r.sr;
    cuts c;
    const bool ok = plan_cuts(r, hop, c);
    const int64_t fr0 = c.fr0, fr1 = c.fr1, fr2 = c.fr2, want_f = c.want_f, n_pre_f = c.n_pre_f;
    g.start_sample = c.s0; g.consonant_sample = c.s1; g.end_sample = c.s2;
    g.start_frame = (int32_t)fr0; g.consonant_frame = (int32_t)fr1; g.end_frame = (int32_t)fr2;
    if (!ok) {
        g.status = 1;
        return;
    }
    std::vector<tap4> &st = s.stage1;
    st.clear();
    bool f64 = false;
    {
        tap_list sink{st};
        loop_taps(r, c, sink, s, f64);
    }
    const int64_t n1 = (int64_t)st.size();                     // T_target
    // -- samples and the velocity prefix stretch (:698-788)
    const int64_t T_full = plan_sizes(r, c, n1, hop, trim_rows, g);
    const int64_t T_env = g.n_out_rows;
    const bool vel_active = g.vel_active != 0;
    const double vel = r.vel_factor;
    if (T_env != rows_expected) {
        g.status = 2;
        return;
    }
    std::vector<tap4> &out = s.taps;
    out.clear();
    if (vel_active) {
        prefix_positions(n1, n_pre_f, vel, s.pos);
        if ((int64_t)s.pos.size() != T_full) {
            g.status = 2;
            return;
        }
        s.xo.resize((size_t)n1);
        for (int64_t q = 0; q < n1; ++q) s.xo[q] = (double)q;
        out.reserve(s.pos.size());
        for (double p : s.pos) {
            const lerp_tap t = interp_tap(s.xo, p);
            const tap4 &a = st[t.j], &b = st[t.j + 1];
            out.push_back(tap4{{a.i[0], a.i[1], b.i[0], b.i[1]}, {a.w[0] * t.w0, a.w[1] * t.w0, b.w[0] * t.w1, b.w[1] * t.w1}});
        }
        f64 = true;
    } else {
        out.reserve((size_t)n1);
        for (const tap4 &a : st) out.push_back(tap4{{a.i[0], a.i[1], a.i[0], a.i[1]}, {a.w[0], a.w[1], 0.0, 0.0}});
    }
    g.env_f64 = f64 ? 1 : 0;
    int32_t lo = INT32_MAX, hi = -1;
    for (int64_t t = 0; t < T_env; ++t) {
        for (int cc = 0; cc < 4; ++cc) {
            if (out[t].w[cc] != 0.0) {
                lo = std::min(lo, out[t].i[cc]);
                hi = std::max(hi, out[t].i[cc]);
            }
            d.tap_idx[(size_t)t * 4 + cc] = out[t].i[cc];
            d.tap_w[(size_t)t * 4 + cc] = out[t].w[cc];
        }
    }
    g.row_lo = hi < 0 ? 0 : lo;
    g.row_hi = hi < 0 ? 0 : hi + 1;

    // -- vocal fry ranges (:883-955)
    {
        const int64_t n = g.n_out;
        const double vf = r.fry;
        if (vf != 0.0) {
            const int64_t L = py_round((double)n * (std::fabs(vf) / 100.0));
            if (L > 0) {
                const int64_t gl = std::min(std::max<int64_t>(py_round((double)L * (r.fry_glide / 100.0)), 0), L), cl = L - gl;
                g.fry_dir = vf > 0 ? 1 : -1;
                if (vf > 0) {
                    g.fry_const_lo = 0; g.fry_const_hi = (int32_t)cl; g.fry_glide_lo = (int32_t)cl; g.fry_glide_hi = (int32_t)L;
                } else {
                    const int64_t stt = n - L;
                    g.fry_glide_lo = (int32_t)stt; g.fry_glide_hi = (int32_t)(stt + gl); g.fry_const_lo = (int32_t)(stt + gl); g.fry_const_hi = (int32_t)n;
                }
            }
            const int64_t mid = floor_div(n, 2);
            int64_t a, b;
            if (vf > 0) {
                a = 0;
                b = std::max<int64_t>(0, std::min(n, py_round((double)mid * (vf / 100.0))));
            } else {
                a = std::max<int64_t>(0, n - py_round((double)(n - mid) * (std::fabs(vf) / 100.0)));
                b = n;
            }
            if (b > a) {
                g.fry_a = (int32_t)a; g.fry_b = (int32_t)b;
                g.fry_fade = (int32_t)(0.01 * sr);
            }
        }
    }

    // -- formant tracks (:714-763, 771-806)
    std::fill(d.F, d.F + (size_t)T_env * 4, 0.0);
    std::fill(d.fst, d.fst + (size_t)T_env * 4, 0.f);
    const float max_hz = (float)(sr * 0.48);
    const float min_hz[4] = {120.0f, 300.0f, 1500.0f, 2000.0f};
    for (int c = 0; c < 4; ++c) {
        const double *src = r.tracks[c];
        const int64_t Tk = r.track_len[c];                     // < 0: the source has no such track
        auto at = [&](int64_t i) { return r.reverse ? src[Tk - 1 - i] : src[i]; };
        std::vector<double> &f = s.f;
        f.clear();
        if (Tk >= 0) {
            int64_t pa, pb, ta, tb;
            clip_slice(fr0, fr1, Tk, pa, pb);
            clip_slice(fr1, fr2, Tk, ta, tb);
            for (int64_t q = pa; q < pb; ++q) f.push_back(at(q));
            const int64_t L = tb - ta;
            std::vector<float> &tr = s.tr, &lp = s.lp;
            tr.resize((size_t)L);
            for (int64_t q = 0; q < L; ++q) tr[q] = (float)at(ta + q);
            lp.clear();
            if (L == 0) {
                lp.assign((size_t)want_f, 0.f);
            } else if (r.loop_mode == 2) {
                const double factor = (double)want_f / (double)L;
                if (factor == 1.0) {
                    lp = tr;
                } else {
                    const int64_t n_new = (int64_t)((double)L * factor);
                    linspace(0.0, 1.0, L, s.xo);
                    linspace(0.0, 1.0, n_new, s.xn);
                    s.tmp.resize((size_t)L);
                    for (int64_t q = 0; q < L; ++q) s.tmp[q] = (double)tr[q];
                    lp.resize((size_t)std::max<int64_t>(n_new, 0));
                    for (int64_t q = 0; q < n_new; ++q) lp[q] = (float)interp_row(s.xo, s.tmp.data(), s.xn[q]);
                }
            } else {
                const int64_t reps = want_f / L, rem = want_f % L;
                lp.reserve((size_t)want_f);
                for (int64_t cc = 0; cc <= reps; ++cc) {
                    const int64_t m = cc < reps ? L : rem;
                    for (int64_t q = 0; q < m; ++q) lp.push_back(r.loop_mode == 1 ? (tr[q] + tr[L - 1 - q]) * 0.5f : tr[q]);
                }
            }
            for (float v : lp) f.push_back((double)v);
        }
        std::vector<float> &canon = s.canon;
        canon.clear();
        if (f.empty() && Tk >= 0 && n1 > 0) {
            // the reference edge-pads this track to the envelope's n1 frames and np.pad refuses an empty array (:755-760): a
            // stretch-mode tail longer than wanted, resampled to int(L * (want / L)) = 0 frames behind an empty prefix
            g.status = 1;
            return;
        }
        if (!f.empty()) {
            if (n1 == 0) {                                     // pad_trim of a non-empty track to zero frames: empty, i.e. no track
                f.clear();
            } else {
                fit_len(f, n1);
                if (vel_active) {                              // (n1 >= 2 here)
                    s.tmp = f;
                    s.xo.resize((size_t)n1);
                    for (int64_t q = 0; q < n1; ++q) s.xo[q] = (double)q;
                    f.resize(s.pos.size());
                    for (size_t q = 0; q < s.pos.size(); ++q) f[q] = interp_row(s.xo, s.tmp.data(), s.pos[q]);
                    if (!f.empty()) fit_len(f, T_full);
                }
                canon.resize(f.size());
                for (size_t q = 0; q < f.size(); ++q) canon[q] = (float)f[q];
                if (!canon.empty()) fit_len(canon, n1);       // the canon tracks use the PRE-velocity frame count (:791-806)
            }
        }
        // repair on [0, T_full): in place on the canon track when it is long enough (the repaired values then reach
        // synthesize: the reference's aliasing), on a padded copy otherwise; an all-bad track becomes 300 Hz and leaves the canon alone
        std::vector<float> &work = s.work;
        const bool have = !canon.empty();
        float *wp;
        if (have && (int64_t)canon.size() >= T_full) {
            wp = canon.data();
        } else {
            work.assign(canon.begin(), canon.end());
            if (have) fit_len(work, T_full);
            else work.assign((size_t)T_full, 0.f);
            wp = work.data();
        }
        bool all_bad = false;
        if (T_full > 0) {
            all_bad = !repair_track(wp, T_full, min_hz[c], max_hz, s);   // (writes only when some value is good: an all-bad canon stays)
        }
        // sigma-4 blur of the repaired track (fp64 taps in ascending order, numpy 'reflect'), rounded to fp32 — unless the caller
        // says the note's strength for this formant is off (the assembly then never reads the column: it stays 0)
        if (T_full > 0 && !r.fst_skip[c]) {
            std::vector<double> &pad = s.pad;
            pad.resize((size_t)(T_full + 2 * gradius));
            const int64_t period = T_full > 1 ? 2 * (T_full - 1) : 1;
            // numpy 'reflect' as a periodic map; one reflection covers every index when the radius is below the track length
            // (two 64-bit divisions per element were half of this function's time)
            const bool near = gradius < T_full;
            auto edge = [&](int64_t q) {
                int64_t m;
                if (T_full <= 1) m = 0;
                else if (near) m = q < 0 ? -q : (q >= T_full ? period - q : q);
                else {
                    m = ((q % period) + period) % period;
                    m = m < T_full ? m : period - m;
                }
                pad[q + gradius] = all_bad ? 300.0 : (double)wp[m];
            };
            // (only the frames the blur below reads: the first T_env + 2 radius of the padded track)
            const int64_t q_end = std::min<int64_t>(T_full + gradius, T_env + gradius);
            for (int64_t q = -gradius; q < 0; ++q) edge(q);
            const int64_t mid = std::min<int64_t>(T_full, q_end);
            if (all_bad) std::fill(pad.begin() + gradius, pad.begin() + gradius + mid, 300.0);
            else for (int64_t q = 0; q < mid; ++q) pad[q + gradius] = (double)wp[q];
            for (int64_t q = T_full; q < q_end; ++q) edge(q);
            // taps outside, frames inside: every frame still adds its products in ascending tap order (the sums are the same
            // bits), and the inner loop is a plain vector loop over t instead of a reduction over j
            std::vector<double> &acc = s.tmp;
            acc.resize((size_t)T_env);
            fir_rows(pad.data(), gtaps, 2 * gradius + 1, T_env, acc.data());
            for (int64_t t = 0; t < T_env; ++t) d.fst[(size_t)t * 4 + c] = (float)acc[t];
        }
        if (have) {
            const int64_t Lc = (int64_t)canon.size();
            for (int64_t t = 0; t < T_env; ++t) d.F[(size_t)t * 4 + c] = (double)canon[t < Lc ? t : Lc - 1];
        }
    }
}

}  // namespace

struct goofer_host_plans {
    std::vector<goofer_plan_geometry> geo;
    std::vector<int32_t> tap_idx;
    std::vector<double> tap_w, F;
    std::vector<float> fst;
    int64_t rows = 0;
};

extern "C" {

// Validation + the threaded per-note planning shared by the two entry points.  No exception leaves a worker thread
// (std::terminate: the whole render server would go down) or the extern "C" functions: each thread records its failure,
// allocation failures come back as GOOFER_ENOMEM.
static int run_threads(int nt, const std::function<void(int)> &fn)
{
    std::vector<int> failed((size_t)nt, 0);
    auto guarded = [&](int t) {
        try {
            fn(t);
        } catch (...) {
            failed[(size_t)t] = 1;
        }
    };
    if (nt == 1) {
        guarded(0);
    } else {
        std::vector<std::thread> th;
        int started = 1;
        try {
            for (int t = 1; t < nt; ++t, ++started) th.emplace_back(guarded, t);
        } catch (...) {                                        // a thread could not be started: its share runs on this one below
        }
        guarded(0);
        for (int t = started; t < nt; ++t) guarded(t);
        for (auto &x : th) x.join();
    }
    for (int f : failed)
        if (f) return GOOFER_ENOMEM;
    return GOOFER_OK;
}

static int check_requests(const goofer_plan_request *req, int n_notes, int hop, const double *gauss_taps, int gauss_radius)
{
    if (!req || n_notes < 0 || hop <= 0 || !gauss_taps || gauss_radius < 0) return GOOFER_EINVAL;
    for (int i = 0; i < n_notes; ++i) {
        const goofer_plan_request &q = req[i];
        if (q.sr <= 0 || q.ylen < 0 || q.n_src_frames < 0 || q.ylen > INT32_MAX || q.loop_mode < 0 || q.loop_mode > 2 || !(q.vel_factor > 0.0))
            return GOOFER_EINVAL;
        // Times that are not finite, or whose sample counts leave int32 (a length above ~13 h at 44.1 kHz): the double -> int64
        // casts below would be undefined and the int32 geometry fields would wrap.  EINVAL sends the caller to the numpy
        // planner, which raises what the reference raises for such a request (ValueError / MemoryError per note).
        const double sr = (double)q.sr, lim = (double)INT32_MAX - 8.0;
        const double times[] = {q.offset, q.length, q.consonant, q.cutoff, q.vel_factor};
        for (double v : times)
            if (!std::isfinite(v)) return GOOFER_EINVAL;
        const double total = (double)q.ylen / sr;
        if (!(std::fabs(q.offset) * sr < lim) || !(std::fabs(q.cutoff) * sr < lim) || !(std::fabs(q.consonant) * sr < lim) ||
            !(std::fabs(q.length) * sr < lim) || !((std::fabs(q.offset) + std::fabs(q.consonant) + std::fabs(q.cutoff) + total) * sr < lim) ||
            !(((double)q.ylen + std::fabs(q.length) * sr) * std::max(1.0, q.vel_factor) < lim))
            return GOOFER_EINVAL;
        for (int c = 0; c < 4; ++c)
            if (q.track_len[c] > 0 && !q.tracks[c]) return GOOFER_EINVAL;
    }
    return GOOFER_OK;
}

static int pick_threads(int n_threads, int n_notes)
{
    const int nt = n_threads > 0 ? n_threads : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    return std::max(1, std::min(nt, (n_notes + 31) / 32));
}

// Pass 1: every note's status and row count, then its first row (tap_off); returns the rows of the batch.  A note is sized
// in ~0.1 us (cut points and the loop mode's counting arithmetic — the same code that emits the rows, with a counter for a
// sink), so the rows can go straight to where they belong in pass 2.  (Until round 6 every thread planned into buffers of
// its own and a second pass gathered them: 18.7 MB of freshly mapped memory and a copy of it per 1024 notes, which was
// 45 % of the planner's time.)
static int64_t size_batch(const goofer_plan_request *req, int n_notes, int hop, int trim_rows, goofer_plan_geometry *geo)
{
    std::fesetround(FE_TONEAREST);
    scratch s;
    int64_t rows = 0;
    for (int i = 0; i < n_notes; ++i) {
        plan_count(req[i], hop, trim_rows, geo[i], s);
        geo[i].tap_off = rows;
        if (geo[i].status == 0) rows += geo[i].n_out_rows;
    }
    return rows;
}

// Pass 2: the notes' plans, each thread a contiguous range of notes holding its share of the rows.
static int fill_batch(const goofer_plan_request *req, int n_notes, int hop, int trim_rows, const double *gauss_taps, int gauss_radius,
                      int nt, int64_t rows, goofer_plan_geometry *geo, int32_t *tap_idx, double *tap_w, double *F, float *fst)
{
    std::vector<int> first((size_t)nt + 1, n_notes);           // thread t: notes [first[t], first[t + 1])
    first[0] = 0;
    for (int t = 1, i = 0; t < nt; ++t) {
        const int64_t want = rows * t / nt;
        while (i < n_notes && geo[i].tap_off < want) ++i;
        first[(size_t)t] = i;
    }
    return run_threads(nt, [&](int t) {
        std::fesetround(FE_TONEAREST);
        scratch s;
        for (int i = first[(size_t)t]; i < first[(size_t)t + 1]; ++i) {
            if (geo[i].status != 0) continue;                  // (its cut points stay unwritten: the caller re-plans the batch)
            const int64_t r0 = geo[i].tap_off, T = geo[i].n_out_rows;
            goofer_plan_geometry g;
            const row_dest d{tap_idx + (size_t)r0 * 4, tap_w + (size_t)r0 * 4, F + (size_t)r0 * 4, fst + (size_t)r0 * 4};
            plan_one(req[i], hop, trim_rows, gauss_taps, gauss_radius, T, g, d, s);
            g.tap_off = r0;
            if (g.status != 0) g.n_out_rows = (int32_t)T;      // (status 2: the rows of the note are not written)
            geo[i] = g;
        }
    });
}

int goofer_host_plan_notes(const goofer_plan_request *req, int n_notes, int hop, int trim_rows, const double *gauss_taps,
                           int gauss_radius, int n_threads, goofer_host_plans **out)
{
    if (!out) return GOOFER_EINVAL;
    *out = nullptr;
    goofer_host_plans *h = nullptr;
    try {
        int rc = check_requests(req, n_notes, hop, gauss_taps, gauss_radius);
        if (rc) return rc;
        h = new (std::nothrow) goofer_host_plans;
        if (!h) return GOOFER_ENOMEM;
        h->geo.resize((size_t)n_notes);
        const int64_t rows = size_batch(req, n_notes, hop, trim_rows, h->geo.data());
        h->rows = rows;
        h->tap_idx.resize((size_t)rows * 4);
        h->tap_w.resize((size_t)rows * 4);
        h->F.resize((size_t)rows * 4);
        h->fst.resize((size_t)rows * 4);
        if ((rc = fill_batch(req, n_notes, hop, trim_rows, gauss_taps, gauss_radius, pick_threads(n_threads, n_notes), rows, h->geo.data(),
                             h->tap_idx.data(), h->tap_w.data(), h->F.data(), h->fst.data()))) {
            delete h;
            return rc;
        }
    } catch (...) {
        delete h;
        return GOOFER_ENOMEM;
    }
    *out = h;
    return GOOFER_OK;
}

/* The same plans written straight into memory of the caller (pinned staging buffers that one H2D copy then ships, re-used from
 * batch to batch: no allocation, no page faults, no second copy).  geometry[n_notes]; the four row arrays hold row_capacity
 * rows x 4.  *rows_out = rows of the batch (sum of n_out_rows over the notes with status 0).  Returns GOOFER_OK with the arrays
 * filled; 1 when row_capacity is too small (only *rows_out and the geometry's status / n_out_rows / tap_off are written: grow
 * and call again); an error code. */
int goofer_host_plan_into(const goofer_plan_request *req, int n_notes, int hop, int trim_rows, const double *gauss_taps,
                          int gauss_radius, int n_threads, goofer_plan_geometry *geometry, int64_t row_capacity, int32_t *tap_idx,
                          double *tap_w, double *formants, float *fst_tracks, int64_t *rows_out)
{
    if (!geometry || !rows_out || row_capacity < 0 || (row_capacity > 0 && (!tap_idx || !tap_w || !formants || !fst_tracks))) return GOOFER_EINVAL;
    try {
        int rc = check_requests(req, n_notes, hop, gauss_taps, gauss_radius);
        if (rc) return rc;
        const int64_t rows = size_batch(req, n_notes, hop, trim_rows, geometry);
        *rows_out = rows;
        if (rows > row_capacity) return 1;
        return fill_batch(req, n_notes, hop, trim_rows, gauss_taps, gauss_radius, pick_threads(n_threads, n_notes), rows, geometry, tap_idx,
                          tap_w, formants, fst_tracks);
    } catch (...) {
        return GOOFER_ENOMEM;
    }
}

int goofer_host_plans_view(const goofer_host_plans *h, const goofer_plan_geometry **geometry, int64_t *rows, const int32_t **tap_idx,
                           const double **tap_w, const double **formants, const float **fst_tracks)
{
    if (!h) return GOOFER_EINVAL;
    if (geometry) *geometry = h->geo.data();
    if (rows) *rows = h->rows;
    if (tap_idx) *tap_idx = h->tap_idx.data();
    if (tap_w) *tap_w = h->tap_w.data();
    if (formants) *formants = h->F.data();
    if (fst_tracks) *fst_tracks = h->fst.data();
    return GOOFER_OK;
}

void goofer_host_plans_free(goofer_host_plans *h) { delete h; }

// UTAU pitch-bend strings (SillySampler.py:56-84): base64 pairs = 12-bit two's complement cents, "#n#" = repeat the last
// value n more times.  text = the strings back to back, text_off[n + 1] their bounds.  out == NULL: only count.  Returns the
// total number of values (out_off[n + 1] = where each note's values start), or -(i + 1) when string i is not well formed (odd
// segment, bad character, a run with nothing in front, an empty string: the caller's character loop then raises — or
// answers — what the reference does).
int64_t goofer_host_decode_bends(const char *text, const int64_t *text_off, int n, float *out, int64_t capacity, int64_t *out_off)
{
    if (!text || !text_off || n < 0 || !out_off) return 0;
    static const auto lut = [] {
        std::vector<int> t(256, -1);
        for (int c = 0; c < 256; ++c) {
            if (c >= 'a' && c <= 'z') t[c] = c - 71;
            else if (c >= 'A' && c <= 'Z') t[c] = c - 65;
            else if (c >= '0' && c <= '9') t[c] = c + 4;
            else if (c == '+') t[c] = 62;
            else if (c == '/') t[c] = 63;
        }
        return t;
    }();
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        out_off[i] = total;
        const char *p = text + text_off[i], *e = text + text_off[i + 1];
        if (e > p && e[-1] == '\0') --e;                       // (strings joined with a NUL separator: it belongs to neither)
        bool have = false;
        float last = 0.f;
        int64_t count = 0;
        while (p < e) {
            // a segment of base64 pairs up to '#' or the end
            const char *q = p;
            while (q < e && *q != '#') ++q;
            if ((q - p) & 1) return -(int64_t)(i + 1);
            for (; p < q; p += 2) {
                const int a = lut[(unsigned char)p[0]], b = lut[(unsigned char)p[1]];
                if (a < 0 || b < 0) return -(int64_t)(i + 1);
                int v = (a << 6) | b;
                if (v & 0x800) v -= 4096;
                last = (float)v;
                have = true;
                if (out && total + count < capacity) out[total + count] = last;
                ++count;
            }
            if (p >= e) break;
            // "#digits#" (the closing '#' may be missing at the end of the string: str.split still yields the count)
            ++p;
            const char *d = p;
            int64_t run = 0;
            while (p < e && *p != '#') {
                if (*p < '0' || *p > '9' || run > (1 << 24)) return -(int64_t)(i + 1);
                run = run * 10 + (*p - '0');
                ++p;
            }
            if (p == d || !have) return -(int64_t)(i + 1);
            for (int64_t k = 0; k < run; ++k) {
                if (out && total + count < capacity) out[total + count] = last;
                ++count;
            }
            if (p < e) ++p;                                   // the closing '#'
        }
        if (count == 0) return -(int64_t)(i + 1);
        total += count;
    }
    out_off[n] = total;
    return total;
}

/* float(text) for n short decimal strings back to back (text_off[n + 1]): out[i] = the correctly rounded double of string i
 * (strtod), ok[i] = 1 — for plain decimal literals only: [+-]digits[.digits][e[+-]digits], at most 63 characters, no spaces,
 * underscores, hex, inf or nan.  Anything else gets ok[i] = 0 and the caller lets Python's float() answer (or raise).  With
 * strip_bang != 0 leading '!' characters are skipped first (the tempo argument, SillySampler.py:298).  Returns the number of
 * strings that were not taken. */
int goofer_host_parse_floats(const char *text, const int64_t *text_off, int n, int strip_bang, double *out, unsigned char *ok)
{
    if (!text || !text_off || n < 0 || !out || !ok) return -1;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const char *p = text + text_off[i], *e = text + text_off[i + 1];
        if (e > p && e[-1] == '\0') --e;                       // (strings joined with a NUL separator: it belongs to neither)
        if (strip_bang)
            while (p < e && *p == '!') ++p;
        char buf[64];
        const int64_t len = e - p;
        bool good = len > 0 && len < 64;
        if (good) {
            const char *q = p;
            if (*q == '+' || *q == '-') ++q;
            int digits = 0;
            while (q < e && *q >= '0' && *q <= '9') { ++q; ++digits; }
            if (q < e && *q == '.') {
                ++q;
                while (q < e && *q >= '0' && *q <= '9') { ++q; ++digits; }
            }
            if (digits == 0) good = false;
            if (good && q < e && (*q == 'e' || *q == 'E')) {
                ++q;
                if (q < e && (*q == '+' || *q == '-')) ++q;
                int ed = 0;
                while (q < e && *q >= '0' && *q <= '9') { ++q; ++ed; }
                if (ed == 0) good = false;
            }
            if (q != e) good = false;
        }
        if (good) {
            std::memcpy(buf, p, (size_t)len);
            buf[len] = 0;
            out[i] = strtod(buf, nullptr);                     // "C" locale in this process: '.' is the decimal point
            ok[i] = 1;
        } else {
            out[i] = 0.0;
            ok[i] = 0;
            ++bad;
        }
    }
    return bad;
}

/* Host arrays back to back into one (pinned) block on several threads: piece i (nbytes[i] bytes at src[i]) lands at the sum of the
 * sizes in front of it.  The voicing masks of a batch of fresh voicebank samples are 200 MB; one thread copies that into pinned
 * memory at a third of the rate the DMA engine then moves it over PCIe.  The byte range is cut evenly over the threads.
 * Returns the bytes written, or a negative error. */
int64_t goofer_host_pack(const void *const *src, const int64_t *nbytes, int64_t count, void *dst, int64_t capacity, int threads)
{
    if (count < 0 || (count > 0 && (!src || !nbytes || !dst))) return GOOFER_EINVAL;
    std::vector<int64_t> at;
    try {
        at.resize((size_t)count + 1);
    } catch (...) {
        return GOOFER_ENOMEM;
    }
    at[0] = 0;
    for (int64_t i = 0; i < count; ++i) {
        if (nbytes[i] < 0 || (nbytes[i] > 0 && !src[i])) return GOOFER_EINVAL;
        at[(size_t)i + 1] = at[(size_t)i] + nbytes[i];
    }
    const int64_t total = at[(size_t)count];
    if (total > capacity) return GOOFER_EINVAL;
    int nt = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
    if (total < (int64_t)(4 << 20)) nt = 1;                   // not worth a thread start
    const int64_t share = (total + nt - 1) / nt;
    const int rc = run_threads(nt, [&](int t) {
        const int64_t lo = (int64_t)t * share, hi = std::min(total, lo + share);
        if (lo >= hi) return;
        size_t i = (size_t)(std::upper_bound(at.begin(), at.end(), lo) - at.begin()) - 1;   // the piece byte `lo` is in
        for (int64_t b = lo; b < hi; ++i) {
            const int64_t end = std::min(hi, at[i + 1]);
            if (end > b) std::memcpy((char *)dst + b, (const char *)src[i] + (b - at[i]), (size_t)(end - b));
            b = end > b ? end : b;
        }
    });
    return rc ? rc : total;
}

}  // extern "C"
