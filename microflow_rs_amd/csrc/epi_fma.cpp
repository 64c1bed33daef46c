// epi_fma.cpp -- host search for the single-fma requantisation ("epilogue mode 3", k_common.hpp).
//
// The reference ends every conv-like operator in   y = sat_T(roundf(fl(A + fl(S * f32(acc)))))   (src/ops/conv_2d.rs:93-98,
// depthwise_conv_2d.rs:90-95; A = fl(f32(ozp) + c0[c]), S = c1[c]).  As a function of the integer accumulator that is a
// monotone staircase with at most 255 steps.  ANY other monotone map with the same step positions over the accumulators the
// operator can produce is bit-identical to it, and the cheapest one the VALU offers is
//
//     x = v_fma_f32(S', F, C')                 F = the accumulator's own bit pattern read as f32 = 1.5 * 2^23 + acc + d
//     y = v_cvt_pk_u8_f32(x)                   saturate to [0, 255]  (u8 domain: i8 results are y ^ 0x80)
//
// both executed with the f32 rounding mode TOWARD ZERO (the kernel sets MODE.FP_ROUND on entry, k_common.hpp epi_enter; both
// instructions follow it -- measured over all 2^32 inputs, scripts/ubench/cvt_pk_probe.hip): x = RZ(S' F + C') and y = trunc(x),
// together y = floor(S' F + C') of the EXACT real value, clamped.  Two instructions per byte instead of six.  Per channel the host
// looks for (S', C', d) -- S' within a few ulps of S, an integer pivot d folded into the accumulator's start value, C' an f32 --
// whose staircase has exactly the reference's steps:
//   1. the reference's steps T_k (first accumulator whose output is >= k) inside the reachable accumulator range [amin, amax] are
//      found by bisection on the exact two-rounding form (monotone: every f32 operation in it is);
//   2. floor(s * acc + e) reproduces them iff  e in [max_k (k - s T_k), min_k (k - s (T_k - 1)))  -- the width w(s) of that
//      interval is evaluated, exactly, for the f32 neighbours of S and the candidates are tried widest first;
//   3. e = C' + s * (1.5 * 2^23 + d) must hit the interval with C' an f32 of magnitude ~ s * 2^23 (ulp 2^-4 .. 2^-11, far coarser
//      than w): the pivot d supplies the fraction -- s * d mod ulp(C') is equidistributed, so one d in ulp / w works;
//   4. every candidate is then checked with the instructions emulated at the 2 x 255 accumulators that decide the steps: T_k must
//      give >= k, T_k - 1 must give < k.  With monotonicity of both maps that is equality on the whole range, not a sample.
// What the host proves here the device re-checks exhaustively, accumulator by accumulator, with the real instructions
// (k_generic.hip: verify_fma_form) before an operator is allowed to use the form; a channel without a solution keeps its operator
// (and every fused launch the operator is part of) on the two-rounding forms.  A channel has no solution when two near-ties of the
// reference were resolved in opposite directions by ITS roundings (fl(S acc) at the grid of S acc, fl(A + .) at the grid of the
// sum): then no line passes all 2 x 255 gaps.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "mf_internal.hpp"

namespace mf {

namespace {
constexpr int64_t M0 = 12582912; // 1.5 * 2^23: the f32 whose bit pattern is 0x4B400000
inline float next_up(float x, int n) { // n ulps away (n may be negative); x > 0 finite normal
    int32_t b;
    std::memcpy(&b, &x, 4);
    b += n;
    float r;
    std::memcpy(&r, &b, 4);
    return r;
}
} // namespace

// the reference's tail for one accumulator, in the u8 domain (off = 128 for i8, 0 for u8): clamp(roundf(A + S acc), lo, hi) + off
int ref_form_eval(float A, float S, int off, int lo, int hi, int64_t acc) {
    volatile float p = S * (float)acc; // |acc| <= 2^24: exact conversion
    volatile float x = A + p;
    float r = h_roundf(x);
    if (std::isnan(r)) r = 0.0f; // Rust: NaN as T == 0
    int y = r >= (float)hi ? hi : (r <= (float)lo ? lo : (int)r);
    return y + off;
}

// the device's form: trunc(RZ_f32(S F + C)) = floor of the exact value for positive values (every integer below 2^24 is an f32, so
// rounding toward zero to f32 never crosses one), 0 for negative ones and NaN, 255 from 255 on.  S F is exact in double (24 x 24
// bits) and so is the sum: both terms are multiples of ulp(S) * 1 or ulp(C) -- at least 2^-60 for any S that gives a staircase --
// and the sum is below 2^9 wherever its value matters (a sum outside [0, 256) saturates whatever its low bits are).
int fma_form_eval(const FmaForm &f, int64_t acc) {
    if (f.patch_delta != 0 && acc == f.patch_acc) acc += f.patch_delta; // the one replaced accumulator (see fma_form_search)
    const double F = (double)(M0 + acc + (int64_t)f.d); // an integer in [2^23, 2^24): what the bit pattern 0x4B400000 + acc + d reads as
    const double v = std::fma((double)f.S, F, (double)f.C);
    if (!(v >= 1.0)) return 0;
    if (v >= 255.0) return 255;
    return (int)std::floor(v);
}

bool fma_form_search(float A, float S, int off, int lo, int hi, int64_t amin, int64_t amax, FmaForm &out, FmaSearchStats *st, bool allow_patch) {
    if (st) *st = FmaSearchStats{};
    out = FmaForm{};
    if (!std::isfinite(A) || !std::isfinite(S) || !(S > 0.0f) || std::fpclassify(S) != FP_NORMAL) return false;
    if (amin > amax || amin <= -(1 << 22) || amax >= (1 << 22) - 1 || lo > hi) return false;
    auto Y = [&](int64_t a) { return ref_form_eval(A, S, off, lo, hi, a); };
    // --- 1. the steps of the reference inside [amin, amax] ---
    struct Con { int64_t a; int k; };
    std::vector<Con> lower, upper; // lower: y(a) >= k;  upper: y(a) < k
    const int y0 = Y(amin), yN = Y(amax);
    if (y0 < 0 || yN > 255 || y0 > yN) return false;
    if (y0 >= 1) lower.push_back({amin, y0});
    if (yN <= 254) upper.push_back({amax, yN + 1});
    for (int k = y0 + 1; k <= yN; ++k) {
        int64_t l = amin, h = amax; // Y(l) < k <= Y(h)
        while (h - l > 1) {
            const int64_t m = l + (h - l) / 2;
            if (Y(m) >= k) h = m;
            else l = m;
        }
        lower.push_back({h, k});
        upper.push_back({h - 1, k});
    }
    if (st) st->steps = yN - y0;
    // --- 2. the EXACT window of e: with v = s acc + e (real numbers; every product and sum below is exact in double) the form gives
    // y >= k  <=>  floor(v) >= k  <=>  v >= k, for k = 1 .. 255 -- one comparison per step, no grid or tie effects.
    double th[257];
    for (int k = 0; k <= 256; ++k) th[k] = (double)k;
    const double INF = std::numeric_limits<double>::infinity();
    auto window = [&](float s, double &elo, double &ehi) {
        elo = -INF, ehi = INF;
        for (const Con &c : lower) elo = std::max(elo, th[c.k] - (double)s * (double)c.a);
        for (const Con &c : upper) ehi = std::min(ehi, th[c.k] - (double)s * (double)c.a);
        if (elo == -INF) elo = ehi - 0.5; // a constant output: any e on the right side
        if (ehi == INF) ehi = elo + 0.5;
    };
    struct Cand { float s; double elo, ehi; };
    std::vector<Cand> cands;
    for (int J = 8; J <= 64 && cands.empty(); J *= 8) // S itself and its neighbours; further out only when those have no window
        for (int j = -J; j <= J; ++j) {
            const float s = next_up(S, j);
            if (!(s > 0.0f) || std::fpclassify(s) != FP_NORMAL) continue;
            double elo, ehi;
            window(s, elo, ehi);
            if (ehi > elo) cands.push_back({s, elo, ehi});
        }
    if (st) st->s_candidates = (int)cands.size();
    std::sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return (a.ehi - a.elo) > (b.ehi - b.elo); });
    if (st && !cands.empty()) st->best_width = cands[0].ehi - cands[0].elo;
    // --- 3 + 4. pivot and offset, exact check ---
    const int64_t dlo = -(1 << 22) - amin, dhi = (1 << 22) - 1 - amax; // 2^23 <= M0 + acc + d < 2^24 for every acc in [amin, amax]
    auto exact_ok = [&](const FmaForm &f) {
        for (const Con &c : lower)
            if (fma_form_eval(f, c.a) < c.k) return false;
        for (const Con &c : upper)
            if (fma_form_eval(f, c.a) >= c.k) return false;
        return true;
    };
    // One patched accumulator.  A channel has no line when two steps of the reference need offsets e that exclude each other (its
    // own roundings pushed two near-ties apart).  Dropping ONE of the 2 x 255 conditions -- the tightest lower or the tightest upper
    // one -- usually leaves a non-empty window; the line then differs from the reference at exactly one accumulator a* (the dropped
    // condition's), by one output step, and replacing a* by its neighbour a* +- 1 BEFORE the fma (a compare and a select in the few
    // lanes and tiles concerned, k_common.hpp epi_patch) restores equality: the neighbour lies on the right side of the line's step.
    auto patch_search = [&]() {
        for (int j = -16; j <= 16; ++j) {
            const float s = next_up(S, j);
            if (!(s > 0.0f) || std::fpclassify(s) != FP_NORMAL) continue;
            // the two largest lower bounds and the two smallest upper bounds, with the conditions they come from
            int l1 = -1, l2 = -1, h1 = -1, h2 = -1;
            auto lv = [&](int i) { return th[lower[(size_t)i].k] - (double)s * (double)lower[(size_t)i].a; };
            auto hv = [&](int i) { return th[upper[(size_t)i].k] - (double)s * (double)upper[(size_t)i].a; };
            for (int i = 0; i < (int)lower.size(); ++i) {
                if (l1 < 0 || lv(i) > lv(l1)) l2 = l1, l1 = i;
                else if (l2 < 0 || lv(i) > lv(l2)) l2 = i;
            }
            for (int i = 0; i < (int)upper.size(); ++i) {
                if (h1 < 0 || hv(i) < hv(h1)) h2 = h1, h1 = i;
                else if (h2 < 0 || hv(i) < hv(h2)) h2 = i;
            }
            if (l1 < 0 || h1 < 0) continue;
            for (int drop_lower = 1; drop_lower >= 0; --drop_lower) {
                double elo, ehi;
                Con dropped;
                if (drop_lower) {
                    if (l2 < 0) continue;
                    elo = lv(l2), ehi = hv(h1), dropped = lower[(size_t)l1];
                } else {
                    if (h2 < 0) continue;
                    elo = lv(l1), ehi = hv(h2), dropped = upper[(size_t)h1];
                }
                if (!(ehi > elo)) continue;
                const double w = ehi - elo, emid = 0.5 * (elo + ehi);
                const int64_t budget = std::min<int64_t>(2 * std::max(dhi, -dlo) + 1, 2000000);
                int tries = 0;
                for (int64_t i = 0; i < budget && tries < 16; ++i) {
                    const int64_t d = (i & 1) ? (i + 1) / 2 : -(i / 2);
                    if (d > dhi || d < dlo) continue;
                    const double t = emid - (double)s * (double)(M0 + d);
                    const float C = (float)t;
                    if (!(std::fabs((double)C - t) < 0.499 * w)) continue;
                    ++tries;
                    FmaForm f{s, C, (int32_t)d};
                    // the neighbour that carries the reference's value at the dropped accumulator
                    const int want = Y(dropped.a);
                    for (int delta = 1; delta >= -1; delta -= 2) {
                        const int64_t nb = dropped.a + delta;
                        if (nb < amin - 1 || nb > amax + 1 || nb + d < -(1 << 22) || nb + d >= (1 << 22)) continue;
                        f.patch_acc = dropped.a, f.patch_delta = 0;
                        if (fma_form_eval(f, nb) != want) continue;
                        f.patch_delta = delta;
                        if (exact_ok(f)) {
                            out = f;
                            if (st) st->s_rank = -2;
                            return true;
                        }
                    }
                }
            }
        }
        return false;
    };
    for (size_t ci = 0; ci < cands.size(); ++ci) {
        const Cand &c = cands[ci];
        const double w = c.ehi - c.elo, emid = 0.5 * (c.elo + c.ehi);
        const double s = (double)c.s;
        // pivots in the order 0, 1, -1, 2, -2, ...: the first hit has the smallest |d|
        const int64_t budget = std::min<int64_t>(2 * std::max(dhi, -dlo) + 1, 2000000);
        int exact_tries = 0;
        for (int64_t i = 0; i < budget; ++i) {
            const int64_t d = (i & 1) ? (i + 1) / 2 : -(i / 2);
            if (d > dhi || d < dlo) continue;
            const double t = emid - s * (double)(M0 + d); // x = s F + C, e = C + s (M0 + d); the product is 24 x 25 bits: exact in double
            const float C = (float)t;
            if (!(std::fabs((double)C - t) < 0.499 * w)) continue;
            const FmaForm f{c.s, C, (int32_t)d};
            ++exact_tries;
            if (exact_ok(f)) {
                out = f;
                if (st) st->pivots_tried = i + 1, st->exact_checks += exact_tries, st->s_rank = (int)ci;
                return true;
            }
            if (exact_tries > 64) break; // (the window is exact: a candidate inside it that fails means the model of the rounding is wrong)
        }
        if (st) st->exact_checks += exact_tries;
    }
    if (!allow_patch) return false;
    return patch_search();
}

// every accumulator of [amin, amax]: the host-side twin of the device verifier (tests)
uint64_t fma_form_mismatches(float A, float S, int off, int lo, int hi, int64_t amin, int64_t amax, const FmaForm &f) {
    uint64_t bad = 0;
    for (int64_t a = amin; a <= amax; ++a) bad += ref_form_eval(A, S, off, lo, hi, a) != fma_form_eval(f, a);
    return bad;
}

} // namespace mf
