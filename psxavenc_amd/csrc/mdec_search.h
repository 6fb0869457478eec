// mdec_search.h -- exact "first quant scale that fits" search without evaluating every scale.
//
// The reference tries scale 1, 2, 3, ... and keeps the first whose bitstream fits (psxavenc/mdec.c:663-723).
// bits(s) is not monotone in s in general, so "the first that fits" cannot be found by bisection on bits(s)
// alone.  What IS monotone is a lower bound:
//
//   going from scale s' to a finer scale s < s' every coefficient's level stays or grows, and zeros may
//   become non-zero, i.e. every AC code (run r, level L) of the s' stream is refined into one or more codes
//   (r_1, l_1) ... (r_k, L'), r_1 + ... + r_k + (k - 1) = r, L' >= L.  Let G(r, L) be the cheapest such
//   refinement (a small dynamic programme over the VLC table, tools/gen_tables.py); G == len for all but
//   21 table codes and the run-escapes.  Then for every s <= s':   bits_AC(s) >= sum over the s' codes of G,
//   DC, end-of-block and end-of-frame bits being scale-independent (mdec.c:671).
//
// So one evaluation at s' that yields  fbits(s') = bits(s') - sum (len - G)  >  limit  PROVES that no scale
// <= s' fits.  The search below keeps
//     lo    highest scale with that proof        (all scales <= lo fail)
//     best  lowest scale known to fit            (64 = none yet; scale 64 stands for "nothing fits")
//     fail  scales evaluated, not fitting, proof or not
// and is finished when every scale in (lo, best) is in `fail`: best is then the first scale that fits, exactly
// the reference's answer.  How the next scales to evaluate are chosen only affects speed: a pass over the frame
// can count the bits at one scale and build the bitstream at another (which also yields its bits), and the
// choice is steered by a two-point model  AC bits ~ a + b / scale  through the closest evaluated scales.
//
// Plain C++ with no device dependencies: the kernel (mdec_kernels.hip) and the CPU test of the policy
// (tests/test_mdec_search.py through tests/cpu/search_sim.cpp) compile the same text.
#pragma once
#include <stdint.h>

#ifndef PSX_HD
#if defined(__HIPCC__)
#define PSX_HD __host__ __device__ __forceinline__
#else
#define PSX_HD inline
#endif
#endif

struct MdecSearch {
    int lo;          // all scales <= lo are proven not to fit
    int best;        // lowest scale known to fit, 64 = none
    int staged;      // scale whose bitstream currently sits in the staging area (0 = none)
    int pad;
    uint64_t fail;   // bit s set: scale s was evaluated and does not fit
    int fs[2], fb[2];   // the two highest failing scales evaluated (fs[0] > fs[1], 0 = none) and their total bits
    int gs[2], gb[2];   // the two lowest fitting scales evaluated (gs[0] < gs[1], 0 = none) and their total bits
};

struct MdecPass {
    int count_scale;   // 0 = none: count bits only
    int emit_scale;    // 0 = none: build the bitstream (also yields its bits)
    int done;          // 1: search finished, result = best (64 = nothing fits), staged == best unless best == 64
};

PSX_HD void mdec_search_init(MdecSearch& s) {
    s.lo = 0;
    s.best = 64;
    s.staged = 0;
    s.pad = 0;
    s.fail = 0;
    s.fs[0] = s.fs[1] = s.fb[0] = s.fb[1] = 0;
    s.gs[0] = s.gs[1] = s.gb[0] = s.gb[1] = 0;
}

// record one evaluated scale: total bits and the proven lower bound for all finer scales
PSX_HD void mdec_search_note(MdecSearch& s, int scale, int tbits, int fbits, int limit_bits) {
    if (tbits <= limit_bits) {
        if (scale < s.best) s.best = scale;
        if (s.gs[0] == 0 || scale < s.gs[0]) {
            s.gs[1] = s.gs[0]; s.gb[1] = s.gb[0];
            s.gs[0] = scale;   s.gb[0] = tbits;
        } else if (scale != s.gs[0] && (s.gs[1] == 0 || scale < s.gs[1])) {
            s.gs[1] = scale;   s.gb[1] = tbits;
        }
    } else {
        s.fail |= 1ull << scale;
        if (fbits > limit_bits && scale > s.lo) s.lo = scale;
        if (scale > s.fs[0]) {
            s.fs[1] = s.fs[0]; s.fb[1] = s.fb[0];
            s.fs[0] = scale;   s.fb[0] = tbits;
        } else if (scale != s.fs[0] && scale > s.fs[1]) {
            s.fs[1] = scale;   s.fb[1] = tbits;
        }
    }
}

// smallest scale the model expects to fit.  Model: AC bits = a + b * x, x = 1 / scale, through two evaluated scales
// (or b * x through one).  `room` = bits left for the AC codes.
PSX_HD int mdec_search_predict(const MdecSearch& s, int guess, int room, int fixed_bits) {
    int s1 = 0, y1 = 0, s2 = 0, y2 = 0;       // two model points, s1 < s2
    if (s.fs[0] && s.gs[0] && s.fs[0] < s.gs[0]) {
        s1 = s.fs[0]; y1 = s.fb[0]; s2 = s.gs[0]; y2 = s.gb[0];       // bracket: interpolate
    } else if (s.fs[0]) {
        s2 = s.fs[0]; y2 = s.fb[0]; s1 = s.fs[1]; y1 = s.fb[1];       // failures only: extrapolate upwards
    } else if (s.gs[0]) {
        s1 = s.gs[0]; y1 = s.gb[0]; s2 = s.gs[1]; y2 = s.gb[1];       // fits only: extrapolate downwards
    } else {
        return guess;
    }
    if (room <= 0) return 63;
    const float r = (float)room;
    float x;      // predicted 1 / scale
    if (s1 && s2 && y1 > y2) {
        const float x1 = 1.0f / (float)s1, x2 = 1.0f / (float)s2;
        float b = (float)(y1 - y2) / (x1 - x2);
        // Two points on the SAME side of the limit that lie close together say little about the slope when they are projections
        // from a sample (the pilot, the quarter-pass checkpoint): their difference is mostly noise, and a line that is nearly flat
        // crosses the limit anywhere (seen: hint 6, both 5 and 6 projected a little too big -> "61"; a frame whose first quarter
        // over-weighs its busy rows: answer 5 -> "39").  The scale-dependent share of the AC bits is never small -- three quarters
        // of them and more scale with 1 / scale on everything measured -- so the slope is held to at least a quarter of the
        // one-point model's (b / scale through the point nearer the limit).  A bracket is interpolated as it is.
        const bool bracket = s.fs[0] && s.gs[0] && s.fs[0] < s.gs[0];
        if (!bracket) {
            const bool fits_only = !s.fs[0];
            const float xn = fits_only ? x1 : x2, yn = (float)((fits_only ? y1 : y2) - fixed_bits);
            if (yn > 0.0f && b < 0.25f * yn / xn) b = 0.25f * yn / xn;
        }
        float xa = x2, ya = (float)(y2 - fixed_bits);        // the line's anchor: the point nearer the limit
        if (!bracket && !s.fs[0]) { xa = x1; ya = (float)(y1 - fixed_bits); }
        x = xa + (r - ya) / b;
    } else {
        const int sa = s2 ? s2 : s1, ya = (s2 ? y2 : y1) - fixed_bits;
        if (ya <= 0) return 1;
        x = r / ((float)ya * (float)sa);
    }
    if (!(x > 1.0f / 64.0f)) return 63;
    if (x >= 1.0f) return 1;
    const float sf = 1.0f / x;
    int p = (int)sf;
    if ((float)p < sf) p++;       // ceil
    return p < 1 ? 1 : (p > 63 ? 63 : p);
}

// Quarter-pass checkpoint: `pa` / `pb` = the frame's total bits at the pass's count / emit scale as PROJECTED from the
// macroblocks done so far (0 = that scale is not part of the pass).  Returns 0 to carry on, or a new guess when the
// projection is clearly (margin_permille of the AC room) on the wrong side: the emit scale will not fit, or the count
// scale fits already.  Projections never enter the search state -- they only steer.
// (`margin` in bits: how far a projection has to be on the wrong side)
PSX_HD int mdec_search_checkpoint_bits(const MdecSearch& s, int a, int pa, int b, int pb, int limit_bits, int fixed_bits,
                                       int margin) {
    const int room = limit_bits - fixed_bits;
    if (room <= 0) return 0;
    // (what an exact evaluation has settled no projection overrules: a frame on the edge -- the true bits of scale 5 within a
    //  standard error of the limit -- had its pass at 5 stopped as "will not fit", learned from the exact count of the next pass
    //  that 5 fits, and had the pass that then emitted 5 stopped by the same projection again: four passes, '5!/6/5!/5')
    const bool b_settled = b && b == s.best;                                   // known to fit
    const bool a_settled = a && (a <= s.lo || ((s.fail >> a) & 1ull));         // known not to fit
    const bool too_low = b && !b_settled && pb > limit_bits + margin;       // the stream being built will not fit
    const bool too_high = a && !a_settled && pa <= limit_bits - margin;     // the scale below it fits as well
    if (!too_low && !too_high) return 0;
    MdecSearch t = s;
    if (a) mdec_search_note(t, a, pa, 0, limit_bits);
    if (b) mdec_search_note(t, b, pb, 0, limit_bits);
    int g = mdec_search_predict(t, b ? b : a, room, fixed_bits);
    const int cur = b ? b : a + 1;
    if (too_low && g <= cur) g = cur + 1;
    if (too_high && g >= cur) g = cur - 1;
    if (g <= s.lo) g = s.lo + 1;
    if (g < 1) g = 1;
    if (g > 63) g = 63;
    return g == cur ? 0 : g;
}

// ... with the margin given in permille of the AC room
PSX_HD int mdec_search_checkpoint(const MdecSearch& s, int a, int pa, int b, int pb, int limit_bits, int fixed_bits,
                                  int margin_permille) {
    const int room = limit_bits - fixed_bits;
    if (room <= 0) return 0;
    return mdec_search_checkpoint_bits(s, a, pa, b, pb, limit_bits, fixed_bits, (int)((long long)room * margin_permille / 1000));
}

// The pilot (a sample of the frame's macroblocks, their AC bits at a few scales scaled up to the frame) steered by the same model:
// the sample's estimates are recorded like evaluations (mdec_search_note with fbits = tbits: `lo` is then the highest scale
// ESTIMATED not to fit, `best` the lowest estimated to fit -- a state of its own, never the exact search's), and the scales to try
// next are the model's prediction and its neighbours.  `round` 0: nothing evaluated yet -- a hint h0 (2..63) is checked first
// (h0 - 1, h0), without one two scales a factor of four apart give the model its two points.  Returns the scales to
// evaluate next (at most 3), or n = 0 with the predicted answer.  Typically two rounds, five evaluations
// (the bracketing by halving it replaces took three to five rounds of four).
struct MdecPilot {
    int n;          // scales to evaluate next (0: finished)
    int s[3];       // ... the scales (unused entries 0)
    int guess;      // n == 0: the predicted answer
};
PSX_HD MdecPilot mdec_pilot_next(const MdecSearch& s, int h0, int limit_bits, int fixed_bits, int round) {
    MdecPilot r;
    r.n = 0; r.s[0] = 0; r.s[1] = 0; r.s[2] = 0; r.guess = 0;
    if (round == 0) {
        const bool hinted = h0 >= 2 && h0 <= 63;
        r.n = 2;
        r.s[0] = hinted ? h0 - 1 : 2;
        r.s[1] = hinted ? h0 : 8;
        return r;
    }
    const int lo = s.lo;
    const int hi = s.best > lo ? s.best : 64;       // (a fit below a failure: a non-monotone estimate -- trust the failure, look above it)
    if (hi - lo <= 1 || lo >= 63 || round >= 6) {
        r.guess = hi > 63 ? 63 : hi;
        return r;
    }
    int p = mdec_search_predict(s, h0, limit_bits - fixed_bits, fixed_bits);
    const int top = hi > 63 ? 63 : hi - 1;          // highest scale still open
    if (p <= lo) p = lo + 1;
    if (p > top) p = top;
    const bool below = p - 1 > lo, above = p + 1 <= top;       // (no indexed stores: the struct stays in registers)
    r.s[0] = below ? p - 1 : p;
    r.s[1] = below ? p : (above ? p + 1 : 0);
    r.s[2] = below && above ? p + 1 : 0;
    r.n = 1 + (below ? 1 : 0) + (above ? 1 : 0);
    return r;
}

// what to do next.  `guess` = predicted answer (used until something has been evaluated), `fixed_bits` = the
// scale-independent part of the total (DC + end-of-block + end-of-frame codes)
PSX_HD MdecPass mdec_search_next(const MdecSearch& s, int guess, int limit_bits, int fixed_bits) {
    MdecPass p;
    p.count_scale = 0;
    p.emit_scale = 0;
    p.done = 0;
    const int top = s.best < 64 ? s.best : 63;           // highest scale that can still be the answer
    // scales in (lo, best) that are not known to fail
    uint64_t open = ~s.fail & (s.best < 64 ? (1ull << s.best) - 1ull : ~0ull);
    open &= ~((2ull << s.lo) - 1ull);                    // drop scales 0..lo (lo = 63: everything)
    if (open == 0) {
        // every scale below `best` fails: best is the answer
        if (s.best < 64 && s.staged != s.best) {
            p.emit_scale = s.best;
            return p;
        }
        p.done = 1;
        return p;
    }
    int b = mdec_search_predict(s, guess, limit_bits - fixed_bits, fixed_bits);
    if (b <= s.lo) b = s.lo + 1;
    if (b > top) b = top;
    if (b != s.best && !((open >> b) & 1ull)) {
        // predicted scale already known to fail: nearest open scale above it, else `best`, else the nearest below
        int u = b + 1;
        while (u <= top && u != s.best && !((open >> u) & 1ull)) u++;
        if (u > top) {
            u = b - 1;
            while (!((open >> u) & 1ull)) u--;
        }
        b = u;
    }
    if (b == s.best && s.staged == s.best) {
        // the candidate's bitstream is staged already: only the gap below it is left
        int c = s.best - 1;
        while (!((open >> c) & 1ull)) c--;
        p.count_scale = c;
        return p;
    }
    p.emit_scale = b;
    if (b - 1 > s.lo && ((open >> (b - 1)) & 1ull)) p.count_scale = b - 1;
    return p;
}
