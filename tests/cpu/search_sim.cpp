// tests/cpu/search_sim.cpp -- TEST INFRASTRUCTURE: drives psxavenc_amd/csrc/mdec_search.h (the rate-control search
// policy the MDEC kernel runs between passes) on the CPU against synthetic bits(scale) curves.
#include <stdint.h>

#include "../../psxavenc_amd/csrc/mdec_search.h"

// tb[s], fb[s] for s = 1..63 (index 0 unused): total bits and the proven lower bound for finer scales.
// overflow_bits: an emit pass whose stream exceeds this does not leave a usable staged stream.
// Returns the chosen scale (64 = nothing fits); *passes = passes used; -1 if the search did not terminate,
// -2 if it finished with the wrong stream staged.
extern "C" int search_sim(const int* tb, const int* fb, int limit_bits, int fixed_bits, int guess, int overflow_bits,
                          int* passes, int* evaluated_mask_lo, int* evaluated_mask_hi) {
    MdecSearch st;
    mdec_search_init(st);
    int n = 0;
    uint64_t ev = 0;
    for (;;) {
        MdecPass p = (limit_bits < fixed_bits) ? MdecPass{0, 0, 1} : mdec_search_next(st, guess, limit_bits, fixed_bits);
        if (p.done) break;
        if (++n > 200) return -1;
        if (p.count_scale < 0 || p.count_scale > 63 || p.emit_scale < 0 || p.emit_scale > 63) return -3;
        if (p.count_scale == 0 && p.emit_scale == 0) return -4;
        if (p.count_scale) {
            mdec_search_note(st, p.count_scale, tb[p.count_scale], fb[p.count_scale], limit_bits);
            ev |= 1ull << p.count_scale;
        }
        if (p.emit_scale) {
            mdec_search_note(st, p.emit_scale, tb[p.emit_scale], fb[p.emit_scale], limit_bits);
            ev |= 1ull << p.emit_scale;
            st.staged = tb[p.emit_scale] > overflow_bits ? 0 : p.emit_scale;
        }
    }
    *passes = n;
    *evaluated_mask_lo = (int)(uint32_t)ev;
    *evaluated_mask_hi = (int)(uint32_t)(ev >> 32);
    if (st.best < 64 && st.staged != st.best) return -2;
    return st.best;
}

// The pilot's policy (mdec_pilot_next) on an ESTIMATED curve est[s] (the sample's bits scaled up to the frame): returns its guess;
// *rounds = rounds used, *evals = scales evaluated.
extern "C" int pilot_sim(const int* est, int limit_bits, int fixed_bits, int h0, int* rounds, int* evals) {
    MdecSearch st;
    mdec_search_init(st);
    int r = 0, ne = 0, guess = 0;
    for (;; r++) {
        const MdecPilot pl = mdec_pilot_next(st, h0, limit_bits, fixed_bits, r);
        guess = pl.guess;
        const int n = pl.n;
        if (n == 0) break;
        if (n < 0 || n > 3 || r > 20) return -1;
        for (int j = 0; j < n; j++) {
            if (pl.s[j] < 1 || pl.s[j] > 63) return -2;
            mdec_search_note(st, pl.s[j], est[pl.s[j]], est[pl.s[j]], limit_bits);
            ne++;
        }
    }
    *rounds = r;
    *evals = ne;
    return guess;
}
