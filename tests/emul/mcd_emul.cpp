// TEST INFRASTRUCTURE ONLY -- compiles the kernels' per-chunk arithmetic (csrc/mcd_math.h) for the CPU
// so that the fast-path algebra (fraction tree, log-product, rsqrt/exp mixture) can be checked against
// the golden vectors and the oracle without a GPU.  Never loaded by the product package.
#include <cstdint>
#include <vector>

#include "mcd_math.h"

using namespace mcd;

template <int MODEL, bool FREE, bool FAST>
static void run(int64_t n, const double* recs, const double* wpar, int64_t W, int64_t chunk_len, double* out) {
    constexpr int ND = record_doubles(MODEL, FREE);
    for (int64_t w = 0; w < W; ++w) {
        const double* p = wpar + w * KD;
        WalkerConsts<double> c;
        c.vsys = p[W_VSYS]; c.s2 = p[W_S2]; c.vx = p[W_VX]; c.vy = p[W_VY];
        c.sac = p[W_SAC]; c.cac = p[W_CAC]; c.sdc = p[W_SDC]; c.cdc = p[W_CDC];
        c.vb = p[W_VB]; c.sb2 = p[W_SB2]; c.fb = p[W_FB];
        double total = 0.0;
        for (int64_t s = 0; s < n; s += chunk_len) {
            const int count = (int)((n - s) < chunk_len ? (n - s) : chunk_len);
            total += chunk_loglike<MODEL, FREE, double, double, FAST>(recs + s * ND, count, c);
        }
        if (MODEL == MODEL_BGFIXED && FAST) {        // walker-independent sum of lnL_bg: added by the reduce kernel on the GPU
            constexpr int XB = FREE ? 6 : 4;
            double sb = 0.0;
            for (int64_t i = 0; i < n; ++i) sb += recs[i * ND + XB];
            total += sb;
        }
        out[w] = total;
    }
}

extern "C" int emul_record_doubles(int model, int free_centre) { return record_doubles(model, free_centre != 0); }
extern "C" int emul_kd() { return KD; }

extern "C" int emul_loglike(int model, int free_centre, int fast, int64_t n, const double* recs, const double* wpar,
                            int64_t W, int64_t chunk_len, double* out) {
#define CASE(M, F, X) if (model == M && (free_centre != 0) == F && (fast != 0) == X) { run<M, F, X>(n, recs, wpar, W, chunk_len, out); return 0; }
    CASE(0, false, false) CASE(0, false, true) CASE(0, true, false) CASE(0, true, true)
    CASE(1, false, false) CASE(1, false, true) CASE(1, true, false) CASE(1, true, true)
    CASE(2, false, false) CASE(2, false, true) CASE(2, true, false) CASE(2, true, true)
#undef CASE
    return -1;
}
