// mcd_prep.h -- device-only: one resolved parameter row (reference column order) -> the derived per-walker constants the
// main kernel reads (mcd_math.h: WalkerConsts).  Shared by the walker-prep kernel (mcd_kernels.hip) and by the resident
// stretch-move chain (mcd_stretch.hip), which builds its parameter rows on the device: the same instructions on the same
// numbers, so a chain that proposes on the device evaluates bit for bit what the host-driven one evaluates.
#pragma once

#include "mcd_math.h"

namespace mcd {

constexpr double kDeg2Rad = 0.017453292519943295769;

// CONST:   v_sys, sigma_max, v_maxx, v_maxy [, ra_c, dec_c] [, v_back, sigma_back, f_back]
// PROFILE: v_sys, sigma_max, a, v_maxx, v_maxy, r_peak [, ra_c, dec_c] [, v_back, sigma_back, f_back | f_back]
template <class T>
__device__ __forceinline__ void walker_constants(const double* __restrict__ p, int model, bool free_centre, T* __restrict__ w) {
    const bool prof = is_profile(model);
    const double sigma = p[1];
    const double a = prof ? p[2] : 0.0, rp = prof ? p[5] : 0.0;
    w[W_VSYS] = (T)p[0];
    w[W_S2] = (T)(sigma * sigma);                                   // runner.py:261 (sigma_los * sigma_los)
    w[W_VX] = (T)(prof ? p[3] : p[2]);
    w[W_VY] = (T)(prof ? p[4] : p[3]);
    w[W_A2] = (T)(a * a); w[W_S2A] = (T)(sigma * sigma * a); w[W_RP2] = (T)(rp * rp); w[W_2RP] = (T)(2.0 * rp);
    w[15] = (T)0;
    int j = prof ? 6 : 4;
    double sac = 0, cac = 1, sdc = 0, cdc = 1;
    if (free_centre) {
        sincos(p[j] * kDeg2Rad, &sac, &cac);
        sincos(p[j + 1] * kDeg2Rad, &sdc, &cdc);
        j += 2;
    }
    w[W_SAC] = (T)sac; w[W_CAC] = (T)cac; w[W_SDC] = (T)sdc; w[W_CDC] = (T)cdc;
    double vb = 0, sb = 0, fb = 0;
    const int bg = bg_kind(model);
    if (bg == BG_GAUSS) { vb = p[j]; sb = p[j + 1]; fb = p[j + 2]; }
    else if (bg == BG_FIXED_DENSITY) { fb = p[j]; }
    w[W_VB] = (T)vb; w[W_SB2] = (T)(sb * sb); w[W_FB] = (T)fb;
}

}  // namespace mcd
