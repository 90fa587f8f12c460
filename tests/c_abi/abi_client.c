/* A plain-C99 client of include/mcd.h (test infrastructure): proves that the header is valid C, that every entry point
 * links from C, and that the argument checks answer with status codes + mcd_last_error() instead of crashing.
 *
 *   abi_client            no GPU needed: version, symbol table, invalid-argument behaviour
 *   abi_client --gpu      additionally evaluates three hand-computable stars on device 0 (closed form, v_max = 0)
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "mcd.h"

#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) { printf("FAILED line %d: %s\n", __LINE__, #cond); return 1; }   \
    } while (0)

int main(int argc, char** argv) {
    /* every entry point, taken by address: an unresolved one fails the link */
    typedef void (*any_fn)(void);
    const any_fn table[] = {
        (any_fn)mcd_ctx_create, (any_fn)mcd_get_unique_id, (any_fn)mcd_ctx_create_rank,
        (any_fn)mcd_ctx_destroy, (any_fn)mcd_ctx_n_devices, (any_fn)mcd_catalog_create,
        (any_fn)mcd_catalog_destroy, (any_fn)mcd_catalog_param_count, (any_fn)mcd_catalog_n_stars,
        (any_fn)mcd_catalog_n_outputs, (any_fn)mcd_loglike_batch, (any_fn)mcd_params_upload,
        (any_fn)mcd_loglike_enqueue, (any_fn)mcd_loglike_fetch, (any_fn)mcd_sync,
        (any_fn)mcd_membership, (any_fn)mcd_loglike_per_star, (any_fn)mcd_kde_background,
        (any_fn)mcd_last_error, (any_fn)mcd_abi_version, (any_fn)mcd_last_kernel_ms,
        (any_fn)mcd_last_device_ms, (any_fn)mcd_set_option, (any_fn)mcd_timing_collect,
        (any_fn)mcd_rerun_count, (any_fn)mcd_last_fast_level, (any_fn)mcd_last_launch_info,
        (any_fn)mcd_ctx_comm_info, (any_fn)mcd_stretch_move, (any_fn)mcd_stretch_info, (any_fn)mcd_last_prefetch,
        /* round 3 (additions: the ABI version stays) */
        (any_fn)mcd_ctx_set_option, (any_fn)mcd_ctx_abort, (any_fn)mcd_ctx_failed, (any_fn)mcd_last_f32_domain,
        (any_fn)mcd_stretch_move_seeded, (any_fn)mcd_chain_numbers};
    size_t i;
    double out[3] = {0.0, 0.0, 0.0};
    for (i = 0; i < sizeof table / sizeof table[0]; ++i) CHECK(table[i] != NULL);
    CHECK(mcd_abi_version() == MCD_ABI_VERSION);

    /* null handles: status codes and a message, never a crash */
    CHECK(mcd_loglike_batch(NULL, 1, 4, out, out) != MCD_OK);
    CHECK(strlen(mcd_last_error()) > 0);
    CHECK(mcd_catalog_create(NULL, NULL, NULL) != MCD_OK);
    CHECK(mcd_kde_background(NULL, 1, out, 1, out, out, 0.0, out, NULL) != MCD_OK);
    CHECK(mcd_last_fast_level(NULL) == -1);
    CHECK(mcd_last_prefetch(NULL) == -1);
    CHECK(mcd_ctx_comm_info(NULL, NULL, NULL, NULL) != MCD_OK);
    CHECK(mcd_stretch_move(NULL, NULL, 1, out, out, NULL, out, out, NULL, NULL, NULL, NULL) != MCD_OK);
    CHECK(mcd_stretch_info(NULL, NULL, NULL, NULL, NULL) != MCD_OK);
    CHECK(mcd_stretch_move_seeded(NULL, NULL, 1, out, out, 1, 0, NULL, NULL, NULL) != MCD_OK);
    CHECK(mcd_chain_numbers(1, 0, 1, 1, 3, 2, NULL, NULL, NULL, NULL) != MCD_OK);      /* odd walker count, null outputs */
    CHECK(mcd_ctx_set_option(NULL, "collective_timeout_ms", 1) != MCD_OK);
    CHECK(mcd_ctx_failed(NULL) == 0);
    CHECK(mcd_catalog_destroy(NULL) == MCD_OK);
    CHECK(mcd_ctx_destroy(NULL) == MCD_OK);
    printf("abi %d: %d entry points link from C\n", mcd_abi_version(), (int)(sizeof table / sizeof table[0]));

    if (argc > 1 && strcmp(argv[1], "--gpu") == 0) {
        /* SURVEY 8(c) known answer (1): v_max = 0 => lnL = sum -1/2 [log(2 pi (e^2 + s^2)) + (v - v_sys)^2 / (e^2 + s^2)] */
        const double pi = 3.14159265358979323846;
        double ra[3] = {10.0, 10.01, 9.99}, dec[3] = {0.0, 0.01, -0.01};
        double v[3] = {1.0, -2.0, 0.5}, verr[3] = {1.0, 2.0, 0.5};
        double rows[2][4] = {{0.25, 3.0, 0.0, 0.0}, {0.0, 7.5, 0.0, 0.0}};
        double got[2], want[2] = {0.0, 0.0};
        mcd_ctx* ctx = NULL;
        mcd_catalog* cat = NULL;
        mcd_catalog_desc d;
        int w, s;
        memset(&d, 0, sizeof d);
        d.n_stars = 3; d.ra = ra; d.dec = dec; d.v = v; d.verr = verr;
        d.model = MCD_MODEL_CONST; d.centre = MCD_CENTRE_FIXED; d.precision = MCD_F64;
        d.ra_center = 10.0; d.dec_center = 0.0;
        if (mcd_ctx_create(1, NULL, &ctx) != MCD_OK) { printf("ctx: %s\n", mcd_last_error()); return 1; }
        if (mcd_catalog_create(ctx, &d, &cat) != MCD_OK) { printf("catalog: %s\n", mcd_last_error()); return 1; }
        CHECK(mcd_catalog_param_count(cat) == 4 && mcd_catalog_n_stars(cat) == 3);
        CHECK(mcd_loglike_batch(cat, 2, 4, &rows[0][0], got) == MCD_OK);
        for (w = 0; w < 2; ++w)
            for (s = 0; s < 3; ++s) {
                const double n = verr[s] * verr[s] + rows[w][1] * rows[w][1], dv = v[s] - rows[w][0];
                want[w] += -0.5 * (log(2.0 * pi * n) + dv * dv / n);
            }
        for (w = 0; w < 2; ++w) CHECK(fabs(got[w] - want[w]) < 1e-13);
        CHECK(mcd_loglike_batch(cat, 2, 5, &rows[0][0], got) != MCD_OK);      /* wrong column count */
        {
            /* one stretch-move step of two walkers driven from C: identity column map, no bounds; with z = 1 the proposal
             * is the walker itself, so the log-probability is unchanged and (thr = -1 < 0) the "move" is accepted */
            const double inf = 1.0 / 0.0;
            int32_t src[4] = {0, 1, 2, 3}, order[2] = {0, 1}, pick[2] = {0, 0};
            double cst[4] = {0, 0, 0, 0}, fac[4] = {1, 1, 1, 1}, lo[4], hi[4], zz[2] = {1.0, 1.0}, thr[2] = {-1.0, -1.0};
            double pos[2][4] = {{0.25, 3.0, 0.0, 0.0}, {0.0, 7.5, 0.0, 0.0}}, lnp[2], chain[8], lnpc[2];
            int64_t acc[2] = {0, 0};
            int comm_size = -1, comm_rank = 0, version = -1;
            mcd_stretch_desc sd;
            for (w = 0; w < 4; ++w) { lo[w] = -inf; hi[w] = inf; }
            lo[1] = 0.0;
            memset(&sd, 0, sizeof sd);
            sd.n_walkers = 2; sd.n_dim = 4; sd.k = 4; sd.col_source = src; sd.col_const = cst; sd.col_factor = fac;
            sd.lo = lo; sd.hi = hi; sd.fixed_ok = 1;
            lnp[0] = got[0]; lnp[1] = got[1];
            CHECK(mcd_stretch_move(cat, &sd, 1, &pos[0][0], lnp, order, zz, thr, pick, chain, lnpc, acc) == MCD_OK);
            CHECK(acc[0] == 1 && acc[1] == 1 && fabs(lnpc[0] - want[0]) < 1e-13 && fabs(lnpc[1] - want[1]) < 1e-13);
            CHECK(chain[1] == 3.0 && chain[5] == 7.5);
            {
                int64_t on_device = -1, on_host = -1, discarded = -1;
                CHECK(mcd_stretch_info(cat, &on_device, &on_host, &discarded, NULL) == MCD_OK);
                CHECK(on_device + on_host == 1 && discarded <= on_host);
            }
            order[1] = 7;                                                       /* an index outside the ensemble */
            CHECK(mcd_stretch_move(cat, &sd, 1, &pos[0][0], lnp, order, zz, thr, pick, chain, lnpc, acc) == MCD_ERR_INVALID);
            CHECK(mcd_ctx_comm_info(ctx, &comm_size, &comm_rank, &version) == MCD_OK && comm_size == 0 && comm_rank == -1);
        }
        CHECK(mcd_catalog_destroy(cat) == MCD_OK && mcd_ctx_destroy(ctx) == MCD_OK);
        printf("gpu closed form ok: %.15g %.15g\n", got[0], got[1]);
    }
    return 0;
}
