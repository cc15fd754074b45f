/* Plain-C consumer of include/pfgrad.h: proves the header is valid C (no C++/torch types),
 * that every entry point links, exercises the no-device error path, and -- on a GPU -- runs a known answer of
 * the REFERENCE (tests/golden/pf_trace.npz:c0 as C arrays, known_answer.h) through pfg_run and compares numbers.
 * Built and run by tests/test_capi_symbols.py::test_plain_c_consumer with gcc -std=c99 -pedantic -Werror. */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "pfgrad.h"
#include "known_answer.h"

static int close_to(double got, double ref, double rtol, double atol) { return fabs(got - ref) <= atol + rtol * fabs(ref); }

int main(void) {
    pfg_ctx *ctx = NULL;
    pfg_problem p;
    pfg_result r;
    int rc, h, bad = 0;
    memset(&p, 0, sizeof p);
    memset(&r, 0, sizeof r);
    /* sgmcmc_ssm: SVMHelper.pf_gradient_estimate(pf='poyiadjis_N', N=32, subsequence_start=3, subsequence_end=13, weights=...)
     * after np.random.seed(1000), as buffered_pf_wrapper sees it (particle_filters/buffered_smoother.py:156-199) */
    p.model = PFG_MODEL_SVM; p.kernel = PFG_KERNEL_PRIOR; p.smoother = PFG_SMOOTHER_NEMETH;
    p.stat = PFG_STAT_SCORE; p.dtype = PFG_F64; p.rng = PFG_RNG_REPLAY;
    p.N = KA_N; p.T = KA_T; p.t1 = KA_T1; p.tL = KA_TL; p.lambduh = 1.0;
    p.prior_mean = KA_PRIOR_MEAN; p.prior_var = KA_PRIOR_VAR;
    p.y = KA_Y; p.weights = KA_WEIGHTS; p.theta = KA_THETA; p.z0 = KA_Z0; p.u = KA_U; p.z = KA_Z;
    if (pfg_version() != PFG_VERSION) return 10;
    if (pfg_struct_size(0) != (int)sizeof(pfg_problem) || pfg_struct_size(1) != (int)sizeof(pfg_result) ||
        pfg_struct_size(2) != (int)sizeof(pfg_dev_problem) || pfg_struct_size(3) != (int)sizeof(pfg_prior_hyper))
        return 11;
    rc = pfg_create(&ctx, 0);
    if (rc != PFG_OK) {                       /* no GPU here: must fail loudly, with a message */
        const char *msg = pfg_last_error(NULL);
        printf("create failed as expected: %d %s\n", rc, msg ? msg : "(null)");
        return (rc == PFG_ERR_DEVICE && msg && strlen(msg) > 0) ? 0 : 12;
    }
    rc = pfg_run(ctx, &p, &r);
    printf("run rc=%d loglik=%.17g stat=[%.17g, %.17g, %.17g]\n", rc, r.loglik, r.mean_stat[0], r.mean_stat[1], r.mean_stat[2]);
    if (rc != PFG_OK) { printf("error: %s\n", pfg_last_error(ctx)); pfg_destroy(ctx); return 13; }
    /* the reference's numbers, rtol 1e-9 */
    if (!close_to(r.loglik, KA_LOGLIK, 1e-9, 1e-9)) { printf("loglik differs: %.17g vs %.17g\n", r.loglik, KA_LOGLIK); bad = 1; }
    for (h = 0; h < 3; ++h)
        if (!close_to(r.mean_stat[h], KA_MEAN_STAT[h], 1e-9, 1e-8)) {
            printf("mean_stat[%d] differs: %.17g vs %.17g\n", h, r.mean_stat[h], KA_MEAN_STAT[h]);
            bad = 1;
        }
    if (!bad) printf("known answer ok: |loglik err| = %.3g\n", fabs(r.loglik - KA_LOGLIK));
    /* bad arguments come back as codes with a message, not as crashes */
    p.N = 0;
    rc = pfg_run(ctx, &p, &r);
    if (rc != PFG_ERR_INVALID || strlen(pfg_last_error(ctx)) == 0) { printf("N = 0 was not refused (rc=%d)\n", rc); bad = 1; }
    /* reference the rest of the ABI so that the link step checks it */
    (void)pfg_run_batch; (void)pfg_launch_device; (void)pfg_launch_device_smoother; (void)pfg_launch_device_traced;
    (void)pfg_sgld_update_device; (void)pfg_imq_ksd; (void)pfg_scratch_bytes; (void)pfg_variant_name;
    (void)pfg_ctx_stream; (void)pfg_synchronize; (void)pfg_last_traced;
    pfg_destroy(ctx);
    return bad ? 14 : 0;
}
