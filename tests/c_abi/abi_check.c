/* Plain-C consumer of include/pfgrad.h: proves the header is valid C (no C++/torch types),
 * that every entry point links, and exercises the no-device error path.
 * Built and run by tests/test_capi_symbols.py::test_plain_c_consumer with gcc. */
#include <stdio.h>
#include <string.h>
#include "pfgrad.h"

int main(void) {
    pfg_ctx *ctx = NULL;
    double y[2] = {0.1, -0.2}, theta[3] = {0.9, 1.0, 1.0}, z0[4] = {0}, u[8] = {0}, z[8] = {0};
    pfg_problem p;
    pfg_result r;
    memset(&p, 0, sizeof p);
    memset(&r, 0, sizeof r);
    p.model = PFG_MODEL_SVM; p.kernel = PFG_KERNEL_PRIOR; p.smoother = PFG_SMOOTHER_NEMETH;
    p.stat = PFG_STAT_SCORE; p.dtype = PFG_F64; p.rng = PFG_RNG_REPLAY;
    p.N = 4; p.T = 2; p.t1 = 0; p.tL = 2; p.lambduh = 1.0; p.prior_var = 1.0;
    p.y = y; p.theta = theta; p.z0 = z0; p.u = u; p.z = z;
    if (pfg_version() != PFG_VERSION) return 10;
    if (pfg_struct_size(0) != (int)sizeof(pfg_problem) || pfg_struct_size(1) != (int)sizeof(pfg_result) ||
        pfg_struct_size(2) != (int)sizeof(pfg_dev_problem) || pfg_struct_size(3) != (int)sizeof(pfg_prior_hyper))
        return 11;
    int rc = pfg_create(&ctx, 0);
    if (rc != PFG_OK) {                       /* no GPU here: must fail loudly, with a message */
        const char *msg = pfg_last_error(NULL);
        printf("create failed as expected: %d %s\n", rc, msg ? msg : "(null)");
        return (rc == PFG_ERR_DEVICE && msg && strlen(msg) > 0) ? 0 : 12;
    }
    rc = pfg_run(ctx, &p, &r);
    printf("run rc=%d loglik=%.17g stat0=%.17g\n", rc, r.loglik, r.mean_stat[0]);
    /* reference the rest of the ABI so that the link step checks it */
    (void)pfg_run_batch; (void)pfg_launch_device; (void)pfg_launch_device_smoother;
    (void)pfg_sgld_update_device; (void)pfg_imq_ksd; (void)pfg_scratch_bytes; (void)pfg_variant_name;
    (void)pfg_ctx_stream; (void)pfg_synchronize;
    pfg_destroy(ctx);
    return rc == PFG_OK ? 0 : 13;
}
