/* The C-ABI of include/fxcorr.h used from plain C (C99, no HIP, no C++): the header must be a valid C header and
 * the library must link from C.  TEST INFRASTRUCTURE ONLY.  No device work: argument validation and the
 * no-device error path (a GPU box gets further and creates + destroys a plan). */
#include <stdio.h>
#include <string.h>

#include "fxcorr.h"

int main(void) {
    static double window[4 * 64];
    fxc_plan* plan = NULL;
    fxc_info info;
    void* pinned = NULL;
    int n_dev = -1, rc, i;
    for (i = 0; i < 4 * 64; ++i) window[i] = 1.0 / (1.0 + i);
    if (fxc_version() != FXC_VERSION) return 10;
    if (strcmp(fxc_status_string(FXC_OK), "ok") != 0) return 11;
    if (fxc_device_count(&n_dev) != FXC_OK || n_dev < 0) return 12;
    if (fxc_plan_create(&plan, 0, 2, 64, 33, 4096, window, NULL, -1) != FXC_ERR_UNSUPPORTED) return 13; /* ntaps > 32 */
    if (fxc_plan_create(NULL, 0, 2, 64, 4, 4096, window, NULL, -1) != FXC_ERR_ARG) return 14;
    if (fxc_reduce(NULL, NULL, 0) != FXC_ERR_ARG || fxc_comm_destroy(NULL) != FXC_OK) return 15;
    if (fxc_fx_rows_iq(NULL, NULL, NULL, 1, FXC_MEM_HOST, FXC_MODE_SPECTRUM, 1.0, FXC_IQ_C128, 1) != FXC_ERR_ARG) return 21;
    if (fxc_host_free(NULL) != FXC_OK) return 22;
    rc = fxc_host_alloc(&pinned, 1 << 16);
    if (n_dev == 0 ? (rc != FXC_ERR_NODEVICE || pinned != NULL) : (rc != FXC_OK || pinned == NULL)) return 23;
    if (pinned != NULL) {
        memset(pinned, 0, 1 << 16);
        if (fxc_host_free(pinned) != FXC_OK || fxc_host_free(pinned) != FXC_ERR_ARG) return 24;
    }
    rc = fxc_plan_create(&plan, 0, 2, 64, 4, 4096, window, NULL, -1);
    if (n_dev == 0) {
        if (rc != FXC_ERR_NODEVICE || plan != NULL) return 16;
        if (strstr(fxc_last_error(NULL), "no CPU backend") == NULL) return 17;
        printf("c-abi ok (no device: %s)\n", fxc_last_error(NULL));
        return 0;
    }
    if (rc != FXC_OK || plan == NULL) return 18;
    if (fxc_plan_get_info(plan, &info) != FXC_OK || info.nchan != 64 || info.n_baselines != 1) return 19;
    if (fxc_plan_destroy(plan) != FXC_OK) return 20;
    printf("c-abi ok (%d device(s), path %d)\n", n_dev, (int)info.path);
    return 0;
}
