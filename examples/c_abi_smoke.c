/* The drop-in boundary is a C ABI: this file uses it from plain C (no C++, no Python, no torch).
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_smoke.c -o c_abi_smoke -ldl && ./c_abi_smoke drloco_amd/csrc/libdrloco_hip.so
 *
 * It loads the library, checks the ABI version and the struct sizes the header promises, and asks for an environment:
 * on a machine without a HIP device that must fail with DL_E_NODEVICE (there is no CPU fallback behind these symbols);
 * with one, the descriptors below are deliberately empty, so dl_create must refuse them with DL_E_INVAL -- a real caller
 * fills dl_model_desc / dl_refs_desc from the MJCF and the mocap file (drloco_amd/mjcf.py, drloco_amd/mocap.py do that
 * for the Python host).  tests/test_abi.py builds and runs it. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "drloco_hip.h"

#define LOAD(name) \
    name##_fn name##_p = (name##_fn)dlsym(lib, #name); \
    if (!name##_p) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

typedef int (*dl_abi_version_fn)(void);
typedef int (*dl_abi_sizeof_fn)(int);
typedef const char* (*dl_last_error_fn)(void);
typedef int (*dl_create_fn)(const dl_model_desc*, const dl_refs_desc*, const dl_config*, int32_t, int32_t, dl_handle*);
typedef int (*dl_destroy_fn)(dl_handle);
typedef int (*dl_fault_check_fn)(dl_handle, int32_t*);

int main(int argc, char** argv) {
    void* lib = dlopen(argc > 1 ? argv[1] : "drloco_amd/csrc/libdrloco_hip.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(dl_abi_version) LOAD(dl_abi_sizeof) LOAD(dl_last_error) LOAD(dl_create) LOAD(dl_destroy) LOAD(dl_fault_check)
    if (dl_abi_version_p() != DL_ABI_VERSION) { fprintf(stderr, "ABI %d, header %d\n", dl_abi_version_p(), DL_ABI_VERSION); return 3; }
    if (dl_abi_sizeof_p(0) != (int)sizeof(dl_model_desc) || dl_abi_sizeof_p(1) != (int)sizeof(dl_refs_desc) || dl_abi_sizeof_p(2) != (int)sizeof(dl_config) ||
        dl_abi_sizeof_p(3) != (int)sizeof(dl_policy_params) || dl_abi_sizeof_p(4) != (int)sizeof(dl_vecnorm_state)) { fprintf(stderr, "struct sizes differ from the header's\n"); return 3; }
    static dl_model_desc model; static dl_refs_desc refs; static dl_config cfg;
    memset(&model, 0, sizeof model); memset(&refs, 0, sizeof refs); memset(&cfg, 0, sizeof cfg);
    cfg.env_kind = DL_ENV_STRAIGHT; cfg.precision = 32;
    dl_handle h = 0;
    const int rc = dl_create_p(&model, &refs, &cfg, 16, 0, &h);
    char why[256];
    strncpy(why, dl_last_error_p(), sizeof why - 1); why[sizeof why - 1] = 0;
    if (rc == DL_OK) { fprintf(stderr, "an empty model descriptor was accepted\n"); dl_destroy_p(h); return 4; }
    if (rc != DL_E_NODEVICE && rc != DL_E_INVAL) { fprintf(stderr, "unexpected error %d: %s\n", rc, dl_last_error_p()); return 4; }
    if (dl_fault_check_p(0, 0) != DL_E_INVAL) { fprintf(stderr, "a null handle must be refused\n"); return 4; }
    printf("ok: ABI %d, dl_create -> %d (%s)\n", dl_abi_version_p(), rc, why);
    return 0;
}
