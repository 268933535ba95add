// tests/hip_emu/rccl/rccl.h - TEST INFRASTRUCTURE: the handful of RCCL declarations csrc/nrdhip_tiler.cpp compiles against, for the
// host-emulated build of the UNMODIFIED product sources (the product resolves librccl with dlopen at nrdhip_tiler_rccl_init and never
// links it, so declarations are all the build needs; on a host without librccl that call returns an error, as it does in the product).
#pragma once
#include <stddef.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
