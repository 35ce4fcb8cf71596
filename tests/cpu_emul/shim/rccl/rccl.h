// TEST INFRASTRUCTURE (tests/cpu_emul): the few RCCL types rtwin_capi.cpp names; the library itself is only ever reached through dlopen.
#pragma once
#include <cstddef>
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclFloat32 = 7, ncclFloat = 7 } ncclDataType_t;
