// altro_device_info: name and CU count of the handle's device (include/altro_hip.h).
#include <hip/hip_runtime.h>
#include <cstring>
#include "../../include/altro_hip.h"
extern "C" altro_status altro_device_info(altro_handle, char* name, int name_len, int* cu_count) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return ALTRO_HIP_ERROR;
  if (name && name_len > 0) {
    std::strncpy(name, p.name, name_len - 1);
    name[name_len - 1] = 0;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  return ALTRO_OK;
}
